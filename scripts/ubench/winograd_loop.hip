// Microbenchmark (round 5, verdict item 3): would Winograd F(2x2,3x3) with an in-LDS input transform beat conv3's direct main loop on
// gfx950?  Shape: the MFMA-bound 3x3 stride-1 layers (256 ch @ 64^2 and up), a block's 512-pixel tile, 16-channel chunks.
//
//   direct   (conv3 today): per chunk a 20-KiB A patch (34 x 18 pixels x 32 B) + an 18-KiB weight slab (9 taps x 64 couts x 32 B) by
//            LDS-DMA into two stages, two blocks per CU; per wave 9 taps x (2 weight + 4 pixel fragments) -> 72 MFMAs; the block
//            produces 512 px x 64 couts.
//   winograd F(2x2,3x3): the same 20-KiB patch + a 16-KiB slab of TRANSFORMED weights U = G g G^T (16 positions x 32 couts x 32 B);
//            every thread transforms one (4x4-pixel tile, 8-channel half) - 16 ds_read_b128, B^T d B in packed fp16, 16 ds_write_b128
//            into the 64-KiB V image [16 positions][128 tiles][2 x 16 B] - then wave w contracts positions 4w .. 4w+3: per position
//            1 weight + 4 tile fragments -> 16 MFMAs per wave and chunk (32 x 32 x 16 each), 16 accumulator tiles per wave = 256
//            registers: the block can hold 32 couts only (64 couts = 512 accumulator registers per wave), ONE block per CU (136 KiB
//            of LDS, 1 wave per SIMD), and a second block repeats the input transform for the other 32 couts.
// Both move their operands with the same LDS-DMA + ds_read_b128 machinery; the output transform (once per item, A^T M A on the
// accumulators) is NOT included - it only adds to the Winograd side.  Output: direct-equivalent TFLOP/s of both loops
// (2 x 9 x 16 MACs per output pixel, cout and chunk for both) and the ratio.  Kill criterion set in advance: schedule Winograd at
// >= 1.3x, drop it for good below.
// Build: hipcc --offload-arch=gfx950 -O3 winograd_loop.hip -o winograd_loop
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GLDS16(gptr, lptr)                                                                   \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

constexpr int A_BYTES = 20480;                 // 34 x 18 patch pixels x 32 B, padded to 20 copies of 1 KiB
constexpr int PW = 34;

// ---------------------------------------------------------------------------------------------------- direct (conv3's loop)
constexpr int BD_BYTES = 18432, STAGE_D = A_BYTES + BD_BYTES;
__global__ __launch_bounds__(256, 2) void direct_kernel(const uint4* __restrict__ A, const uint4* __restrict__ B, float* __restrict__ out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint4* Ab = A + (size_t)blockIdx.x * chunks * (A_BYTES / 16);
    const int nB = (wave < 2) ? 5 : 4;
    auto stage = [&](int c, int buf) {
        unsigned char* const Sa = smem + buf * STAGE_D;
        const uint4* ac = Ab + (size_t)c * (A_BYTES / 16);
#pragma unroll
        for (int k = 0; k < 5; ++k) GLDS16(ac + (k * 4 + wave) * 64 + lane, Sa + (k * 4 + wave) * 1024);
        const uint4* bc = B + (size_t)(c & 15) * (BD_BYTES / 16);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (k < nB) GLDS16(bc + k * 256 + tid, Sa + A_BYTES + (k * 256 + wave * 64) * 16);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int aj[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = (wave * 4 + j) * 32 + l31, tx = m & 31, ty = m >> 5;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) aj[j][dx] = (ty * PW + tx + dx) * 32 + (((((tx + dx) >> 3) & 1) ^ hh) << 4);
    }
    auto compute = [&](int buf) {
        const unsigned char* Sa = smem + buf * STAGE_D;
        const unsigned char* Sb = Sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) asm volatile("" : "+v"(aj[j][dx]));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned char* Ar = Sa + (t / 3) * (PW * 32);
            f16x8 wf[2], xa[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const f16x8*>(Sb + ((((i * 9 + t) * 2 + hh) * 32) + l31) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ar + aj[j][t % 3]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xa[j], acc[i][j], 0, 0, 0);
        }
    };
    stage(0, 0);
    for (int c = 0; c < chunks; ++c) {
        __syncthreads();
        if (c + 1 < chunks) stage(c + 1, (c + 1) & 1);
        compute(c & 1);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][9];
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}

// ---------------------------------------------------------------------------------------------------- Winograd F(2x2, 3x3)
constexpr int BW_BYTES = 16 * 32 * 32, STAGE_W = A_BYTES + BW_BYTES;      // U: [16 positions][2 k8 planes][32 couts][16 B]
constexpr int V_BYTES = 16 * 128 * 32;                                     // V: [16 positions][128 tiles][2 x 16 B]
// TRANSFORM = 0: MFMAs on a stale V image only (what the contraction alone costs); 1: the full loop
template <int TRANSFORM>
__global__ __launch_bounds__(256, 1) void winograd_kernel(const uint4* __restrict__ A, const uint4* __restrict__ B, float* __restrict__ out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const V = smem + 2 * STAGE_W;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint4* Ab = A + (size_t)blockIdx.x * chunks * (A_BYTES / 16);
    auto stage = [&](int c, int buf) {
        unsigned char* const Sa = smem + buf * STAGE_W;
        const uint4* ac = Ab + (size_t)c * (A_BYTES / 16);
#pragma unroll
        for (int k = 0; k < 5; ++k) GLDS16(ac + (k * 4 + wave) * 64 + lane, Sa + (k * 4 + wave) * 1024);
        const uint4* bc = B + (size_t)(c & 15) * (BW_BYTES / 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) GLDS16(bc + k * 256 + tid, Sa + A_BYTES + (k * 256 + wave * 64) * 16);
    };
    f32x16 acc[4][4];                          // [position of this wave][32-tile group]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transform item of this thread: tile tt = tid >> 1 (16 x 8 tiles of the 32 x 16 output tile), channel half ch = tid & 1
    const int tt = tid >> 1, chh = tid & 1;
    const int tx2 = (tt & 15) * 2, ty2 = (tt >> 4) * 2;            // top-left patch pixel of the 4x4 input tile
    auto transform = [&](int buf) {
        const unsigned char* Sa = smem + buf * STAGE_W;
        f16x8 d[4][4];
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int px = tx2 + x;
                d[y][x] = *reinterpret_cast<const f16x8*>(Sa + ((ty2 + y) * PW + px) * 32 + ((((px >> 3) & 1) ^ chh) << 4));
            }
        // B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows then columns, packed fp16
        f16x8 t[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            t[0][x] = d[0][x] - d[2][x]; t[1][x] = d[1][x] + d[2][x]; t[2][x] = d[2][x] - d[1][x]; t[3][x] = d[1][x] - d[3][x];
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const f16x8 v0 = t[y][0] - t[y][2], v1 = t[y][1] + t[y][2], v2 = t[y][2] - t[y][1], v3 = t[y][1] - t[y][3];
            // V[position y*4 + x][tile][half]: the half placed like the A image's column key so that the MFMA reads are conflict-free
            unsigned char* const vp = V + (size_t)(y * 4) * (128 * 32) + tt * 32 + ((((tt >> 3) & 1) ^ chh) << 4);
            *reinterpret_cast<f16x8*>(vp) = v0;
            *reinterpret_cast<f16x8*>(vp + 128 * 32) = v1;
            *reinterpret_cast<f16x8*>(vp + 2 * 128 * 32) = v2;
            *reinterpret_cast<f16x8*>(vp + 3 * 128 * 32) = v3;
        }
    };
    int vj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tile = j * 32 + l31;
        vj[j] = tile * 32 + ((((tile >> 3) & 1) ^ hh) << 4);
    }
    auto compute = [&](int buf) {
        const unsigned char* Sb = smem + buf * STAGE_W + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(vj[j]));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pos = wave * 4 + i;
            const f16x8 wf = *reinterpret_cast<const f16x8*>(Sb + ((pos * 2 + hh) * 32 + l31) * 16);
            f16x8 xa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[j] = *reinterpret_cast<const f16x8*>(V + (size_t)pos * (128 * 32) + vj[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[j], acc[i][j], 0, 0, 0);
        }
    };
    stage(0, 0);
    for (int c = 0; c < chunks; ++c) {
        __syncthreads();                               // chunk c landed; every wave is done with V and with stage (c+1)&1
        if (c + 1 < chunks) stage(c + 1, (c + 1) & 1);
        if (TRANSFORM) {
            transform(c & 1);
            __syncthreads();                           // V complete
        }
        compute(c & 1);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][9];
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}

template <typename K>
static double run(K kern, int blocks_per_cu, size_t lds, const uint4* A, const uint4* B, float* out, int chunks, const char* label, double couts) {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = ncu * blocks_per_cu * 4;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, B, out, chunks);
    hipEventRecord(t0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, B, out, chunks);
    hipEventRecord(t1, 0);
    hipEventSynchronize(t1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, t0, t1);
    ms /= reps;
    // direct-equivalent work of a block and chunk: 512 output pixels x couts x 16 channels x 9 taps
    const double flops = (double)grid * chunks * 512.0 * couts * 16 * 9 * 2.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-64s %8.1f us  %8.1f TFLOP/s direct-equivalent (%s)\n", label, ms * 1e3, tf, hipGetErrorString(hipGetLastError()));
    hipEventDestroy(t0); hipEventDestroy(t1);
    return tf;
}

int main() {
    const int chunks = 32;
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t a_bytes = (size_t)ncu * 2 * 4 * chunks * A_BYTES;
    uint4 *A = nullptr, *B = nullptr;
    float* out = nullptr;
    if (hipMalloc((void**)&A, a_bytes) != hipSuccess || hipMalloc((void**)&B, 16 * BD_BYTES) != hipSuccess ||
        hipMalloc((void**)&out, (size_t)ncu * 8 * 256 * sizeof(float)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A, 0x38, a_bytes);      // small normal halfs: non-trivial operands
    hipMemset(B, 0x34, 16 * BD_BYTES);
    for (int rep = 0; rep < 2; ++rep) {
        const double d = run(direct_kernel, 2, 2 * STAGE_D, A, B, out, chunks, "direct 3x3 (conv3 loop: 512 px x 64 couts, 2 blocks / CU)", 64);
        const double w0 = run(winograd_kernel<0>, 1, 2 * STAGE_W + V_BYTES, A, B, out, chunks, "winograd contraction only (no transform; 512 px x 32 couts)", 32);
        const double w1 = run(winograd_kernel<1>, 1, 2 * STAGE_W + V_BYTES, A, B, out, chunks, "winograd F(2x2,3x3) + in-LDS input transform (512 px x 32 couts)", 32);
        printf("winograd / direct = %.2f (contraction alone: %.2f); output transform not included\n", w1 / d, w0 / d);
    }
    hipDeviceSynchronize();
    return 0;
}
