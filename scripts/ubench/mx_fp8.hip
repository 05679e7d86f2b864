// Microbenchmark: what would the MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the MAC rate of every other
// fp8 / fp16 MFMA on gfx950) buy conv3's main loop?
//
// Same loop skeleton as ring_depth.hip ("conv3 today": two LDS stages, two blocks per CU, per 16-"unit" chunk a 20-KiB A patch
// + an 18-KiB weight slab by LDS-DMA, 9 taps x (2 weight + 4 pixel fragments) per wave), three operand flavours that move the
// SAME bytes through the DMA and the LDS reads:
//   f16      fragment = 16 B = 8 halfs         -> 1 x v_mfma_f32_32x32x16_f16          per (weight, pixel) pair and tap
//   fp8      fragment = 16 B = 16 e4m3 bytes   -> 2 x v_mfma_f32_32x32x16_fp8_fp8      (what conv3<Q=1> does today)
//   fp8-mx   fragment = 2 x 16 B = 32 e4m3     -> 1 x v_mfma_scale_f32_32x32x64_f8f6f4 (unit E8M0 scales) per TWO chunks' bytes
// Output: useful MAC rate of each flavour (TFLOP/s, counting 2 flops per MAC) and the ratio, on non-trivial data.
// Build: hipcc --offload-arch=gfx950 -O3 mx_fp8.hip -o mx_fp8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef long i64x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define GLDS16(gptr, lptr)                                                                   \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

constexpr int A_BYTES = 20480, B_BYTES = 18432, STAGE = A_BYTES + B_BYTES;

// MODE 0 f16, 1 fp8 (non-scaled), 2 fp8 MX (consumes two consecutive stages' worth of k per MFMA: the loop pairs chunks)
template <int MODE>
__global__ __launch_bounds__(256, 2) void loop_kernel(const uint4* __restrict__ A, const uint4* __restrict__ B, float* __restrict__ out,
                                                      int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint4* Ab = A + (size_t)blockIdx.x * chunks * (A_BYTES / 16);
    const int nB = (wave < 2) ? 5 : 4;
    auto stage = [&](int c, int buf) {
        unsigned char* const Sa = smem + buf * STAGE;
        const uint4* ac = Ab + (size_t)c * (A_BYTES / 16);
#pragma unroll
        for (int k = 0; k < 5; ++k) GLDS16(ac + (k * 4 + wave) * 64 + lane, Sa + (k * 4 + wave) * 1024);
        const uint4* bc = B + (size_t)(c & 15) * (B_BYTES / 16);
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
        for (int dx = 0; dx < 3; ++dx) aj[j][dx] = (ty * 34 + tx + dx) * 32 + (((((tx + dx) >> 3) & 1) ^ hh) << 4);
    }
    auto compute = [&](int buf) {
        const unsigned char* Sa = smem + buf * STAGE;
        const unsigned char* Sb = Sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) asm volatile("" : "+v"(aj[j][dx]));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned char* Ar = Sa + (t / 3) * (34 * 32);
            if constexpr (MODE == 2) {
                // 32 bytes per operand: both 16-byte halves of the pixel / weight cell (the second half sits 16 B further)
                i32x8 wf[2], xa[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(Sb + ((((i * 9 + t) * 2 + 0) * 32) + l31) * 16);
                    const i32x4 hi = *reinterpret_cast<const i32x4*>(Sb + ((((i * 9 + t) * 2 + 1) * 32) + l31) * 16);
                    wf[i] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(Ar + aj[j][t % 3]);
                    const i32x4 hi = *reinterpret_cast<const i32x4*>(Ar + (aj[j][t % 3] ^ 16));
                    xa[j] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], xa[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else {
                f16x8 wf[2], xa[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const f16x8*>(Sb + ((((i * 9 + t) * 2 + hh) * 32) + l31) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ar + aj[j][t % 3]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (MODE == 0) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xa[j], acc[i][j], 0, 0, 0);
                        } else {
                            const i64x2 wq = __builtin_bit_cast(i64x2, wf[i]);
                            const i64x2 xq = __builtin_bit_cast(i64x2, xa[j]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wq[0], xq[0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wq[1], xq[1], acc[i][j], 0, 0, 0);
                        }
                    }
            }
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

template <int MODE>
static double run(const uint4* A, const uint4* B, float* out, int chunks, const char* label, double macs_per_chunk_wave) {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = ncu * 2 * 4;
    const size_t lds = 2 * STAGE;
    hipFuncSetAttribute((const void*)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(grid), dim3(256), lds, 0, A, B, out, chunks);
    hipEventRecord(t0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((loop_kernel<MODE>), dim3(grid), dim3(256), lds, 0, A, B, out, chunks);
    hipEventRecord(t1, 0);
    hipEventSynchronize(t1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, t0, t1);
    ms /= reps;
    const double flops = (double)grid * chunks * 4 * macs_per_chunk_wave * 2.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-44s %8.1f us  %8.1f TFLOP/s  DMA %5.2f TB/s (%s)\n", label, ms * 1e3, tf, (double)grid * chunks * STAGE / (ms * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
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
    if (hipMalloc((void**)&A, a_bytes) != hipSuccess || hipMalloc((void**)&B, 16 * B_BYTES) != hipSuccess ||
        hipMalloc((void**)&out, (size_t)ncu * 8 * 256 * sizeof(float)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A, 0x38, a_bytes);      // 0x38 bytes: 0.5 as e4m3 pairs / a small normal half pattern -- non-trivial operands for every flavour
    hipMemset(B, 0x34, 16 * B_BYTES);
    // MACs per chunk and wave: 9 taps x 8 (weight, pixel) pairs x 32x32 outputs x k; k = 16 (f16), 32 (fp8: two k16 MFMAs), 64 (MX)
    const double f16 = run<0>(A, B, out, chunks, "f16   32x32x16          (8 MFMA / tap)", 9 * 8 * 1024.0 * 16);
    const double f8 = run<1>(A, B, out, chunks, "fp8   32x32x16 x2       (16 MFMA / tap)", 9 * 8 * 1024.0 * 32);
    const double mx = run<2>(A, B, out, chunks, "fp8   MX 32x32x64       (8 MFMA / tap, 2x LDS reads)", 9 * 8 * 1024.0 * 64);
    printf("fp8 / f16 = %.2f   fp8-MX / f16 = %.2f   fp8-MX / fp8 = %.2f   (same DMA bytes per chunk; MX reads both 16-B halves of a cell)\n",
           f8 / f16, mx / f16, mx / f8);
    hipDeviceSynchronize();
    return 0;
}
