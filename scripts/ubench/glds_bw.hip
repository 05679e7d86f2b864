// Microbenchmark: global_load_lds_dwordx4 staging throughput per CU on gfx950 for the access
// patterns a conv A-patch / weight slab produces.  Build: hipcc --offload-arch=gfx950 -O3 glds_bw.hip -o glds_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define GLDS16(gptr, lptr)                                                                   \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

// each iteration: K wave-instructions per wave (K KiB per wave, 4K KiB per block) then vmcnt(0)+barrier.
// lpp = lanes per "pixel" (contiguous lpp*16 B), pixstride = bytes between pixels, footprint = bytes the
// block cycles through (power of two), blocks share the footprint region when shared != 0.
template <int K>
__global__ __launch_bounds__(256) void k_glds(const unsigned char* __restrict__ src, int iters, int lpp, int pixstride,
                                              size_t footprint, int shared, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const size_t blockbase = shared ? 0 : (size_t)blockIdx.x * footprint;
    const size_t mask = footprint - 1;
    size_t pos = (size_t)blockIdx.x * 4096 * 7;   // de-phase blocks inside a shared footprint
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int item = k * 256 + tid;
            const size_t off = (pos + (size_t)(item / lpp) * pixstride + (size_t)(item % lpp) * 16) & mask;
            GLDS16(src + blockbase + off, smem + ((it & 1) * K * 4 + k * 4 + wave) * 1024);
        }
        pos += (size_t)(K * 256 / lpp) * pixstride;
        __syncthreads();
    }
    if (sink && tid == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 16);
}

// same traffic through registers (global_load_dwordx4 + ds_write_b128)
template <int K>
__global__ __launch_bounds__(256) void k_reg(const unsigned char* __restrict__ src, int iters, int lpp, int pixstride,
                                             size_t footprint, int shared, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const size_t blockbase = shared ? 0 : (size_t)blockIdx.x * footprint;
    const size_t mask = footprint - 1;
    size_t pos = (size_t)blockIdx.x * 4096 * 7;
    for (int it = 0; it < iters; ++it) {
        uint4 r[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int item = k * 256 + tid;
            const size_t off = (pos + (size_t)(item / lpp) * pixstride + (size_t)(item % lpp) * 16) & mask;
            r[k] = *reinterpret_cast<const uint4*>(src + blockbase + off);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) *reinterpret_cast<uint4*>(smem + ((it & 1) * K * 256 + k * 256 + tid) * 16) = r[k];
        pos += (size_t)(K * 256 / lpp) * pixstride;
        __syncthreads();
    }
    if (sink && tid == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 16);
}

template <typename F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const size_t total = (size_t)4 << 30;
    unsigned char* src;
    unsigned* sink;
    if (hipMalloc(&src, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, total);
    hipMalloc(&sink, 1 << 20);
    const int iters = 64;
    constexpr int K = 8;   // 8 KiB per wave, 32 KiB per block per iteration
    hipFuncSetAttribute((const void*)k_glds<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_reg<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Pat { const char* name; int lpp, pixstride; };
    const Pat pats[] = {{"linear 1KiB/instr", 64, 1024}, {"8 lanes x 16B @128B (full line)", 8, 128},
                        {"4 lanes x 16B @128B (half line)", 4, 128}, {"2 lanes x 16B @128B", 2, 128},
                        {"1 lane x 16B @128B", 1, 128}, {"1 lane x 16B @256B", 1, 256}, {"4 lanes x 16B @256B", 4, 256}};
    struct Foot { const char* name; size_t bytes; int shared; };
    const Foot foots[] = {{"L2-hot (256 KiB shared)", (size_t)256 << 10, 1}, {"L2/MALL (64 MiB shared)", (size_t)64 << 20, 1},
                          {"HBM (4 MiB per block, private)", (size_t)4 << 20, 0}};
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        const int nblk = 256 * bpc;
        const size_t lds = (size_t)2 * K * 4 * 1024;   // 64 KiB -> 2 blocks/CU; 4/CU runs as two rounds
        printf("== %d blocks per CU (%d blocks), %d KiB per block per barrier, lds %zu\n", bpc, nblk, K * 4, lds);
        for (const Foot& f : foots)
            for (const Pat& p : pats) {
                if (!f.shared && (size_t)nblk * f.bytes > total) continue;
                float ms_g = time_ms([&] { hipLaunchKernelGGL(k_glds<K>, dim3(nblk), dim3(256), lds, 0, src, iters, p.lpp, p.pixstride, f.bytes, f.shared, sink); }, 5);
                float ms_r = time_ms([&] { hipLaunchKernelGGL(k_reg<K>, dim3(nblk), dim3(256), lds, 0, src, iters, p.lpp, p.pixstride, f.bytes, f.shared, sink); }, 5);
                const double bytes = (double)nblk * iters * K * 4 * 1024;
                printf("%-32s %-34s glds %7.1f us %6.2f TB/s (%5.1f GB/s/CU) | reg+ds_write %7.1f us %6.2f TB/s\n", f.name, p.name,
                       ms_g * 1e3, bytes / ms_g / 1e9, bytes / ms_g / 1e6 / 256, ms_r * 1e3, bytes / ms_r / 1e9);
            }
    }
    return 0;
}
