// Probe: does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) write ZEROS for lanes whose offset fails
// the descriptor's range check?  conv's halo slots rely on it (every DMA piece unconditional: exact vmcnt counts, no zero fill).
// Also times buffer-descriptor LDS-DMA against global_load_lds on a streaming read.
// Build: hipcc --offload-arch=gfx950 -O3 buf_lds_oob.hip -o buf_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define GLDS16(gptr, lptr)                                                                   \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

__global__ void probe(const unsigned char* x, unsigned* out, unsigned nrec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) ((unsigned*)smem)[i] = 0xABABABABu;      // poison 16 KiB
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nrec, 0x00027000);
    // even lanes: in range; lanes = 1 mod 4: far out of range; lanes = 3 mod 4: straddle the end (offset + 16 > nrec)
    unsigned voff = (lane & 1) ? ((lane & 2) ? nrec - 8u : 0xFFFFFFF0u) : (unsigned)(wave * 64 + lane) * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)(smem + wave * 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = ((unsigned*)smem)[i];
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void stream(const unsigned char* x, unsigned* out, size_t bytes_per_block, unsigned nrec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* base = x + (size_t)blockIdx.x * bytes_per_block;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00027000);
    const int iters = (int)(bytes_per_block / (16 * 1024));
    for (int it = 0; it < iters; ++it) {
        unsigned char* dst = smem + (it & 3) * 16384;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned off = (unsigned)it * 16384u + (unsigned)((k * 4 + wave) * 64 + lane) * 16u;
            if (MODE == 0) GLDS16(base + off, dst + (k * 4 + wave) * 1024);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)(dst + (k * 4 + wave) * 1024), 16, off, 0, 0, 0);
        }
        if ((it & 3) == 3) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = ((unsigned*)smem)[lane];
}

int main() {
    const size_t N = 1u << 30;
    unsigned char* x; unsigned* out;
    hipMalloc(&x, N); hipMalloc(&out, 1 << 20);
    std::vector<unsigned char> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)(i * 7 + 1) | 1;      // never zero
    hipMemcpy(x, h.data(), h.size(), hipMemcpyHostToDevice);
    const unsigned nrec = 4096 * 4;       // 16 KiB window
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 65536, 0, x, out, nrec);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost);
    int in_ok = 0, oob_zero = 0, oob_poison = 0, oob_other = 0, straddle_zero = 0, straddle_other = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
                const unsigned v = r[(w * 64 + l) * 4 + d];
                if (!(l & 1)) { unsigned e; memcpy(&e, &h[((w * 64 + l) * 16) + d * 4], 4); in_ok += (v == e); }
                else if (l & 2) { if (v == 0) ++straddle_zero; else ++straddle_other; }
                else { if (v == 0) ++oob_zero; else if (v == 0xABABABABu) ++oob_poison; else ++oob_other; }
            }
    printf("in-range dwords ok %d/512; far-OOB lanes: zero %d poison(untouched) %d other %d of 256; straddling lanes: zero %d other %d of 256\n",
           in_ok, oob_zero, oob_poison, oob_other, straddle_zero, straddle_other);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512; const size_t bpb = N / blocks;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(256), 65536, 0, x, out, bpb, (unsigned)bpb);
            else hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(256), 65536, 0, x, out, bpb, (unsigned)bpb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s LDS-DMA stream: %.1f us, %.2f TB/s\n", mode ? "buffer_load..lds" : "global_load_lds ", ms * 1e3, N / (ms * 1e-3) / 1e12);
        }
    return 0;
}
