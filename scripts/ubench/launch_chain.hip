// What a DEPENDENT launch costs on the GPU: chains of N kernels on one stream - launched one by one, and as ONE hipGraph - timed with
// events around the whole chain (the host runs ahead of the GPU after the first launches, so the figure is the device-side cost per
// link: command-processor dispatch + barrier + the cache write-back / invalidate between dependent kernels + the kernel's own ramp).
// Kernels: an empty one (1 block / 512 blocks) and one that reads what its predecessor wrote (2 MB, 512 blocks x 256 threads x 16 B).
//   hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip && ./launch_chain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(const uint4* __restrict__ in, uint4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v = in[i];
    v.x += 1u;
    out[i] = v;
}

template <typename F>
static int chain(const char* name, int n, int reps, hipStream_t s, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < n; ++i) launch(i);                      // warm-up
    CK(hipStreamSynchronize(s));
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) launch(i);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    printf("%-44s stream : %7.2f us per link (best %7.2f)\n", name, sum / reps / n * 1e3f, best / n * 1e3f);
    // the same chain as one graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    best = 1e30f; sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    printf("%-44s graph  : %7.2f us per link (best %7.2f)\n", name, sum / reps / n * 1e3f, best / n * 1e3f);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t n16 = (size_t)512 * 256;
    uint4 *a, *b;
    CK(hipMalloc((void**)&a, n16 * sizeof(uint4))); CK(hipMalloc((void**)&b, n16 * sizeof(uint4)));
    CK(hipMemset(a, 0, n16 * sizeof(uint4))); CK(hipMemset(b, 0, n16 * sizeof(uint4)));
    const int N = 64, R = 20;
    if (chain("empty kernel, 1 block x 64 threads", N, R, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); })) return 1;
    if (chain("empty kernel, 512 blocks x 256 threads", N, R, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, s); })) return 1;
    if (chain("2 MB read-after-write, 512 blocks x 256", N, R, s, [&](int i) {
            hipLaunchKernelGGL(k_touch, dim3(512), dim3(256), 0, s, (const uint4*)((i & 1) ? b : a), (i & 1) ? a : b); })) return 1;
    if (chain("2 MB read-after-write, 64 blocks x 256 (x8 B)", N, R, s, [&](int i) {
            hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, (const uint4*)((i & 1) ? b : a), (i & 1) ? a : b); })) return 1;
    return 0;
}
