// Dependent-launch floor micro-benchmark: chains of small kernels on one stream (what a batch-1 UNet forward is made of).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_store(float* y) { y[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }
// depth dependent loads: each load's address depends on the previous value (all zeros) -> serial round trips
template <int DEPTH>
__global__ void k_chain(const int* __restrict__ idx, const float* __restrict__ x, float* __restrict__ y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = i;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) j = idx[j] + i;        // idx is all zeros: j stays i, but the compiler cannot know
    y[i] = x[j] * 2.f;
}
// same work, but the kernel is padded with NOPs that execute (I-cache footprint)
template <int KB>
__global__ void k_fat(const float* __restrict__ x, float* __restrict__ y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = x[i];
#pragma unroll
    for (int r = 0; r < KB * 64; ++r) asm volatile("s_nop 0");   // 4 bytes each -> KB * 256 B ... (KB*64*4 = KB*256 bytes)
    y[i] = v * 2.f;
}

int main() {
    const int n = 1 << 20;
    float *x, *y; int* idx;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&idx, n * 4));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(idx, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 50; ++i) launch();
        hipStreamSynchronize(s);
        const int K = 2000;
        hipEventRecord(e0, s);
        for (int i = 0; i < K; ++i) launch();
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-48s %7.2f us per launch\n", name, ms * 1e3 / K);
    };
    // graph replay of the same chains: removes the host from the picture
    auto graphit = [&](const char* name, auto launch) {
        hipGraph_t g; hipGraphExec_t ge;
        const int K = 500;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < K; ++i) launch();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 4; ++r) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-48s %7.2f us per launch (graph)\n", name, ms * 1e3 / (4 * K));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
#define BOTH(name, ...) do { auto f = [&]() { __VA_ARGS__; }; timeit(name, f); graphit(name, f); } while (0)
    BOTH("empty, 1 WG", k_empty<<<1, 64, 0, s>>>());
    BOTH("empty, 256 WG x 256", k_empty<<<256, 256, 0, s>>>());
    BOTH("empty, 2048 WG x 256", k_empty<<<2048, 256, 0, s>>>());
    BOTH("store only, 32 WG x 256", k_store<<<32, 256, 0, s>>>(y));
    BOTH("store only, 256 WG x 256", k_store<<<256, 256, 0, s>>>(y));
    BOTH("1 load + store, 32 WG", k_chain<1><<<32, 256, 0, s>>>(idx, x, y, n));
    BOTH("1 load + store, 256 WG", k_chain<1><<<256, 256, 0, s>>>(idx, x, y, n));
    BOTH("2 dependent loads + store, 32 WG", k_chain<2><<<32, 256, 0, s>>>(idx, x, y, n));
    BOTH("3 dependent loads + store, 32 WG", k_chain<3><<<32, 256, 0, s>>>(idx, x, y, n));
    BOTH("5 dependent loads + store, 32 WG", k_chain<5><<<32, 256, 0, s>>>(idx, x, y, n));
    BOTH("1 load + store + 4 KB of code, 256 WG", k_fat<16><<<256, 256, 0, s>>>(x, y));
    BOTH("1 load + store + 32 KB of code, 256 WG", k_fat<128><<<256, 256, 0, s>>>(x, y));
    BOTH("1 load + store + 128 KB of code, 256 WG", k_fat<512><<<256, 256, 0, s>>>(x, y));
    // alternating two different kernels (I-cache / code object switch)
    BOTH("alternate store / chain<1>, 256 WG", k_store<<<256, 256, 0, s>>>(y); k_chain<1><<<256, 256, 0, s>>>(idx, x, y, n));
    return 0;
}
