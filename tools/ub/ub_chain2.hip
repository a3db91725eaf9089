// Producer -> consumer launch chains on FRESH data (each kernel reads what the previous one wrote, from other CUs / XCDs):
// the floor of a small dependent kernel inside a batch-1 UNet forward.  Graph replay (host out of the picture).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// y[i] = f(x[perm(i)]): block b reads what block (b + shift) % grid wrote (other CU, other XCD when shift % 8 != 0)
template <int DEPTH, int VEC>
__global__ void k_pp(const float4* __restrict__ x, float4* __restrict__ y, const int* __restrict__ zero, int shift) {
    const int b = (blockIdx.x + shift) % gridDim.x;
    int base = (b * blockDim.x + threadIdx.x) * VEC;
    int j = 0;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) j = zero[(base + j) & 1023];   // dependent loads of a constant table (L2-warm after the first launch)
    float4 v[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) v[u] = x[base + u + j];
    const int ob = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
#pragma unroll
    for (int u = 0; u < VEC; ++u) { v[u].x += 1.f; y[ob + u] = v[u]; }
}
// two-phase kernel: phase 1 reads a small "partials" array written by the previous kernel (fresh), reduces through LDS, phase 2 reads x
template <int VEC>
__global__ void k_fin(const float4* __restrict__ x, float4* __restrict__ y, const float* __restrict__ part, float* __restrict__ part_out) {
    __shared__ float s[256];
    s[threadIdx.x] = part[threadIdx.x];
    __syncthreads();
    float a = 0.f;
    if (threadIdx.x < 32) { for (int k = 0; k < 8; ++k) a += s[threadIdx.x * 8 + k]; s[threadIdx.x] = a; }
    __syncthreads();
    const float m = s[threadIdx.x & 31];
    const int ob = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
#pragma unroll
    for (int u = 0; u < VEC; ++u) { float4 v = x[ob + u]; v.x += m; y[ob + u] = v; }
    if (blockIdx.x == 0) part_out[threadIdx.x] = m * 0.5f;
}

int main() {
    const size_t n4 = 1 << 22;     // 64 MB per buffer
    float4 *a, *b; int* zero; float *p0, *p1;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&zero, 4096)); CK(hipMalloc(&p0, 1024)); CK(hipMalloc(&p1, 1024));
    CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16)); CK(hipMemset(zero, 0, 4096)); CK(hipMemset(p0, 0, 1024)); CK(hipMemset(p1, 0, 1024));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto graphit = [&](const char* name, auto launch_pair) {
        hipGraph_t g; hipGraphExec_t ge;
        const int K = 250;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < K; ++i) launch_pair();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 4; ++r) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %7.2f us per kernel\n", name, ms * 1e3 / (4 * K * 2));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
#define PP(name, D, V, G, SH) graphit(name, [&]() { k_pp<D, V><<<G, 256, 0, s>>>(a, b, zero, SH); k_pp<D, V><<<G, 256, 0, s>>>(b, a, zero, SH); })
    PP("ping-pong 16 B/thread, 32 WG (128 KB), same block", 1, 1, 32, 0);
    PP("ping-pong 16 B/thread, 32 WG (128 KB), other XCD", 1, 1, 32, 3);
    PP("ping-pong 16 B/thread, 256 WG (1 MB), same block", 1, 1, 256, 0);
    PP("ping-pong 16 B/thread, 256 WG (1 MB), other XCD", 1, 1, 256, 3);
    PP("ping-pong 64 B/thread, 256 WG (4 MB), other XCD", 1, 4, 256, 3);
    PP("ping-pong 64 B/thread, 1024 WG (16 MB), other XCD", 1, 4, 1024, 3);
    PP("ping-pong 64 B/thread, 2048 WG (32 MB), other XCD", 1, 4, 2048, 3);
    PP("2 dependent (1 warm table + fresh), 256 WG (1 MB), other XCD", 2, 1, 256, 3);
    PP("3 dependent (2 warm table + fresh), 256 WG (1 MB), other XCD", 3, 1, 256, 3);
    graphit("partials (fresh) -> LDS reduce -> x (fresh), 256 WG (1 MB)", [&]() { k_fin<1><<<256, 256, 0, s>>>(a, b, p0, p1); k_fin<1><<<256, 256, 0, s>>>(b, a, p1, p0); });
    graphit("partials (fresh) -> LDS reduce -> x (fresh), 32 WG (128 KB)", [&]() { k_fin<1><<<32, 256, 0, s>>>(a, b, p0, p1); k_fin<1><<<32, 256, 0, s>>>(b, a, p1, p0); });
    return 0;
}
