// Diagnostic micro-benchmark (not product code): does a 16-byte-per-lane state layout ([group of 4 words][game], dwordx4
// accesses) shorten the load / store phases of a launch shaped like the step kernel — 64k lanes, one wave per SIMD, all
// waves loading at once, ~2.5 us of dependent ALU work, all waves storing at once — against the [word][game] layout with
// dword accesses?  (profiles/floor_micro.hip cannot tell: without the ALU phase everything hides under the ~2.8 us floor
// of back-to-back launches.)   hipcc --offload-arch=gfx950 -O3 -o layout_micro layout_micro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t churn(uint32_t v, int iters) {
    for (int k = 0; k < iters; k++) v = (v * 1664525u + 1013904223u) ^ (v >> 7);     // dependent chain
    return v;
}

template <int W>
__global__ __launch_bounds__(256) void k_dword(uint32_t* s, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v[W], acc = 0;
#pragma unroll
    for (int w = 0; w < W; w++) { v[w] = __builtin_nontemporal_load(&s[(size_t)w * n + i]); }
#pragma unroll
    for (int w = 0; w < W; w++) acc ^= v[w];
    acc = churn(acc, iters);
#pragma unroll
    for (int w = 0; w < W; w++) __builtin_nontemporal_store(v[w] + acc, &s[(size_t)w * n + i]);
}

template <int G>
__global__ __launch_bounds__(256) void k_x4(u32x4* s, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 v[G];
    uint32_t acc = 0;
#pragma unroll
    for (int g = 0; g < G; g++) v[g] = __builtin_nontemporal_load(&s[(size_t)g * n + i]);
#pragma unroll
    for (int g = 0; g < G; g++) acc ^= v[g].x ^ v[g].y ^ v[g].z ^ v[g].w;
    acc = churn(acc, iters);
#pragma unroll
    for (int g = 0; g < G; g++) { u32x4 o = v[g]; o.x += acc; o.y += acc; o.z += acc; o.w += acc; __builtin_nontemporal_store(o, &s[(size_t)g * n + i]); }
}

template <int G>
__global__ __launch_bounds__(256) void k_x2(u32x2* s, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    u32x2 v[G];
    uint32_t acc = 0;
#pragma unroll
    for (int g = 0; g < G; g++) v[g] = __builtin_nontemporal_load(&s[(size_t)g * n + i]);
#pragma unroll
    for (int g = 0; g < G; g++) acc ^= v[g].x ^ v[g].y;
    acc = churn(acc, iters);
#pragma unroll
    for (int g = 0; g < G; g++) { u32x2 o = v[g]; o.x += acc; o.y += acc; __builtin_nontemporal_store(o, &s[(size_t)g * n + i]); }
}

template <typename F>
static float time_launches(hipStream_t st, int reps, F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 64; i++) launch();
    (void)hipEventRecord(a, st);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const int n = 65536;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t* s; CK(hipMalloc((void**)&s, (size_t)64 * n * 4)); CK(hipMemset(s, 1, (size_t)64 * n * 4));
    dim3 g(n / 256), b(256);
    for (int iters : {0, 300, 600, 900}) {
        float d = time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_dword<28>), g, b, 0, st, s, n, iters); });
        float x2 = time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_x2<14>), g, b, 0, st, (u32x2*)s, n, iters); });
        float x4 = time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_x4<7>), g, b, 0, st, (u32x4*)s, n, iters); });
        float none = time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_dword<1>), g, b, 0, st, s, n, iters); });
        printf("ALU iters %4d: 1 dword (ALU only) %.2f us | 28 x dword %.2f us | 14 x dwordx2 %.2f us | 7 x dwordx4 %.2f us\n", iters, none, d, x2, x4);
    }
    return 0;
}
