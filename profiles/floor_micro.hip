// Diagnostic micro-benchmark (not product code): what does one launch over 64k lanes cost on MI355X
// for different ways of moving ~100 B of state per lane in and out?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(uint32_t* p) { if (p == nullptr) p[0] = 1; }

template <int W, bool NT>
__global__ __launch_bounds__(256) void k_dword(uint32_t* s, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v[W];
#pragma unroll
    for (int w = 0; w < W; w++) v[w] = NT ? __builtin_nontemporal_load(&s[(size_t)w * n + i]) : s[(size_t)w * n + i];
#pragma unroll
    for (int w = 0; w < W; w++) {
        uint32_t o = v[w] + v[(w + 1) % W];
        if (NT) __builtin_nontemporal_store(o, &s[(size_t)w * n + i]); else s[(size_t)w * n + i] = o;
    }
}

template <int W4, bool NT>
__global__ __launch_bounds__(256) void k_x4(u32x4* s, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 v[W4];
#pragma unroll
    for (int w = 0; w < W4; w++) v[w] = NT ? __builtin_nontemporal_load(&s[(size_t)w * n + i]) : s[(size_t)w * n + i];
#pragma unroll
    for (int w = 0; w < W4; w++) {
        u32x4 o = v[w]; o.x += v[(w + 1) % W4].y;
        if (NT) __builtin_nontemporal_store(o, &s[(size_t)w * n + i]); else s[(size_t)w * n + i] = o;
    }
}

template <int W, bool NT>
__global__ __launch_bounds__(256) void k_loadonly(uint32_t* s, int n, uint32_t* sink) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int w = 0; w < W; w++) acc += NT ? __builtin_nontemporal_load(&s[(size_t)w * n + i]) : s[(size_t)w * n + i];
    if (acc == 0x12345678u) sink[i] = acc;
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
    printf("empty kernel                 : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL(k_empty, g, b, 0, st, s); }));
    printf("load-only 26 dwords          : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_loadonly<26, false>), g, b, 0, st, s, n, s); }));
    printf("load-only 26 dwords nt       : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_loadonly<26, true>), g, b, 0, st, s, n, s); }));
    printf("26 dwords ld+st plain        : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_dword<26, false>), g, b, 0, st, s, n); }));
    printf("26 dwords ld+st nt           : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_dword<26, true>), g, b, 0, st, s, n); }));
    printf("20 dwords ld+st nt           : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_dword<20, true>), g, b, 0, st, s, n); }));
    printf("16 dwords ld+st nt           : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_dword<16, true>), g, b, 0, st, s, n); }));
    printf("7 x dwordx4 ld+st plain      : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_x4<7, false>), g, b, 0, st, (u32x4*)s, n); }));
    printf("7 x dwordx4 ld+st nt         : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_x4<7, true>), g, b, 0, st, (u32x4*)s, n); }));
    printf("5 x dwordx4 ld+st nt         : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_x4<5, true>), g, b, 0, st, (u32x4*)s, n); }));
    printf("4 x dwordx4 ld+st nt         : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_x4<4, true>), g, b, 0, st, (u32x4*)s, n); }));
    dim3 g4(n / 64), b4(64);
    printf("7 x dwordx4 ld+st nt, 64-thr : %.2f us\n", time_launches(st, 2048, [&] { hipLaunchKernelGGL((k_x4<7, true>), g4, b4, 0, st, (u32x4*)s, n); }));
    return 0;
}
