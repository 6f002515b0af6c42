// Diagnostic micro-benchmark (not product code): cost of back-to-back trivial launches by grid shape, and via hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty(unsigned* p) { if (p == nullptr) p[0] = 1; }
__global__ void k_touch(unsigned* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* s; CK(hipMalloc((void**)&s, 65536 * 4 * 64)); CK(hipMemset(s, 0, 65536 * 4 * 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 4096;
    int shapes[][2] = {{256, 256}, {1024, 64}, {64, 1024}, {128, 512}, {512, 128}};
    for (auto& sh : shapes) {
        for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), 0, st, s);
        CK(hipEventRecord(a, st));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), 0, st, s);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("empty  %4d blocks x %4d threads: %.2f us per launch\n", sh[0], sh[1], ms * 1e3f / reps);
        CK(hipEventRecord(a, st));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_touch, dim3(sh[0]), dim3(sh[1]), 0, st, s, 65536);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("touch  %4d blocks x %4d threads: %.2f us per launch\n", sh[0], sh[1], ms * 1e3f / reps);
    }
    // hipGraph with 64 kernel nodes, replayed
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, s, 65536);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 4; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < 64; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("graph of 64 touch kernels (256x256): %.2f us per kernel\n", ms * 1e3f / (64 * 64));
    return 0;
}
