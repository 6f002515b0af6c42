// Host cost of a kernel launch on this box: empty kernel with a 256-byte by-value argument, 1024 x 64 threads.
//   hipcc --offload-arch=gfx950 -O2 -o profiles/_ab/launch_cost profiles/micro/launch_cost.hip && profiles/_ab/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { unsigned long long w[32]; };
__global__ void k_empty(Args a) { if (a.w[0] == 0xdeadbeefULL) __builtin_trap(); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s[2];
    for (auto& st : s) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Args a = {};
    const int N = 4000;
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            const double t0 = now();
            for (int i = 0; i < N; i++) {
                a.w[1] = i;
                hipStream_t st = (mode & 1) ? s[i & 1] : s[0];
                if (mode < 2) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, st, a);
                else {
                    size_t sz = sizeof a;
                    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
                    hipFunction_t f; 
                    static hipFunction_t cached = nullptr;
                    if (!cached) hipGetFuncBySymbol(&cached, (const void*)k_empty);
                    f = cached;
                    hipModuleLaunchKernel(f, 1024, 1, 1, 64, 1, 1, 0, st, nullptr, cfg);
                }
            }
            const double t1 = now();
            hipDeviceSynchronize();
            const double t2 = now();
            printf("%s, %s: host %.2f us per launch, until drained %.2f us per launch\n", mode < 2 ? "hipLaunchKernelGGL" : "hipModuleLaunchKernel(extra)",
                   (mode & 1) ? "two streams alternating" : "one stream", (t1 - t0) * 1e6 / N, (t2 - t0) * 1e6 / N);
        }
    }
    return 0;
}
