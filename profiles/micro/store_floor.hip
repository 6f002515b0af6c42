// What does it cost to WRITE the output of one C4 enumerate call (16 384 boards x 40 placements x 43 B = 28.2 MB) and nothing else?
// Pure 16-byte stores from a grid shaped like k_enumerate's (512 workgroups x 320 threads) and from a memset-shaped grid; back-to-back
// launches, HIP events.  The floor for any kernel that produces this output.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/_ab/store_floor profiles/micro/store_floor.hip && profiles/_ab/store_floor
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void k_fill(u32x4* out, size_t n16, unsigned v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        u32x4 w = {v, v + 1, v + 2, (unsigned)i};
        if (NT) __builtin_nontemporal_store(w, &out[i]); else out[i] = w;
    }
}
__global__ void k_read_fill(const u32x4* in, u32x4* out, size_t n16_in, size_t n16) {     // + the 44 B per board read first
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a = {0, 0, 0, 0};
    if (t < n16_in) a = __builtin_nontemporal_load(&in[t]);
    for (size_t i = t; i < n16; i += stride) { u32x4 w = a; w.x += (unsigned)i; __builtin_nontemporal_store(w, &out[i]); }
}
int main() {
    const size_t bytes = (size_t)16384 * 40 * 43, n16 = bytes / 16, in_bytes = (size_t)16384 * 44, n16_in = in_bytes / 16;
    u32x4 *out, *in;
    hipMalloc(&out, bytes + 64); hipMalloc(&in, in_bytes + 64); hipMemset(in, 1, in_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int grid, block; } shapes[] = {{512, 320}, {1024, 256}, {2048, 256}, {4096, 256}, {6912, 256}};
    for (int variant = 0; variant < 3; variant++)
        for (auto sh : shapes) {
            for (int warm = 0; warm < 2; warm++) {
                hipEventRecord(e0, 0);
                for (int r = 0; r < 200; r++) {
                    if (variant == 0) hipLaunchKernelGGL(k_fill<true>, dim3(sh.grid), dim3(sh.block), 0, 0, out, n16, (unsigned)r);
                    else if (variant == 1) hipLaunchKernelGGL(k_fill<false>, dim3(sh.grid), dim3(sh.block), 0, 0, out, n16, (unsigned)r);
                    else hipLaunchKernelGGL(k_read_fill, dim3(sh.grid), dim3(sh.block), 0, 0, in, out, n16_in, n16);
                }
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-22s grid %5d x %3d: %.2f us per launch = %.2f TB/s\n", variant == 0 ? "nt stores" : variant == 1 ? "plain stores" : "read 44 B/board + nt",
                   sh.grid, sh.block, ms * 1e3 / 200, (double)bytes / (ms * 1e-3 / 200) / 1e12);
        }
    return 0;
}
