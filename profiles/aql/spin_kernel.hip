// device code of profiles/aql/aql_probe.cpp (hipcc --genco --offload-arch=gfx950 -> spin_kernel.hsaco)
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(64) void k_spin(unsigned long long ticks, unsigned int* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, 1u);
}
