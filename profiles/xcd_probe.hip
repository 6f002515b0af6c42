// Probe for the chained launches' hand-off (profiles/r03/xcd_probe.txt).  Standalone: hipcc --offload-arch=gfx950 -O2 xcd_probe.hip
//  A. placement: which XCD does block b of launch E land on, for launches rotating over 3 streams (as the chained rollout does)
//     and for launches on one stream?  (HW_REG_XCC_ID per block.)
//  B. hand-off latency between two workgroups (one wave each) that pass a 27-row x 64-lane state back and forth, as wave w of
//     launch E hands its games to wave w of launch E + 1: payload stores -> s_waitcnt vmcnt(0) -> flag store | sc1 poll of the
//     flag -> sc1 loads of the payload.  Same XCD against different XCDs; payload + flag stored sc1 (written through, today's
//     form) against stored plain (line kept in the XCD's L2; only valid when both are on ONE XCD, which is all it is run for).
// Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t xcc_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
__device__ inline uint32_t hw_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x)); return x; }

__global__ __launch_bounds__(64) void k_place(uint32_t* out, int launch, int grid, int busy_ticks) {
    if (threadIdx.x == 0) out[(size_t)launch * grid + blockIdx.x] = xcc_id() | (hw_id() << 8);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)busy_ticks) __builtin_amdgcn_s_sleep(4);
}

constexpr int ROWS = 27, LANES = 64, STRIDE = 65536;      // state[row][game] as in the library
constexpr int AUX_SC1 = 16;
struct PP {
    uint32_t* payload[2];     // [ROWS * STRIDE] each; the lanes of the workgroup use games [0, 64)
    uint32_t* flag[2];        // one 128-byte line each
    uint32_t* role;           // [2] claimed-by markers
    uint32_t* result;         // [0] ticks (100 MHz), [1] errors, [2] timeouts, [3] xcc of role 0, [4] xcc of role 1, [5]/[6] hw ids
    int want_xcc[2];
    int rounds, plain, diff_cu;
};

__device__ inline void st_row(uint32_t* base, int row, uint32_t lane_off, uint32_t v, bool plain) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, -1, 0x00020000);
    if (plain) __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)lane_off, row * STRIDE * 4, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)lane_off, row * STRIDE * 4, AUX_SC1);
}
__device__ inline uint32_t ld_row(const uint32_t* base, int row, uint32_t lane_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), (short)0, -1, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, row * STRIDE * 4, AUX_SC1);
}
__device__ inline void st_flag(uint32_t* p, uint32_t v, bool plain) {
    if (plain) { *(volatile uint32_t*)p = v; } else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool wait_flag(const uint32_t* p, uint32_t want) {
    for (int spin = 0; spin < (1 << 20); spin++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__global__ __launch_bounds__(64) void k_pingpong(PP a) {
    const uint32_t x = xcc_id(), hw = hw_id(), lane = threadIdx.x;
    __shared__ int s_role;
    if (lane == 0) {
        int role = -1;
        for (int r = 0; r < 2 && role < 0; r++) {
            if ((int)x != a.want_xcc[r]) continue;
            if (r == 1 && a.diff_cu) {                      // role 1 must not share role 0's CU (cu[11:8] sh[12] se[15:13] of HW_ID)
                const uint32_t other = __hip_atomic_load(a.role + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (other == 0 || ((other - 1) & 0xff00) == (hw & 0xff00)) continue;
            }
            uint32_t expect = 0;
            if (__hip_atomic_compare_exchange_strong(a.role + r, &expect, hw + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) role = r;
        }
        s_role = role;
    }
    __syncthreads();
    const int role = s_role;
    if (role < 0) return;
    if (lane == 0) { a.result[3 + role] = x; a.result[5 + role] = hw; }
    // wait for the partner to exist (bounded)
    bool ok = true;
    if (lane == 0) {
        ok = false;
        for (int spin = 0; spin < (1 << 20) && !ok; spin++) { ok = __hip_atomic_load(a.role + (1 - role), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; __builtin_amdgcn_s_sleep(2); }
    }
    ok = __shfl(ok ? 1 : 0, 0) != 0;
    if (!ok) { if (lane == 0) atomicAdd(a.result + 2, 1u); return; }
    const bool plain = a.plain != 0;
    uint32_t errors = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < a.rounds; r++) {
        uint32_t v[ROWS];
        if (role == 1 || r > 0) {
            // my turn comes when the other side's flag says so
            const uint32_t want = role == 1 ? (uint32_t)r + 1 : (uint32_t)r;
            if (!wait_flag(a.flag[1 - role], want)) { if (lane == 0) atomicAdd(a.result + 2, 1u); return; }
            for (int k = 0; k < ROWS; k++) v[k] = ld_row(a.payload[1 - role], k, lane * 4);
            const uint32_t expect_base = (role == 1 ? 2u * r : 2u * r - 1u) * 1000u;
            for (int k = 0; k < ROWS; k++) errors += v[k] != expect_base + k * 64 + lane;
        }
        const uint32_t my_base = (role == 0 ? 2u * r : 2u * r + 1u) * 1000u;
        for (int k = 0; k < ROWS; k++) st_row(a.payload[role], k, lane * 4, my_base + k * 64 + lane, plain);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_flag(a.flag[role], (uint32_t)r + 1, plain);
    }
    const unsigned long long t1 = wall_clock64();
    atomicAdd(a.result + 1, errors);
    if (role == 0 && lane == 0) a.result[0] = (uint32_t)(t1 - t0);
}

static void placement(int n_streams, int launches, int grid, int busy_ticks) {
    std::vector<hipStream_t> st(n_streams);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t* d; CK(hipMalloc(&d, (size_t)launches * grid * 4)); CK(hipMemset(d, 0xff, (size_t)launches * grid * 4));
    CK(hipDeviceSynchronize());
    for (int l = 0; l < launches; l++) hipLaunchKernelGGL(k_place, dim3(grid), dim3(64), 0, st[l % n_streams], d, l, grid, busy_ticks);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h((size_t)launches * grid);
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    printf("placement: %d stream(s), %d launches of %d blocks, each block busy %.1f us\n", n_streams, launches, grid, busy_ticks / 100.0);
    for (int l = 0; l < launches; l++) {
        const uint32_t* r = &h[(size_t)l * grid];
        int per[8] = {0}, rr = 0, same_prev = 0, same_prev_stream = 0;
        for (int b = 0; b < grid; b++) {
            per[r[b] & 7]++;
            rr += (r[b] & 7) == (((r[0] & 7) + b) & 7);
            if (l > 0) same_prev += (r[b] & 7) == (h[(size_t)(l - 1) * grid + b] & 7);
            if (l >= n_streams) same_prev_stream += (r[b] & 7) == (h[(size_t)(l - n_streams) * grid + b] & 7);
        }
        printf("  launch %2d stream %d: block 0 on XCD %u; blocks on XCD (xcd(0) + b) %% 8: %4d of %d; per XCD %d %d %d %d %d %d %d %d; same XCD as launch before: %4d, as %d launches before: %4d\n",
               l, l % n_streams, r[0] & 7, rr, grid, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], same_prev, n_streams, same_prev_stream);
    }
    CK(hipFree(d));
    for (auto& s : st) CK(hipStreamDestroy(s));
}

static void pingpong(int xa, int xb, int plain, int diff_cu, int rounds) {
    PP a{};
    for (int r = 0; r < 2; r++) {
        CK(hipMalloc(&a.payload[r], (size_t)ROWS * STRIDE * 4)); CK(hipMemset(a.payload[r], 0, (size_t)ROWS * STRIDE * 4));
        CK(hipMalloc(&a.flag[r], 128)); CK(hipMemset(a.flag[r], 0, 128));
    }
    CK(hipMalloc(&a.role, 128)); CK(hipMemset(a.role, 0, 128));
    CK(hipMalloc(&a.result, 128)); CK(hipMemset(a.result, 0, 128));
    a.want_xcc[0] = xa; a.want_xcc[1] = xb; a.rounds = rounds; a.plain = plain; a.diff_cu = diff_cu;
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_pingpong, dim3(2048), dim3(64), 0, 0, a);
    CK(hipDeviceSynchronize());
    uint32_t res[8]; CK(hipMemcpy(res, a.result, sizeof res, hipMemcpyDeviceToHost));
    printf("  XCD %d -> XCD %d, %s stores, %s: %7.3f us per hand-off (flag seen + 27x64 words loaded + stored + acknowledged), errors %u, timeouts %u  [xcc %u/%u cu %x/%x]\n",
           xa, xb, plain ? "plain" : "sc1  ", diff_cu ? "different CUs" : "any CU       ", res[0] / 100.0 / (2.0 * rounds), res[1], res[2], res[3], res[4], (res[5] >> 8) & 0xff, (res[6] >> 8) & 0xff);
    for (int r = 0; r < 2; r++) { CK(hipFree(a.payload[r])); CK(hipFree(a.flag[r])); }
    CK(hipFree(a.role)); CK(hipFree(a.result));
}

int main() {
    CK(hipSetDevice(0));
    placement(3, 12, 1024, 0);
    placement(3, 12, 1024, 400);
    placement(2, 8, 1024, 400);
    placement(1, 6, 1024, 0);
    placement(3, 6, 1000, 400);      // a grid that is no multiple of 8
    printf("hand-off ping-pong, 2000 rounds each:\n");
    for (int rep = 0; rep < 2; rep++) {
        pingpong(0, 3, 0, 1, 2000);
        pingpong(2, 2, 0, 1, 2000);
        pingpong(2, 2, 1, 1, 2000);
        pingpong(5, 5, 0, 1, 2000);
        pingpong(5, 5, 1, 1, 2000);
    }
    return 0;
}
