// The general step kernel (all players of a game in one lane) and its per-translation-unit shape table.  Included by
// tetris_hip.hip (one and two players, every other kernel, the C ABI) and by tetris_hip_multi.hip (three and four players:
// 36 more instantiations of this kernel, compiled in parallel with the first file because they double its compile time).
#pragma once
#include <hip/hip_runtime.h>

#include "tetris_kernels.h"

namespace te {}
using namespace te;

static __device__ const ShapeTable d_shape_table = make_shape_table();

template <int P, int MODE, bool TINT>
__global__ __launch_bounds__(256) void k_game(KArgs a) {
    // Shape table in LDS, one private 128-byte copy per wave: no workgroup barrier, so a wave starts computing as soon as
    // the state words it needs first have arrived instead of waiting for all loads of all four waves.  The table load is
    // issued before the state loads (loads return in order), and ds_write -> ds_read order within a wave is by lgkmcnt.
    __shared__ __attribute__((aligned(16))) uint32_t s_shapes_all[4][SHAPE_WORDS];
    uint32_t* s_shapes = s_shapes_all[threadIdx.x >> 6];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = lane_active(a, i);
    LaneCounters cnt = {0, 0, 0, 0};                   // (per-lane sums feed the CPU test harness only)
    TE_STAMP(0); TE_STAMP_RT(1);
    const uint32_t shape_word = d_shape_table.s[threadIdx.x & 63];
    Game<P> g;
    if (active) game_load<P, MODE, TINT>(a, i, g);
    TE_STAMP(2);
#if defined(TE_PHASE_TRACE) && TE_PHASE_TRACE == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // variant: time until ALL state words have arrived
    TE_STAMP(3);
#endif
    s_shapes[threadIdx.x & 63] = shape_word;
    __builtin_amdgcn_wave_barrier();
    if (active) game_run<P, MODE, TINT>(a, i, s_shapes, g, cnt);
    TE_STAMP(9);
#if defined(TE_PHASE_TRACE) && TE_PHASE_TRACE == 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // variant: time until all stores are acknowledged
    TE_STAMP(10);
#endif
    TE_STAMP(14); TE_STAMP_RT(15);
}

// launches k_game<P, mode, tint> for P = 3, 4 (tetris_hip_multi.hip)
__attribute__((visibility("hidden"))) int tetris_launch_game_multi(int n_players, int tint, int mode, dim3 grid, dim3 block, hipStream_t stream, const te::KArgs& a);
