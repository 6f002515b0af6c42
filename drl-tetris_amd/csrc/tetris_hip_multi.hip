// libtetris_hip.so, second translation unit: the step kernel for three and four players per game (tetris_game_kernel.h).
// Nothing else lives here; the two files are compiled in parallel (__graft_entry__.build_hip).
#include "tetris_game_kernel.h"

template <int P, bool TINT>
static int launch_mode(int mode, dim3 grid, dim3 block, hipStream_t st, const KArgs& a) {
    switch (mode) {
#define TE_CASE(M) case M: hipLaunchKernelGGL((k_game<P, M, TINT>), grid, block, 0, st, a); return 0
        TE_CASE(M_INIT); TE_CASE(M_RESET); TE_CASE(M_MAKE); TE_CASE(M_FINISH); TE_CASE(M_STEP_KEYS); TE_CASE(M_STEP_RT);
        TE_CASE(M_ROLLOUT); TE_CASE(M_STEP_RT_AUTO); TE_CASE(M_RESET_SCHED);
#undef TE_CASE
        default: return -1;
    }
}

__attribute__((visibility("hidden"))) int tetris_launch_game_multi(int n_players, int tint, int mode, dim3 grid, dim3 block, hipStream_t st, const KArgs& a) {
    if (n_players == 3) return tint ? launch_mode<3, true>(mode, grid, block, st, a) : launch_mode<3, false>(mode, grid, block, st, a);
    if (n_players == 4) return tint ? launch_mode<4, true>(mode, grid, block, st, a) : launch_mode<4, false>(mode, grid, block, st, a);
    return -1;
}
