// libtetris_hip.so — kernels and C ABI of the MI355X batched Tetris environment (gfx950 only).
//
// One lane = one game (all its players), state SoA in HBM (tetris_layout.h), step logic in
// tetris_engine.h.  Launch geometry: 256-thread workgroups (one wave per SIMD of a CU); 64k games
// = 256 workgroups = one per CU, so the kernel is latency/issue bound, not occupancy bound.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <new>
#include <chrono>
#include <string>
#include <vector>

#include "tetris_kernels.h"

using namespace te;

// ============================================================================ device side

#include "tetris_game_kernel.h"

// Chained launches of the built-in rollout (one env-step per launch).  A game's step E depends on nothing but the same game's
// step E - 1, yet launches on one stream are separated by a full barrier: launch E waits for the SLOWEST wave of launch E - 1
// plus the kernel boundary (~1.4 us of a ~5.7 us launch period, profiles/).  Here consecutive launches rotate over CHAIN_STREAMS (3)
// streams, so launch E is dispatched while E - 1 and E - 2 still run, and the dependence is enforced per WAVE (one workgroup = one
// wave = 64 games): wave w of launch E polls an epoch word until wave w of launch E - 1 has published E - 1.  Every state
// load / store is agent-scope (`sc1`: written through, read past the per-XCD L2), the storing wave drains its stores
// (`s_waitcnt vmcnt(0)`) before it publishes — the measured-valid hand-off of MI355X_MICROARCH.md.  Spins are bounded: a wave
// that gives up marks its epoch word (CHAIN_ABANDONED), raises F_CHAIN and leaves its games untouched; so does the same wave of
// every later launch, and the host finishes those games un-chained once the streams have drained (chain_recover).
#ifndef TE_CHAIN_LANES
#define TE_CHAIN_LANES 64          // games per wave of k_chain (experiment knob: 32 / 16 = emptier waves, more of them per SIMD)
#endif
constexpr int CHAIN_LANES = TE_CHAIN_LANES;
#ifndef TE_CHAIN_STREAMS
#define TE_CHAIN_STREAMS 3         // streams the chained launches rotate over = launches in flight.  3 x 1024 waves of 64k single-player
                                   // boards fit in the 15 x 256 wave slots chain_fits counts; GPU-paced 4.02 us per launch against 4.14 with 2
                                   // (with the epoch words still packed 32 to a line it had been 4.9-5.06 against 4.71-4.80)
#endif
constexpr int CHAIN_STREAMS = TE_CHAIN_STREAMS;
// XCD-AFFINE form (k_chain_affine; direct dispatch only, tetris_aql.h).  The write-through hand-off above crosses the fabric twice per
// step because block b of consecutive launches never meets its own XCD again (a queue deals its blocks round-robin over the eight XCDs
// from a start of its own: profiles/r03/handoff_experiments.txt).  Here the GAMES follow the XCD instead: the workgroup that finds
// itself on XCD x takes the game block (b & ~7) | x, so a block is stepped on the same XCD in every launch and its state and epoch
// word can stay in that XCD's L2 — plain stores, `sc1` loads (past the CU's L1, served by the L2).  One L2 is coherent for all CUs
// of its XCD; nothing else is relied on.  That the eight workgroups of a group of eight really sit on eight different XCDs is
// CHECKED, by every workgroup for itself: the queue's start XCD is measured when the queue is made (a.xcd_base), and a
// workgroup that is not on XCD (xcd_base + b) mod 8 touches nothing, raises F_PLACE and leaves — its games then look abandoned to the
// next launch, whose waves give up, and the host finishes the call un-chained (chain_recover) and switches the affine form off.
// What makes this FASTER only with direct dispatch: the packets between a queue's first and last carry no cache maintenance
// (a kernel boundary's L2 write-back + invalidate, three times per 12 us, cost more than the fabric: +0.1 us per launch through
// streams); the queue's last packet releases, so memory is current when the call returns.
template <int P, bool AFFINE>
__device__ __forceinline__ void chain_body(const KArgs& a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_shapes[SHAPE_WORDS];
    const int lane = threadIdx.x;
    if (a.steps < 0) { chain_census(a, lane == 0); return; }
    constexpr int CMEM = AFFINE ? MEM_AFFINE : MEM_AGENT;
    int wave = blockIdx.x;
    if (AFFINE) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (blockIdx.x == 0 && lane == 0) ((volatile uint32_t*)a.status)[F_XCC0 + (a.xcd_slot & 3u)] = 0x100u | xcc;      // where this queue starts dealing: the host's expectation for the next call
        if (xcc != ((a.xcd_base + blockIdx.x) & 7u)) {          // not where the host expects this block to run: hands off
            if (lane == 0) ((volatile uint32_t*)a.status)[F_PLACE] = 1u;
            return;
        }
        wave = (int)((blockIdx.x & ~7u) | xcc);
        if (wave * CHAIN_LANES >= a.n) return;                  // (padding of the last group of eight: no games, no epoch word)
    }
    const int i = wave * CHAIN_LANES + lane;
    const bool active = lane < CHAIN_LANES && i < a.n;
    LaneCounters cnt = {0, 0, 0, 0};
    const uint32_t shape_word = d_shape_table.s[lane];
    Game<P> g;
    // the policy draw of this step depends on kernel arguments only: its 40 dependent multiplies run while the wave waits.
    // (Issuing the first poll of the epoch word BEFORE the draw — most waves find their predecessor done at that poll — measured
    // +0.06 us per launch, GPU-paced: profiles/r02/chain_ab_gpu_paced.txt.)
    TE_STAMP_CHAIN(a.epoch, 0); TE_STAMP_PLACE(a.epoch);
    if (active) policy_draw(a, (uint32_t)i, a.first_step, g.draw0, g.draw1);
    const uint32_t d0 = g.draw0, d1 = g.draw1;
    TE_STAMP_CHAIN(a.epoch, 1);
    if (!chain_wait(a, (uint32_t)wave, lane == 0)) return;      // gave up: the games stay as launch E - 1 (or an earlier one) left them
    TE_STAMP_CHAIN(a.epoch, 2);
    if (active) { load_game<P>(geo_of(a), (size_t)i, g, false, P > 1, true, CMEM, CHAIN_LANES == 64); g.draw0 = d0; g.draw1 = d1; }
#if defined(TE_PHASE_TRACE)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // diagnostic: when have ALL state words arrived
    TE_STAMP_CHAIN(a.epoch, 3);
#endif
    s_shapes[lane] = shape_word;
    __builtin_amdgcn_wave_barrier();
    if (active) game_run<P, M_ROLLOUT, false, CMEM>(a, i, s_shapes, g, cnt);
    TE_STAMP_CHAIN(a.epoch, 4);
#if !defined(TE_EXPERIMENT_NO_ACK)      // (timing experiment only, results INVALID: what would a hand-off that does not wait for the store acknowledgements gain?)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every store (and counter atomic) of this wave has been acknowledged
#endif
    TE_STAMP_CHAIN(a.epoch, 5);
    if (lane == 0) {
        if (AFFINE) *(volatile uint32_t*)(a.chain + (size_t)wave * CHAIN_STRIDE) = a.epoch;       // stays in this XCD's L2, where the next launch's wave polls it
        else st_agent(a.chain + (size_t)wave * CHAIN_STRIDE, a.epoch);
    }
    TE_STAMP_CHAIN(a.epoch, 6);
}
template <int P>
__global__ __launch_bounds__(64) void k_chain(KArgs a) { chain_body<P, false>(a); }
template <int P>
__global__ __launch_bounds__(64) void k_chain_affine(KArgs a) { chain_body<P, true>(a); }
template __global__ void k_chain_affine<1>(KArgs);
// the XCD every workgroup of a launch lands on (calibration of the affine form: aql::make_queues)
extern "C" __global__ void tetris_k_xcc_probe(uint32_t* out) {
    if (threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = xcc & 0xFu;
    }
}

// Measurement / test aid: keeps a stream busy until the HOST sets `*go` (a word in pinned host memory; NULL: no such word) or
// `ticks` of the 100 MHz real-time clock have passed, whichever comes first.  TETRIS_PREQUEUE=1 parks the chain streams behind it
// while the host enqueues, so that every launch of the call is queued before the first one starts — the GPU-paced launch period
// without any host pacing; tetris_debug_stall uses it to hold a stream or the device's wave slots (tests of the give-up path).
__global__ void k_blocker(const uint32_t* go, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (go && __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// Measurement aid (tetris_debug_clock_khz): shader clock of the moment, from the ratio of the shader cycle counter to the constant
// 100 MHz real-time counter over ~`ticks` of the latter, by one wave
__global__ void k_clock_probe(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(4); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

// Sums the per-game cumulative rollout counters (G_STEPS, G_EPISODE, G_LINES, G_SENT): run once before and once
// after a rollout, outside its timed region, instead of any cross-lane reduction inside the step
// kernel (4096 same-address atomics per launch cost ~28 us; a shuffle + LDS + read-modify-write tail
// still ~1.2 us of a 8 us launch).
__global__ __launch_bounds__(256) void k_totals(Geo geo, unsigned long long* out /*[4]*/) {
    unsigned long long v[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (int)geo.n_games; i += gridDim.x * blockDim.x) {
        unsigned long long t[4];
        totals_of_game(geo, i, t);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    for (int k = 0; k < 4; k++) {
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
        if ((threadIdx.x & 63) == 0 && v[k]) atomicAdd(&out[k], v[k]);
    }
}

// ---- step + observation in one launch (tetris_step_rt_observe_dev) ------------------------------------------------------------
// One agent-loop iteration in ONE launch (worker.py:91-118: perform_action, then get_state + unpack for the next decision): the
// (r, t) step and, straight from the registers the step leaves behind, the packed observation k_observe_packed would rebuild
// from the stored state — no second launch, no second read of the state.  A wave's byte planes go through an LDS tile and leave
// as contiguous runs of 16-byte stores.  Even heights only (H * 10 cells = whole words per board).
// one board -> its row of the tile, its 12 vector bytes and its piece byte (slot sl, game i): as k_observe_packed
// (a macro, not a function: as an inlined function with the board passed by reference, by value or as seventeen scalars, the
// compiler left the step's column array in scratch memory — 128 B per lane and +1.5 us per launch)
#define TE_OBS_BUILD(a, row, sl, i, COL, qx, qy, qnext, qkind, qinc, qcombo, qrem)                                              \
    do {                                                                                                                        \
        uint32_t col_[NCOL];                                                                                                    \
        TE_UNROLL                                                                                                               \
        for (int c = 0; c < NCOL; c++) col_[c] = COL(c);                                                                        \
        for (int yp = 0; yp < (a).H / 2; yp++) {                                                                                \
            uint32_t lo[NCOL], hi[NCOL]; /* cells of rows 2 yp and 2 yp + 1 */                                                  \
            TE_UNROLL                                                                                                           \
            for (int c = 0; c < NCOL; c++) { lo[c] = col_[c] & 1u; hi[c] = (col_[c] >> 1) & 1u; col_[c] >>= 2; }                \
            (row)[5 * yp + 0] = lo[0] | (lo[1] << 8) | (lo[2] << 16) | (lo[3] << 24);                                           \
            (row)[5 * yp + 1] = lo[4] | (lo[5] << 8) | (lo[6] << 16) | (lo[7] << 24);                                           \
            (row)[5 * yp + 2] = lo[8] | (lo[9] << 8) | (hi[0] << 16) | (hi[1] << 24);                                           \
            (row)[5 * yp + 3] = hi[2] | (hi[3] << 8) | (hi[4] << 16) | (hi[5] << 24);                                           \
            (row)[5 * yp + 4] = hi[6] | (hi[7] << 8) | (hi[8] << 16) | (hi[9] << 24);                                           \
        }                                                                                                                       \
        /* state_processors.py:23-54 vector: x, y, incoming, combo time, combo count, one-hot next piece */                     \
        const uint32_t x_ = (uint32_t)(qx) & 0xFFu, y_ = (uint32_t)(qy) & 31u, next_ = (uint32_t)(qnext) & 7u;                  \
        uint32_t t_ = (((uint32_t)(qrem) & 0xFFFFu) + 50u) & 0xFFFFu; /* uint16 + 50 wraps like numpy (state_processors.py:38) */ \
        if (t_ > 25000u) t_ = 25000u;                                                                                           \
        uint32_t* v_ = (uint32_t*)((a).obs_vector + ((size_t)(sl) * (a).n + (i)) * 12);                                         \
        const uint64_t hot_ = next_ < 7u ? (1ull << (8 * next_)) : 0ull; /* bytes 5..11 */                                      \
        v_[0] = x_ | (y_ << 8) | (((uint32_t)(qinc) & 255u) << 16) | ((t_ / 100u) << 24);                                       \
        v_[1] = ((uint32_t)(qcombo) & 255u) | ((uint32_t)(hot_ & 0xFFFFFFu) << 8);                                              \
        v_[2] = (uint32_t)(hot_ >> 24);                                                                                         \
        (a).obs_piece[(size_t)(sl) * (a).n + (i)] = (uint8_t)((uint32_t)(qkind) & 7u);                                          \
    } while (0)
// the same as a function of scalars — the form that stays in registers inside k_duo (there the macro cost 784 B of scratch per lane)
__device__ __forceinline__ void obs_build(const KArgs& a, uint32_t* row, int sl, int i, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4,
                                          uint32_t c5, uint32_t c6, uint32_t c7, uint32_t c8, uint32_t c9, int qx, int qy, int qnext, int qkind,
                                          int inc_count, int combo_count, uint32_t combo_remaining) {
    const uint32_t cols[NCOL] = {c0, c1, c2, c3, c4, c5, c6, c7, c8, c9};
#define TE_OBS_COL(c) cols[c]
    TE_OBS_BUILD(a, row, sl, i, TE_OBS_COL, qx, qy, qnext, qkind, inc_count, combo_count, combo_remaining);
#undef TE_OBS_COL
}
// `nb` consecutive tile rows (boards first .. first + nb of slot sl) -> visual, by the 64 lanes of one wave.  The tile is NOT padded
// (row pitch = nw words), so it is the output image itself: one 16-byte LDS read per 16-byte store, no index arithmetic.  (The rows
// of a wave's lanes then start 50 / 55 / ... words apart: two lanes per LDS bank at most while the rows are written.)
__device__ __forceinline__ void obs_stream(const KArgs& a, const uint32_t* rows, int sl, int first, int nb, int lane) {
    const int nw = a.H * NCOL / 4;
    uint32_t* dst = (uint32_t*)(a.obs_visual + ((size_t)sl * a.n + first) * (size_t)(a.H * NCOL));
    const uint32_t words = (uint32_t)nb * (uint32_t)nw;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (((((uintptr_t)dst) | ((uintptr_t)rows)) & 15u) == 0) {
        const uint32_t whole = words & ~3u;
        for (uint32_t k = 4u * (uint32_t)lane; k < whole; k += 4u * 64u)
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(rows + k), reinterpret_cast<u32x4*>(dst + k));
        if ((uint32_t)lane < words - whole) dst[whole + lane] = rows[whole + lane];
    } else
        for (uint32_t k = (uint32_t)lane; k < words; k += 64u) dst[k] = rows[k];
}

// lane = game (k_step_observe): slot 0 = the deciding player's board, slot 1 = the opponent's, one slot after the other
template <int P>
__device__ __forceinline__ void observe_emit(const KArgs& a, const Game<P>& g, int first, int lane, bool active, uint32_t* tile) {
    const int pitch = a.H * NCOL / 4, i = first + lane;
    const int nb = (a.n - first < 64) ? a.n - first : 64;
    const int me = active ? safe_player(a.next_player, i, P) : 0;
    TE_UNROLL
    for (int sl = 0; sl < P; sl++) {
        if (active) {
            const bool sel = P > 1 && (sl == 0 ? me == 1 : me == 0);      // field-wise select of the second player's board
            const Player& q0 = g.pl[0];
            const Player& q1 = g.pl[P - 1];
            uint32_t* row = tile + (size_t)lane * pitch;
#define TE_OBS_COL(c) (sel ? q1.col[c] : q0.col[c])
            TE_OBS_BUILD(a, row, sl, i, TE_OBS_COL, sel ? q1.x : q0.x, sel ? q1.y : q0.y, sel ? q1.next : q0.next, sel ? q1.kind : q0.kind,
                         sel ? q1.inc_count : q0.inc_count, sel ? q1.combo_count : q0.combo_count, sel ? q1.combo_remaining : q0.combo_remaining);
#undef TE_OBS_COL
        }
        __syncthreads();
        obs_stream(a, tile, sl, first, nb, lane);
        __syncthreads();
    }
}

// "Duo" mapping for two-player games (BASELINE config 3): the two players of a game sit in lanes l and l ^ 32 of ONE wave,
// so 64k games are 2 waves per SIMD instead of 1 and the second player's clear/spawn, timers and state traffic overlap with
// the first's.  The in-step order dependence between the players is the split-mode stage protocol (tetris_engine.h) with
// the exchange words moved by __shfl_xor(.., 32): A = key interpreter + loop 1 (player 1 speculatively, with a register
// backup for the rollback when player 0 died), B0 / B1 = delayCheck of player 0, then player 1, C = winner logic.
// CHAIN (rollout only, 64-thread workgroups): chained launches as in k_chain — a wave's 32 games wait for the epoch word the same
// wave of the previous launch published, and all state traffic is agent-scope.
// OBS (tetris_step_rt_observe_dev): after the step every lane turns its own board into the packed observation — slot 0 if its
// player is the one the game's next decision is for, slot 1 otherwise; rows 0..31 / 32..63 of the wave's LDS tile.
// AFFINE (with CHAIN; k_duo_affine, direct dispatch only): the XCD-affine hand-over of k_chain_affine — a wave's 32 games follow the XCD.
template <int MODE, bool CHAIN, bool OBS, bool AFFINE>
__device__ __forceinline__ void duo_body(const KArgs& a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_shapes_all[4][SHAPE_WORDS];      // per-wave copy, no block barrier (see k_game)
    extern __shared__ __attribute__((aligned(16))) uint32_t s_duo_tile[];               // OBS: 4 waves x 64 rows x nw words
    uint32_t* s_shapes = s_shapes_all[threadIdx.x >> 6];
    const uint32_t shape_word = d_shape_table.s[threadIdx.x & 63];
    const int lane = threadIdx.x & 63, side = lane >> 5;
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    constexpr bool ROLL = MODE == M_ROLLOUT, AUTO = MODE == M_STEP_RT_AUTO;
    constexpr int MEM = CHAIN ? (AFFINE ? MEM_AFFINE : MEM_AGENT) : MEM_STREAM;
    if (CHAIN && a.steps < 0) { chain_census(a, lane == 0); return; }
    if (AFFINE) {                        // (64-thread workgroups: one wave each; see chain_body)
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (blockIdx.x == 0 && lane == 0) ((volatile uint32_t*)a.status)[F_XCC0 + (a.xcd_slot & 3u)] = 0x100u | xcc;
        if (xcc != ((a.xcd_base + blockIdx.x) & 7u)) {
            if (lane == 0) ((volatile uint32_t*)a.status)[F_PLACE] = 1u;
            return;
        }
        wave = (int)((blockIdx.x & ~7u) | xcc);
        if (wave * 32 >= a.n) return;
    }
    const int gi = wave * 32 + (lane & 31);
    const bool active = gi < a.n;
    Geo geo = geo_of(a);
    geo.P = 2;                           // compile-time stride factor for the hot loads
    const Ref gr = game_ref(geo, (size_t)gi, true);
#if TE_TILED
    const Ref br = board_ref(geo, side, (size_t)gi);
#else
    // one UNIFORM row base for both half-waves (it becomes a buffer resource): the player's offset goes into the lane offset
    const Ref br = {geo.state, (uint32_t)((size_t)side * geo.stride + (size_t)gi) * 4u, 2 * geo.stride};
#endif
    Game<1> g;
    Player& q = g.pl[0];
    uint32_t pd0 = 0, pd1 = 0;
    if (CHAIN) {
        if (active) policy_draw(a, (uint32_t)gi, a.first_step, pd0, pd1);        // while the wave waits for its predecessor
        if (!chain_wait(a, (uint32_t)wave, lane == 0)) return;
    }
    if (active) {
        load_game_words<1>(gr, g, ROLL, MEM);
        load_player(br.s, br.o, br.ws, q, false, true, MEM);
        if (CHAIN) { g.draw0 = pd0; g.draw1 = pd1; }
        else if (ROLL) policy_draw(a, (uint32_t)gi, a.first_step, g.draw0, g.draw1);   // under the loads (see game_load)
        else { g.draw0 = a.rot[gi]; g.draw1 = (uint32_t)a.trans[gi] | ((a.player ? (uint32_t)a.player[gi] : 0u) << 8); }
    }
    s_shapes[lane] = shape_word;
    __builtin_amdgcn_wave_barrier();
    Ctx cx = make_ctx(a, s_shapes, false);
    uint32_t my_lines = 0, my_sent = 0;
    int done = 0, out_reward = 0, out_dead = 0;
    // ONE env-step per launch (launch_game sends fused rollouts to k_game<2>): as a loop over a.steps this kernel needed 199
    // registers per lane instead of ~130 — everything stayed live around the loop's back edge.
    {
        int r = 0, t = 0, acting = 0;
        ResetPrefetch rpf;
        rpf.ok = 0; rpf.seed16 = 0; rpf.word = 0;
        uint32_t sent_start = 0;
        uint32_t wa = 0;
        uint32_t undo_pose = 0, undo_group = 0, undo_draws = 0, undo_cleared = 0;
        bool undo_ran = false;
        Player pre;                                                    // un-chained kernels only: player 1's board before its speculative pass
        if (active) {
            if (ROLL) {
                r = (int)(g.draw0 & 3u); t = (int)(g.draw1 % 10u); acting = (int)(a.first_step % 2ull);
            } else {
                r = (int)(g.draw0 & 3u); t = (int)(g.draw1 & 0xFFu); acting = (int)(g.draw1 >> 8);
            }
            if (ROLL || AUTO) prefetch_reset(cx, episode_seed(a.game_offset + (uint32_t)gi, g.episode + 1), rpf);
            prefetch_next(cx, q, g.seed16, g.status);
            sent_start = q.lines_sent;
            // stage A
            if (!g.round_over && !q.dead && acting == side) play_rt(cx, q, r, t);
            // Player 1's loop-1 pass is speculative (the reference skips it when player 0 died in loop 1, PythonHandle.cpp:153-156,
            // which this lane learns from the shuffle below).  What the pass changes when it clears no row and the new piece
            // fits — nearly always — is the piece (kind, rotation, position, next), the draw counter, now and then the piece
            // group, and 200 ms of combo time: an undo record of three registers.  (A full copy of the board in registers cost
            // ~80 of them in every lane of every step; loading the state again and replaying the keys on EVERY rollback made the
            // slowest wave of most launches ~2 us longer.)  The chained kernel — which has to stay within 128 registers for two
            // launches of 64k games to fit on the device — falls back to that reload where the record does not do; a wave that
            // takes it (1-2 % of them per launch) delays only its own chain.  The un-chained kernels keep the copy in registers
            // instead (they have room): there a launch ends with its SLOWEST wave, and one reload per launch cost every launch
            // 1.4 us (10.0 against 8.6 us, same box: profiles/r03/ab_r02.txt).
            if (!CHAIN && side == 1) pre = q;
            undo_pose = pose_pack(q);
            undo_group = q.pgroup; undo_draws = q.piece_draws;
            undo_cleared = q.lines_cleared; undo_ran = !g.round_over && !q.dead;
            wa = split_settle(cx, g);
        }
        const uint32_t opp_a = __shfl_xor(wa, 32);
        uint32_t wb0 = 0, wb1 = 0;
        if (active) {
            if (side == 1 && (opp_a & XW_DIED)) {
                if (!CHAIN) {
                    q = pre;
                } else if (undo_ran && q.lines_cleared == undo_cleared && !q.dead) {
                    undo_simple_settle(q, undo_pose, undo_group, undo_draws);
                } else if (undo_ran) {
                    // rows were cleared or the new piece did not fit (rare together with a rollback): nothing has been stored by
                    // this launch yet, so the state before this step is still in memory — load it again and replay the keys
                    load_player(br.s, br.o, br.ws, q, false, true, MEM);
                    prefetch_next(cx, q, g.seed16, g.status);
                    if (!g.round_over && !q.dead && acting == side) play_rt(cx, q, r, t);
                    sent_start = q.lines_sent;
                }
                wa = 0;
            }
            if (side == 0) {                                           // stage B0
                const int in = (!(wa & XW_DIED) && (opp_a & XW_RAN) && !(opp_a & XW_DIED)) ? xw_sent(opp_a) : 0;
                wb0 = split_tick(cx, g, a.ms, in);
            }
        }
        const uint32_t x0 = __shfl_xor(wb0, 32);
        if (active && side == 1) {                                     // stage B1
            const int in1 = ((opp_a & XW_RAN) && !(opp_a & XW_DIED)) ? xw_sent(opp_a) : 0;
            if (!g.round_over && in1 > 0) q.incoming = q.incoming + (float)in1 / 1.0f;
            wb1 = split_tick(cx, g, a.ms, (x0 & XW_DIED) ? 0 : xw_sent(x0));
        }
        const uint32_t x1 = __shfl_xor(wb1, 32);
        if (active) {                                                  // stage C
            const uint32_t opp_b = side == 0 ? x1 : x0;
            const int in = (side == 0 && !(opp_b & XW_DIED)) ? xw_sent(opp_b) : 0;
            g.flags = (uint32_t)side;
            done = split_finish(g, in, (opp_b & XW_DEAD_NOW) != 0, (opp_b & XW_ERR) != 0);
            out_reward = q.reward; out_dead = q.dead;                  // what the step reports: the state BEFORE an auto-reset
            if (ROLL) {
                g.steps++;
                if (!q.dead) my_lines += (unsigned)q.reward;
                my_sent += (q.lines_sent - sent_start) & 0xFFFFu;
            }
            if ((ROLL || AUTO) && done) {
                g.episode++;
                reset_split(cx, g, episode_seed(a.game_offset + (uint32_t)gi, g.episode));
            }
        }
    }
    const uint32_t opp_lines = __shfl_xor(my_lines, 32), opp_sent = __shfl_xor(my_sent, 32);
    if (active) {
        store_player(br.s, br.o, br.ws, q, false, true, MEM);
        if (!ROLL) {
            if (a.lines) a.lines[(size_t)side * a.n + gi] = (uint8_t)out_reward;
            if (a.dead) a.dead[(size_t)side * a.n + gi] = (uint8_t)out_dead;
        }
        const uint32_t st = g.status | __shfl_xor(g.status, 32);
        if (side == 0) {
            if (!ROLL && a.done) a.done[gi] = (uint8_t)done;
            g.flags = 0;
            g.add_lines = my_lines + opp_lines;
            g.add_sent = my_sent + opp_sent;
            store_game_words<1>(gr, g, ROLL, MEM);
            report_status(a, st);
        }
    }
    if (CHAIN) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store (and counter atomic) of this wave has been acknowledged
        if (lane == 0) {
            if (AFFINE) *(volatile uint32_t*)(a.chain + (size_t)wave * CHAIN_STRIDE) = a.epoch;
            else st_agent(a.chain + (size_t)wave * CHAIN_STRIDE, a.epoch);
        }
    }
    if (OBS) {
        const int pitch = a.H * NCOL / 4, first = wave * 32;
        uint32_t* tile = s_duo_tile + ((size_t)(threadIdx.x >> 6) * 64 * pitch + 3) / 4 * 4;       // (16-byte aligned per wave)
        if (active) {
            const int sl = side == safe_player(a.next_player, gi, 2) ? 0 : 1;
            obs_build(a, tile + (size_t)(sl * 32 + (lane & 31)) * pitch, sl, gi, q.col[0], q.col[1], q.col[2], q.col[3], q.col[4], q.col[5], q.col[6],
                      q.col[7], q.col[8], q.col[9], q.x, q.y, q.next, q.kind, q.inc_count, q.combo_count, q.combo_remaining);
        }
        __syncthreads();
        const int nb = (a.n - first < 32) ? a.n - first : 32;
        if (nb > 0) {
            obs_stream(a, tile, 0, first, nb, lane);
            obs_stream(a, tile + (size_t)32 * pitch, 1, first, nb, lane);
        }
    }
}
template <int MODE, bool CHAIN = false, bool OBS = false>
__global__ __launch_bounds__(CHAIN ? 64 : 256) void k_duo(KArgs a) { duo_body<MODE, CHAIN, OBS, false>(a); }
__global__ __launch_bounds__(64) void k_duo_affine(KArgs a) { duo_body<M_ROLLOUT, true, false, true>(a); }

template <int STAGE, bool TINT>
__global__ __launch_bounds__(256) void k_split(KArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_shapes_all[4][SHAPE_WORDS];      // per-wave copy, no block barrier (see k_game)
    uint32_t* s_shapes = s_shapes_all[threadIdx.x >> 6];
    s_shapes[threadIdx.x & 63] = d_shape_table.s[threadIdx.x & 63];
    __builtin_amdgcn_wave_barrier();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) split_body<STAGE, TINT>(a, i, s_shapes);
}

// ---- RNG tables: one lane per 16-bit seed, MT state strided [word][seed] (coalesced)
__global__ __launch_bounds__(256) void k_gen_seed(uint32_t* mt) {
    uint32_t seed = blockIdx.x * blockDim.x + threadIdx.x;
    if (seed >= 65536u) return;
    // randomizer.cpp:34-36,47-49: `short` seed converted to the engine's result type
    mt_seed(mt + seed, 65536, (uint32_t)(int32_t)(int16_t)(uint16_t)seed);
}

struct MapArg { uint8_t m[8]; };

__global__ __launch_bounds__(256) void k_gen_chunk(uint32_t* mt, float* w, uint8_t* out, uint64_t* start, int chunk,
                                                   MapArg map, int only_sz) {
    uint32_t seed = blockIdx.x * blockDim.x + threadIdx.x;
    if (seed >= 65536u) return;
    uint64_t word = 0;
    gen_chunk_for_seed(mt + seed, 65536, w + seed, 65536, out + (size_t)seed * CHUNK, &word, chunk, map.m, only_sz != 0);
    if (chunk == 0) start[seed] = word;
}

template <int P, bool TINT>
__global__ __launch_bounds__(256) void k_observe(Geo geo, int n, const int32_t* idx, int H, tetris_record* rec, uint8_t* round_over,
                                                 int8_t* last_winner) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) observe_body<P, TINT>(geo, i, idx, H, d_shape_table.s, rec, round_over, last_winner);
}

// Observation kernel: one lane reads its board's ten column words (coalesced SoA), expands them to H*10 bytes
// in LDS; after a barrier the workgroup streams its 256 boards' planes out as ONE contiguous run of dwords, so the
// uint8 rows leave the chip fully coalesced (a lane writing its own 200 bytes would not be).
template <int P>
__global__ __launch_bounds__(256) void k_observe_packed_bytes(Geo geo, int n, const int32_t* idx,
                                                        const uint8_t* player, int H, uint8_t* visual, uint8_t* vector,
                                                        uint8_t* piece) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_cells[];      // 256 * H * 10 bytes
    const int cells = H * NCOL;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int first = blockIdx.x * 256;
    const int nb = (n - first < 256) ? n - first : 256;
    for (int sl = 0; sl < P; sl++) {
        if (i < n) {
            const size_t slot = safe_slot(idx, i, (int)geo.n_games);
            const int me = safe_player(player, i, P);
            const int p = (sl == 0) ? me : (P - 1 - me);
            uint8_t vec[12];
            const int kind = observe_board(geo, slot, p, H, s_cells + (size_t)threadIdx.x * cells, vec);
            uint8_t* v = vector + ((size_t)sl * n + i) * 12;
            for (int k = 0; k < 12; k++) v[k] = vec[k];
            piece[(size_t)sl * n + i] = (uint8_t)kind;
        }
        __syncthreads();
        uint8_t* out = visual + ((size_t)sl * n + first) * cells;
        const size_t bytes = (size_t)nb * cells;
        if ((((uintptr_t)out | bytes) & 3u) == 0) {
            const uint32_t* src = (const uint32_t*)s_cells;
            uint32_t* dst = (uint32_t*)out;
            for (size_t k = threadIdx.x; k < bytes / 4; k += 256) dst[k] = src[k];
        } else {
            for (size_t k = threadIdx.x; k < bytes; k += 256) out[k] = s_cells[k];
        }
        __syncthreads();
    }
}

// Even heights (H * 10 cells = a whole number of words per board, the usual 20 / 22 rows): the byte planes are built as
// words in registers — two rows = 20 cells = 5 words per loop trip, columns indexed statically — and go to LDS as dwords
// instead of 200 byte writes per lane; the workgroup's tile then leaves as coalesced 16-byte stores.  The 12 vector bytes of a
// board leave as 3 dwords.
template <int P, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_observe_packed(Geo geo, int n, const int32_t* idx,
                                                          const uint8_t* player, int H, uint8_t* visual, uint8_t* vector,
                                                          uint8_t* piece) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_words[];      // BLOCK * nw words: the workgroup's slice of `visual` itself
    const int nw = H * NCOL / 4;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int first = blockIdx.x * BLOCK;
    const int nb = (n - first < BLOCK) ? n - first : BLOCK;
    {
        const int sl = blockIdx.y;                       // one workgroup per (64 boards, slot): twice the waves in flight for P = 2
        if (i < n) {
            const size_t slot = safe_slot(idx, i, (int)geo.n_games);
            const int me = safe_player(player, i, P);
            const int p = (sl == 0) ? me : (P - 1 - me);
            const Ref br = board_ref(geo, p, slot);
            uint32_t col[NCOL];
            for (int c = 0; c < NCOL; c++) col[c] = word_at(br, W_COL0 + c);
            const uint32_t w = word_at(br, W_PIECE), m = word_at(br, W_MISC), dc = word_at(br, W_DROPCOMBO);
            uint32_t* row = s_words + (size_t)threadIdx.x * nw;
            for (int yp = 0; yp < H / 2; yp++) {
                uint32_t lo[NCOL], hi[NCOL];                 // cells of rows 2 yp and 2 yp + 1
                for (int c = 0; c < NCOL; c++) { lo[c] = col[c] & 1u; hi[c] = (col[c] >> 1) & 1u; col[c] >>= 2; }
                row[5 * yp + 0] = lo[0] | (lo[1] << 8) | (lo[2] << 16) | (lo[3] << 24);
                row[5 * yp + 1] = lo[4] | (lo[5] << 8) | (lo[6] << 16) | (lo[7] << 24);
                row[5 * yp + 2] = lo[8] | (lo[9] << 8) | (hi[0] << 16) | (hi[1] << 24);
                row[5 * yp + 3] = hi[2] | (hi[3] << 8) | (hi[4] << 16) | (hi[5] << 24);
                row[5 * yp + 4] = hi[6] | (hi[7] << 8) | (hi[8] << 16) | (hi[9] << 24);
            }
            // state_processors.py:23-54 vector: x, y, incoming, combo time, combo count, one-hot next piece
            const uint32_t x = (uint32_t)((int)((w >> 5) & 15) - 4) & 0xFFu, y = (w >> 9) & 31u, next = (w >> 14) & 7u;
            uint32_t t = ((dc >> 16) + 50u) & 0xFFFFu;     // uint16 + 50 wraps like numpy (state_processors.py:38)
            if (t > 25000u) t = 25000u;
            uint32_t* v = (uint32_t*)(vector + ((size_t)sl * n + i) * 12);
            const uint64_t hot = next < 7u ? (1ull << (8 * next)) : 0ull;          // bytes 5..11
            v[0] = x | (y << 8) | ((m & 255u) << 16) | ((t / 100u) << 24);
            v[1] = ((m >> 8) & 255u) | ((uint32_t)(hot & 0xFFFFFFu) << 8);
            v[2] = (uint32_t)(hot >> 24);
            piece[(size_t)sl * n + i] = (uint8_t)(w & 7u);
        }
        __syncthreads();
        // the tile is not padded (rows nw words apart: at most two lanes per LDS bank while it is written), so it IS the output
        // image: one 16-byte LDS read per 16-byte streaming store, no index arithmetic
        uint32_t* dst = (uint32_t*)(visual + ((size_t)sl * n + first) * (size_t)(H * NCOL));
        const uint32_t words = (uint32_t)nb * (uint32_t)nw;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        if ((((uintptr_t)dst) & 15u) == 0) {
            const uint32_t whole = words & ~3u;
            for (uint32_t g = 4u * threadIdx.x; g < whole; g += 4u * BLOCK)
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(s_words + g), reinterpret_cast<u32x4*>(dst + g));
            if (threadIdx.x < words - whole) dst[whole + threadIdx.x] = s_words[whole + threadIdx.x];
        } else
            for (uint32_t g = threadIdx.x; g < words; g += BLOCK) dst[g] = s_words[g];
    }
}

// BASELINE config 4.  One workgroup = one wave = ENUM_BOARDS (6) boards x 10 lanes; lane (board, xi) places the piece in column xi for
// the four rotations in turn.  Phase A: a board's ten lanes fetch its ten columns (ONE global load per word and board) and its piece
// word.  Phase B: they fill its BoardPre in LDS (tetris_kernels.h): band window (LDS atomics), depth strip (byte writes), prefix /
// suffix ANDs.  Phase C: four placements per lane from ~10 LDS reads each.  16 384 boards = 2 731 waves, all resident at once;
// the phases are ordered inside the wave (no workgroup barrier).
// PLANAR: rotation-minor planes — valid / land_y / cleared [n][10][4], after [10][n][10][4] (column plane c, game, column index,
// rotation) — so that a lane's four placements are one 4-byte / 16-byte store; else the layouts tetris_enumerate_drops documents
// ([n][4][10] and [n][4][10][10]).
template <int P, bool PLANAR>
__global__ __launch_bounds__(ENUM_BLOCK) void k_enumerate(Geo geo, int n, const int32_t* idx, const uint8_t* player, int H,
                                                          uint8_t* valid, int8_t* land_y, uint8_t* cleared, uint32_t* after) {
    __shared__ __attribute__((aligned(16))) uint32_t s_pre[ENUM_BOARDS][PRE_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t s_shapes[SHAPE_WORDS];
    const int tid = threadIdx.x, b = tid / 10, j = tid - b * 10;
    const int i = blockIdx.x * ENUM_BOARDS + b;                  // board of this lane
    const bool live = b < ENUM_BOARDS && i < n;
    uint32_t* pre = s_pre[b < ENUM_BOARDS ? b : 0];
    const uint32_t floor_bits = ~0u << H;
    if (tid < SHAPE_WORDS) s_shapes[tid] = d_shape_table.s[tid];
    uint32_t mine = 0;
    if (live) {
        const Ref br = board_ref(geo, safe_player(player, i, P), safe_slot(idx, i, (int)geo.n_games));
        mine = word_at(br, W_COL0 + j);
        pre[PRE_COL + j] = mine;
        if (j == 0) {
            pre[PRE_PIECE] = word_at(br, W_PIECE);
            pre[PRE_BAND] = 0xFFu; pre[PRE_BAND + 1] = 0xFFFF0000u;
        }
        if (j >= 1 && j < 5) pre[PRE_STRIP + (j - 1)] = 0u;
    }
    if (ENUM_ONE_WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads();
    if (live) {
        int word;
        const uint32_t bits = pre_band_bits(mine, floor_bits, j, word);
        atomicOr(&pre[PRE_BAND + word], bits);
        ((uint8_t*)(pre + PRE_STRIP))[j + 2] = (uint8_t)pre_depth(mine, floor_bits);
        pre[PRE_PRE + j] = pre_and_below(pre + PRE_COL, j);
        pre[PRE_SUF + j] = pre_and_from(pre + PRE_COL, j);
        if (j == 0) { pre[PRE_PRE + NCOL] = pre_and_below(pre + PRE_COL, NCOL); pre[PRE_SUF + NCOL] = ~0u; }
    }
    if (ENUM_ONE_WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads();
    if (!live) return;
    const ColumnCtx cc = enum_column(pre, j);
    if (PLANAR) {
        // rotation-minor planes: the lane's four placements are adjacent, so each result array takes ONE 4-byte store per lane and
        // each afterstate column ONE 16-byte store (13 stores per lane instead of 52; a wave's store = 1 KB contiguous)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        Placement pl[4];
        uint32_t v_ok = 0, v_y = 0, v_cl = 0;
        TE_UNROLL
        for (int r = 0; r < 4; r++) {
            pl[r] = enum_place(pre, cc, s_shapes, H, r);
            v_ok |= (uint32_t)pl[r].ok << (8 * r);
            v_y |= ((uint32_t)pl[r].y & 0xFFu) << (8 * r);
            v_cl |= (uint32_t)pl[r].cleared << (8 * r);
        }
        const size_t t4 = (size_t)i * 10 + j;                     // element (game i, column index j) of the [n][10][4] arrays
        reinterpret_cast<uint32_t*>(valid)[t4] = v_ok;
        reinterpret_cast<uint32_t*>(land_y)[t4] = v_y;
        reinterpret_cast<uint32_t*>(cleared)[t4] = v_cl;
        if (after) {
            u32x4* out = reinterpret_cast<u32x4*>(after);
            const size_t plane = (size_t)n * 10;
            for (int c = 0; c < NCOL; c++) {
                const uint32_t bc = pre[PRE_COL + c];
                u32x4 v;
                v.x = enum_after_col(bc, pl[0], c); v.y = enum_after_col(bc, pl[1], c);
                v.z = enum_after_col(bc, pl[2], c); v.w = enum_after_col(bc, pl[3], c);
                __builtin_nontemporal_store(v, &out[(size_t)c * plane + t4]);
            }
        }
    } else {
        uint32_t board[NCOL];                                   // the board's columns, read once for the four afterstates
        if (after)
            for (int c = 0; c < NCOL; c++) board[c] = pre[PRE_COL + c];
        TE_UNROLL
        for (int r = 0; r < 4; r++) {
            const Placement pl = enum_place(pre, cc, s_shapes, H, r);
            const size_t t = ((size_t)i * 4 + r) * 10 + j;
            valid[t] = (uint8_t)pl.ok;
            land_y[t] = (int8_t)pl.y;
            cleared[t] = (uint8_t)pl.cleared;
            if (after)
                for (int c = 0; c < NCOL; c++) after[t * NCOL + c] = enum_after_col(board[c], pl, c);      // 40 contiguous bytes per lane: left to the L2 to merge
        }
    }
}

template <int P, int MODE>
__global__ __launch_bounds__(64) void k_step_observe(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_step_obs[];     // [SHAPE_WORDS] shape table, then the 64 x nw tile
    uint32_t* s_shapes = s_step_obs;
    uint32_t* tile = s_step_obs + SHAPE_WORDS;
    const int lane = threadIdx.x, first = blockIdx.x * 64, i = first + lane;
    const bool active = i < a.n;
    LaneCounters cnt = {0, 0, 0, 0};
    const uint32_t shape_word = d_shape_table.s[lane];
    Game<P> g;
    if (active) game_load<P, MODE, false>(a, i, g);
    s_shapes[lane] = shape_word;
    __builtin_amdgcn_wave_barrier();
    if (active) game_run<P, MODE, false>(a, i, s_shapes, g, cnt);
    observe_emit<P>(a, g, first, lane, active, tile);
}

template <int P>
__global__ __launch_bounds__(64) void k_actions(Geo geo, int n, const int32_t* idx, const uint8_t* player,
                                                int H, uint8_t* count, uint8_t* lens, uint8_t* keys, int max_lists, int max_keys,
                                                uint32_t* status) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (size_t)n * 40)
        actions_body<P>(geo, t, idx, player, H, d_shape_table.s, count, lens, keys, max_lists, max_keys, status);
}

__global__ __launch_bounds__(256) void k_snapshot(Geo geo, int n, const int32_t* idx, uint32_t* blob, int restore) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (size_t)n * (NGWORDS + geo.P * geo.nw)) snapshot_body(geo, t, idx, blob, restore);
}

__global__ __launch_bounds__(256) void k_set_dead(Geo geo, int n, const int32_t* idx, const uint8_t* dead /*[n][P]*/) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * geo.P) set_dead_body(geo, t, idx, dead);
}

// ============================================================================ host side

struct tetris_batch;
static void chain_release(tetris_batch* b);
static int chain_recover(tetris_batch* b);
static thread_local std::string g_err;
static int default_direct_min();
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return fail(TETRIS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

// RNG tables, shared by every batch on the same device with the same piece map
struct Tables {
    int device = -1;
    uint8_t map[8] = {0};
    int only_sz = 0;
    int refs = 0;
    uint32_t* d_mt = nullptr;        // [624][65536]
    float* d_w = nullptr;            // [7][65536]
    uint64_t* d_start = nullptr;     // [65536] per-seed start entries
    double* d_pow = nullptr;         // [256]
    uint8_t* d_table = nullptr;      // [(chunk * 65536 + seed) * 624 + r], capacity `cap_chunks`
    int n_chunks = 0, cap_chunks = 0;
    std::vector<uint8_t*> retired;   // outgrown tables: other streams may still be reading them
};
static const size_t CHUNK_BYTES = (size_t)65536 * CHUNK;
static std::mutex g_tab_mutex;
static std::vector<Tables*> g_tables;

static int tables_extend(Tables* t, hipStream_t stream) {
    if (t->n_chunks >= MAX_CHUNKS) return fail(TETRIS_E_STREAM, "RNG tables: MAX_CHUNKS reached");
    if (t->n_chunks == t->cap_chunks) {            // grow: new allocation, copy, retire the old one
        int cap = t->cap_chunks ? t->cap_chunks * 2 : 2;
        if (cap > MAX_CHUNKS) cap = MAX_CHUNKS;
        uint8_t* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, CHUNK_BYTES * cap));
        if (t->n_chunks) HIP_TRY(hipMemcpyAsync(d, t->d_table, CHUNK_BYTES * t->n_chunks, hipMemcpyDeviceToDevice, stream));
        if (t->d_table) t->retired.push_back(t->d_table);
        t->d_table = d;
        t->cap_chunks = cap;
    }
    MapArg m;
    memcpy(m.m, t->map, 8);
    hipLaunchKernelGGL(k_gen_chunk, dim3(256), dim3(256), 0, stream, t->d_mt, t->d_w, t->d_table + CHUNK_BYTES * t->n_chunks,
                       t->d_start, t->n_chunks, m, t->only_sz);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    t->n_chunks++;
    return TETRIS_OK;
}

static void tables_free(Tables* t) {
    (void)hipFree(t->d_mt); (void)hipFree(t->d_w); (void)hipFree(t->d_start); (void)hipFree(t->d_pow);
    (void)hipFree(t->d_table);
    for (uint8_t* c : t->retired) (void)hipFree(c);
    delete t;
}

static int tables_build(Tables* t, hipStream_t stream);

static int tables_acquire(Tables** out, int device, const uint8_t map[7], hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    for (Tables* t : g_tables)
        if (t->device == device && memcmp(t->map, map, 7) == 0) { t->refs++; *out = t; return TETRIS_OK; }
    Tables* t = new (std::nothrow) Tables();
    if (!t) return fail(TETRIS_E_HIP, "out of host memory");
    t->device = device;
    memcpy(t->map, map, 7);
    t->only_sz = 1;                                             // PythonHandle.h:116-121 set_pieces
    for (int i = 0; i < 7; i++) if (map[i] != 2 && map[i] != 3) t->only_sz = 0;
    int rc = tables_build(t, stream);
    if (rc) { std::string keep = g_err; tables_free(t); return fail(rc, keep); }
    t->refs = 1;
    g_tables.push_back(t);
    *out = t;
    return TETRIS_OK;
}

static int tables_build(Tables* t, hipStream_t stream) {
    HIP_TRY(hipMalloc((void**)&t->d_mt, (size_t)624 * 65536 * 4));
    HIP_TRY(hipMalloc((void**)&t->d_w, (size_t)7 * 65536 * 4));
    HIP_TRY(hipMalloc((void**)&t->d_start, 65536 * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void**)&t->d_pow, 256 * sizeof(double)));
    double powtab[256];
    for (int c = 0; c < 256; c++) powtab[c] = pow((double)c, 1.4 + (double)c * 0.01);   // Combo.cpp:41, host libm
    HIP_TRY(hipMemcpyAsync(t->d_pow, powtab, sizeof powtab, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_gen_seed, dim3(256), dim3(256), 0, stream, t->d_mt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    for (int c = 0; c < 2; c++) {
        int rc = tables_extend(t, stream);
        if (rc) return rc;
    }
    return TETRIS_OK;
}

static void tables_release(Tables* t) {
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    if (--t->refs > 0) return;
    for (size_t i = 0; i < g_tables.size(); i++)
        if (g_tables[i] == t) { g_tables.erase(g_tables.begin() + i); break; }
    tables_free(t);
}

// grow-on-demand device + pinned-host staging pair
struct Stage {
    void* h = nullptr;
    void* d = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return TETRIS_OK;
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr; cap = 0;
        HIP_TRY(hipHostMalloc(&h, want, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&d, want));
        cap = want;
        return TETRIS_OK;
    }
    void release() {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr; cap = 0;
    }
};

// 120: the host may be 241 launches (~1 ms of GPU work at 64k boards) ahead of what it has seen finish.  With 32 (round 2: 65
// launches, 0.26 ms) a host thread that loses its core for a fraction of a millisecond — other tenants' jobs share the box's CPUs —
// starves the GPU: 8192-launch runs measured 4.2-5.1 us per launch in bad minutes against 4.02-4.21 with 120, and 4.02 either way in
// good ones (profiles/r03/gate_depth_ab.txt).  The low-water margin of the RNG tables is sized from it (< one chunk of 624 draws).
#ifndef TE_GATE_GROUP
#define TE_GATE_GROUP 120
#endif
#include "tetris_aql.h"

struct tetris_batch {
    int device = 0, N = 0, P = 0, H = 0;
    int stride = 0;                      // games per row of the state arrays: N + padding (see create_impl)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t* d_state = nullptr;
    uint32_t* d_gstate = nullptr;
    uint32_t* flags = nullptr;           // te::Flag words: pinned, host-coherent, written by kernels with plain stores
    unsigned long long* d_counters = nullptr;
    unsigned long long* h_counters = nullptr;   // pinned [8]
    Tables* tab = nullptr;
    uint32_t margin = 2 * (2 * TE_GATE_GROUP + 2) + 16;      // low-water mark of the RNG tables (draws); covers the launches in flight (gate below)
    uint32_t game_offset = 0;
    int split = 0, side = 0;
    int tint = 0, nw = NWORDS;           // colour planes tracked; words per player-board
    int use_duo = 1;                     // two-player rollout / step_rt through k_duo (TETRIS_NO_DUO=1 in the environment: k_game<2>)
    uint32_t* d_shadow = nullptr;        // split mode, side 1
    hipStream_t own_stream = nullptr;
    // chained launches (k_chain): two extra streams, one epoch word per wave, the number of the last chained launch
    hipStream_t chain_stream[CHAIN_STREAMS] = {};     // the device's (device_chain_streams): shared with its other batches, never destroyed
    hipEvent_t chain_ev[CHAIN_STREAMS + 1] = {};      // [k]: end of chain stream k's last launch (join); [CHAIN_STREAMS]: fork from the batch's stream
    hipEvent_t worker_gate_ev[CHAIN_STREAMS][2] = {}; // run-ahead gates of the per-stream enqueue threads (tetris_rollout_launch)
    uint32_t* d_chain = nullptr;
    uint32_t chain_epoch = 0;
    int use_chain = 1;                   // TETRIS_NO_CHAIN=1 in the environment: every rollout launch on the batch's one stream
    int use_graph = 0;                   // TETRIS_GRAPH=1: un-chained rollout launches replayed from HIP graphs (profiling aid)
    long long chain_capacity = -1;       // wave slots of the device for the chained kernel (computed on first use)
    int chain_depth = 0;                 // launches in flight = streams rotated over (3, or 2 where only two launches fit; 0: not chained)
    bool chain_pending = false;          // chained launches were enqueued since the last drain
    uint32_t chain_spin_limit = CHAIN_SPIN_LIMIT;     // polls before a waiting wave gives up (tetris_set_chain_spin_limit)
    // the chained call in progress (or the last one): what chain_recover needs to finish abandoned games un-chained
    struct { uint32_t epoch0 = 0; uint64_t first_step = 0; int steps_per_launch = 0; uint32_t policy_seed = 0; int ms = 0; } chain_call;
    bool chain_fell_back = false;        // a chained call was finished un-chained since the last tetris_take_errors (TETRIS_ERR_CHAIN_FELL_BACK)
    uint32_t* h_chain = nullptr;         // host copy of the epoch words (chain_recover)
    hipStream_t stall_stream = nullptr;  // tetris_debug_stall(.., -1, ..)
    // direct dispatch of the chained launches (tetris_aql.h): queues of the batch's own; off when anything about it failed
    int direct_min = 16;                 // calls of at least this many launches go through the device's own queues; 0: never (TETRIS_DIRECT=0 / TETRIS_DIRECT_MIN, tetris_set_direct_dispatch)
    bool last_direct = false;            // the last rollout call went through those queues
    bool direct_used = false;            // ... and so did some call of this batch (its destruction waits for the queues)
    int use_affine = 1;                  // XCD-affine launches (k_chain_affine) where the device's queues allow them; TETRIS_AFFINE=0 / tetris_set_xcd_affine
    bool last_affine = false;            // the last rollout call ran them
    uint32_t xcd_skew = 0;               // test aid (tetris_debug_xcd_skew): added to the queues' measured start XCDs
    int affine_failures = 0;             // calls in which a workgroup found itself misplaced (three: the affine form is switched off)
    bool home_async = false;             // asynchronous (_dev) work was enqueued on the batch's stream since the last drain
    bool busy = true;                    // something was enqueued on one of the batch's streams since the last drain
    // Run-ahead gate of the asynchronous entry points: every GATE_GROUP launches an event is recorded; before a new group is
    // enqueued the host waits for the event of the group before the previous one.  At most 2 * GATE_GROUP + 1 launches are
    // therefore in flight whose flag words the host has not seen; `margin` is sized for that many steps.
    hipEvent_t gate_ev[2] = {nullptr, nullptr};
    int gate_count = 0;                  // launches since the last recorded event
    int gate_slot = 0;                   // event to record next
    int gate_pending[2] = {0, 0};        // event has been recorded and not waited for
    Stage s_idx, s_in0, s_in1, s_in2, s_out0, s_out1, s_out2, s_big, s_act0, s_act1, s_act2;
};
// (TETRIS_GATE_GROUP in the environment: experiment knob, 8..120)
static const int GATE_GROUP = [] { const char* e = getenv("TETRIS_GATE_GROUP"); const int v = e ? atoi(e) : TE_GATE_GROUP; return v < 8 ? 8 : (v > 120 ? 120 : v); }();

static Geo geo_of_batch(tetris_batch* b) {
    Geo g = {b->d_state, b->d_gstate ? b->d_gstate : b->d_state, (size_t)b->N, b->P, b->nw, (size_t)b->stride};
    return g;
}

static KArgs base_args(tetris_batch* b, int n, const int32_t* d_idx) {
    KArgs a;
    memset(&a, 0, sizeof a);
    a.state = b->d_state; a.gstate = b->d_gstate ? b->d_gstate : b->d_state; a.status = b->flags;      // (tiled layout: one allocation)
    {   // tables are shared between batches: take pointer and size together (another batch may be growing them)
        std::lock_guard<std::mutex> lock(g_tab_mutex);
        a.table = b->tab->d_table;
        a.n_draws = (uint32_t)b->tab->n_chunks * CHUNK;
    }
    a.start = b->tab->d_start; a.combo_pow = b->tab->d_pow; a.margin = b->margin;
    a.H = b->H; a.n_games = b->N; a.n_stride = b->stride; a.n_players = b->P; a.nw = b->nw; a.n = n; a.idx = d_idx; a.game_offset = b->game_offset;
    return a;
}

template <int MODE>
static int launch_game(tetris_batch* b, const KArgs& a) {
    dim3 grid((unsigned)((a.n + 255) / 256)), block(256);
    if constexpr (MODE == M_ROLLOUT || MODE == M_STEP_RT || MODE == M_STEP_RT_AUTO) {
        if (b->P == 2 && !b->tint && !a.idx && b->use_duo && (MODE != M_ROLLOUT || a.steps == 1)) {
            // two-player full-batch single steps: players in adjacent half-waves (k_duo), 32 games per wave.  Measured on
            // MI355X at 64k games: 9.37 us vs 9.73 us for k_game<2> at one step per launch, but 4.4 vs 3.9 us per step when
            // 16 steps are fused, so fused rollouts stay on k_game<2>.
            hipLaunchKernelGGL((k_duo<MODE>), dim3((unsigned)((a.n + 127) / 128)), block, 0, b->stream, a);
            HIP_TRY(hipGetLastError());
            return TETRIS_OK;
        }
    }
    // (three and four players per game: the same kernel with all players of a game in one lane; it spills, and nothing is tuned for it)
    if (b->P == 1 && !b->tint) hipLaunchKernelGGL((k_game<1, MODE, false>), grid, block, 0, b->stream, a);
    else if (b->P == 1) hipLaunchKernelGGL((k_game<1, MODE, true>), grid, block, 0, b->stream, a);
    else if (b->P == 2 && !b->tint) hipLaunchKernelGGL((k_game<2, MODE, false>), grid, block, 0, b->stream, a);
    else if (b->P == 2) hipLaunchKernelGGL((k_game<2, MODE, true>), grid, block, 0, b->stream, a);
    else if (tetris_launch_game_multi(b->P, b->tint, MODE, grid, block, b->stream, a)) return fail(TETRIS_E_ARG, "no kernel for this player count / mode");
    HIP_TRY(hipGetLastError());
    return TETRIS_OK;
}

// Chained launches are deadlock-free only if the waves of all launches in flight can be resident together: the waves of
// launch E spin (in their slots) until the same waves of launch E - 1 have published, so E - 1 must never be short of a slot
// because of them.  At most `chain_depth` launches are in flight — the streams rotated over: CHAIN_STREAMS (3), or 2 where only two
// launches fit (E follows E - depth on its stream, so the oldest launch in flight never waits for an epoch: its predecessor has
// completed).  The occupancy API can be
// one workgroup per CU too high for kernels of this SGPR count (MI355X_MICROARCH.md, correctness boundaries): one is
// subtracted.  64k single-player boards: 1 024 waves per launch, 15 x 256 slots.  64k two-player boards (k_duo, 220
// VGPRs: 8 waves per CU): 2 048 waves per launch, 7 x 256 slots — does not fit, those launches stay on one stream.
static const void* chain_kernel(tetris_batch* b, long long* waves) {
    if (b->P == 1) { *waves = ((long long)b->N + CHAIN_LANES - 1) / CHAIN_LANES; return (const void*)k_chain<1>; }
    *waves = ((long long)b->N + 31) / 32;
    return (const void*)k_duo<M_ROLLOUT, true>;
}
// `waves` workgroups of the chained kernel resident at once?  Asked of the device itself: one census launch (chain_census).
static bool chain_census_ok(tetris_batch* b, long long waves) {
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, 2 * CHAIN_STRIDE * sizeof(uint32_t)) != hipSuccess) return false;
    bool ok = false;
    uint32_t h[2] = {0, 1};
    KArgs a;
    memset(&a, 0, sizeof a);
    a.steps = -1; a.chain = d; a.epoch = (uint32_t)waves; a.chain_spin_limit = 4000;       // a few ms at most, and only where the answer is no
    if (hipMemsetAsync(d, 0, 2 * CHAIN_STRIDE * sizeof(uint32_t), b->own_stream) == hipSuccess) {
        if (b->P == 1) hipLaunchKernelGGL((k_chain<1>), dim3((unsigned)waves), dim3(64), 0, b->own_stream, a);
        else hipLaunchKernelGGL((k_duo<M_ROLLOUT, true>), dim3((unsigned)waves), dim3(64), 0, b->own_stream, a);
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&h[0], d, 4, hipMemcpyDeviceToHost, b->own_stream) == hipSuccess &&
            hipMemcpyAsync(&h[1], d + CHAIN_STRIDE, 4, hipMemcpyDeviceToHost, b->own_stream) == hipSuccess &&
            hipStreamSynchronize(b->own_stream) == hipSuccess)
            ok = h[0] == (uint32_t)waves && h[1] == 0;
    }
    (void)hipFree(d);
    return ok;
}

static bool chain_fits(tetris_batch* b) {
    long long waves = 0;
    const void* fn = chain_kernel(b, &waves);
    // TETRIS_CHAIN_DEPTH=1..3 (measurement aid): at most that many launches in flight.  1 = the chained kernel on ONE stream: its
    // dispatches are then serialised by the stream — the reference point for per-dispatch PMC counters (profiles/pmc_summary.py).
    static const int cap = [] { const char* e = getenv("TETRIS_CHAIN_DEPTH"); const int v = e ? atoi(e) : CHAIN_STREAMS; return v < 1 ? 1 : (v > CHAIN_STREAMS ? CHAIN_STREAMS : v); }();
    if (b->chain_capacity < 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess) per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) != hipSuccess) cus = 0;
        // The API's answer less one workgroup per CU is taken on trust (the API can be one too high: MI355X_MICROARCH.md, correctness
        // boundaries).  A batch that would chain only WITH that last workgroup per CU — 64k two-player boards: two launches of
        // 2 048 waves are exactly the 16 x 256 slots a 116-register kernel has — asks the device: a census launch of that many
        // waves of the very kernel, once per batch.
        b->chain_capacity = per_cu > 1 ? (long long)(per_cu - 1) * cus : 0;
        const long long full = (long long)per_cu * cus;
        int want = 0;
        for (int d = cap; d >= 2 && !want; d--)
            if (d * waves <= full) want = d;
        if (want && want * waves > b->chain_capacity && b->stream == b->own_stream && chain_census_ok(b, want * waves)) b->chain_capacity = want * waves;
    }
    b->chain_depth = 0;
    for (int d = cap; d >= 1 && !b->chain_depth; d--)
        if (d * waves <= b->chain_capacity) b->chain_depth = d;
    return b->chain_depth >= (cap == 1 ? 1 : 2);
}

// Looks at the flag words WITHOUT enqueuing or waiting for anything: answers a pending "extend the RNG tables" request
// (the generation kernel goes on the batch's stream, i.e. before every launch made after this call).
static int service_flags(tetris_batch* b) {
    volatile uint32_t* f = b->flags;
    const uint32_t want = f[F_EXTEND];
    if (want) {
        std::lock_guard<std::mutex> lock(g_tab_mutex);
        // not yet answered (requests carry the size their kernel saw); at MAX_CHUNKS the tables stay as they are and a game
        // that outruns them is ended with ERR_STREAM
        if (want >= (uint32_t)b->tab->n_chunks * CHUNK && b->tab->n_chunks < MAX_CHUNKS) {
            int rc = tables_extend(b->tab, b->stream);
            if (rc) return rc;
            b->busy = true;
        }
        f[F_EXTEND] = 0;
    }
    return TETRIS_OK;
}

// One more asynchronous launch is about to be enqueued: bound the run-ahead (see tetris_batch) and service the flags.
static int gate_launch(tetris_batch* b, int group) {
    if (b->gate_count >= group) {
        const int k = b->gate_slot;
        if (b->gate_pending[k]) {                               // the group before the previous one must have finished
            // polled, not slept on: a thread blocked in hipEventSynchronize is woken through an interrupt — 10-20 us late at
            // best, milliseconds when the scheduler has given its core away meanwhile — and the GPU has only the launches of two
            // groups (~0.26 ms) to live on; TETRIS_GATE_BLOCK=1: the blocking wait
            static const bool block = [] { const char* e = getenv("TETRIS_GATE_BLOCK"); return e && e[0] == '1'; }();
            hipError_t qe = hipErrorNotReady;
            for (int spin = 0; !block && spin < 2000000 && qe == hipErrorNotReady; spin++) qe = hipEventQuery(b->gate_ev[k]);
            if (qe == hipErrorNotReady) qe = hipEventSynchronize(b->gate_ev[k]);
            HIP_TRY(qe);
            b->gate_pending[k] = 0;
        }
        HIP_TRY(hipEventRecord(b->gate_ev[k], b->stream));
        b->gate_pending[k] = 1;
        b->gate_slot = k ^ 1;
        b->gate_count = 0;
    }
    b->gate_count++;
    {
        bool on_chain = false;
        for (hipStream_t st : b->chain_stream) on_chain |= b->stream == st;
        if (!on_chain) b->home_async = true;
    }
    return service_flags(b);
}

// Waits for a stream by polling it for a while before falling back to the blocking wait: a blocked host thread is woken
// through an interrupt, 10-20 us after the GPU is done — as long as a whole 20-launch rollout of 64k boards.
static hipError_t drain_stream(hipStream_t st) {
    for (int spin = 0; spin < 20000; spin++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
    return hipStreamSynchronize(st);
}

// drain the batch's stream(s), then the flag words: sticky errors surface, the RNG tables are extended when a board came close to their end
// `drained`: the caller has already seen an event complete that is ordered behind everything the batch's streams held (a stream
// query makes the runtime push a marker through the queue and wait for it: ~6 us per stream even when the stream is idle)
static int finish_call(tetris_batch* b, bool drained = false) {
    if (!b->busy && !b->chain_pending && b->stream == b->own_stream) {      // drained already and nothing enqueued since: only the flag words
        int rc0 = service_flags(b);
        if (rc0) return rc0;
        if (!b->busy) return TETRIS_OK;           // (an extension would have enqueued work)
    }
    if (b->chain_pending && !drained)
        for (int k = 0; k < CHAIN_STREAMS; k++) HIP_TRY(drain_stream(b->chain_stream[k]));
    b->chain_pending = false;
    if (!drained) HIP_TRY(drain_stream(b->stream));
    b->home_async = false;
    b->busy = false;
    b->gate_count = 0; b->gate_pending[0] = b->gate_pending[1] = 0;
    volatile uint32_t* f = b->flags;
    // (F_EXHAUSTED / F_FIFO: capacity errors are confined to the games they happened in — tetris_take_errors)
    int rc = TETRIS_OK;
    // a workgroup of an affine launch was not where its queue's calibration put it (it left its games alone): that form stays off,
    // and the games it — and whoever waited for it — left behind are finished like abandoned ones
    const bool misplaced = f[F_PLACE] != 0;
    if (misplaced) {
        // The games such workgroups — and whoever waited for them — left behind are finished like abandoned ones.  A queue's start XCD
        // moves when the driver re-maps hardware queues (many queues in the process); the next call expects what this one saw.
        // Not the caller's business unless it keeps happening: chaining stays as it was, the affine form goes after three such calls.
        const int keep_chain = b->use_chain;
        const bool keep_fell_back = b->chain_fell_back;
        f[F_PLACE] = 0;
        if ((rc = chain_recover(b))) return rc;
        b->use_chain = keep_chain; b->chain_fell_back = keep_fell_back;
        if (++b->affine_failures >= 3) b->use_affine = 0;
    }
    if (f[F_CHAIN] && (rc = chain_recover(b))) return rc;       // waves of a chained launch gave up: their games are finished un-chained
    if ((rc = service_flags(b))) return rc;                     // before the argument error below: an extend request is never dropped
    if (f[F_BADARG]) {
        f[F_BADARG] = 0;
        return fail(TETRIS_E_ARG, "output capacity exceeded (max_lists / max_keys too small)");
    }
    return TETRIS_OK;
}

// every entry point starts here; `enqueues`: the call may put work on one of the batch's streams (all but the pure waits)
static int check_batch(tetris_batch* b, bool enqueues = true) {
    if (!b) return fail(TETRIS_E_ARG, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    if (enqueues) b->busy = true;
    return TETRIS_OK;
}

// copies idx to the device (or returns NULL for identity); validates range
static int stage_idx(tetris_batch* b, const int32_t* idx, int n, const int32_t** d_idx) {
    *d_idx = nullptr;
    if (n < 0 || (!idx && n > b->N)) return fail(TETRIS_E_ARG, "n out of range");
    if (!idx) return TETRIS_OK;
    for (int i = 0; i < n; i++)
        if (idx[i] < 0 || idx[i] >= b->N) return fail(TETRIS_E_ARG, "game index out of range");
    int rc = b->s_idx.ensure((size_t)n * 4 + 4);
    if (rc) return rc;
    memcpy(b->s_idx.h, idx, (size_t)n * 4);
    HIP_TRY(hipMemcpyAsync(b->s_idx.d, b->s_idx.h, (size_t)n * 4, hipMemcpyHostToDevice, b->stream));
    *d_idx = (const int32_t*)b->s_idx.d;
    return TETRIS_OK;
}

static int stage_in(tetris_batch* b, Stage& s, const void* src, size_t bytes) {
    int rc = s.ensure(bytes + 4);
    if (rc) return rc;
    memcpy(s.h, src, bytes);
    HIP_TRY(hipMemcpyAsync(s.d, s.h, bytes, hipMemcpyHostToDevice, b->stream));
    return TETRIS_OK;
}

extern "C" {

const char* tetris_last_error(void) { return g_err.c_str(); }

#if defined(TE_PHASE_TRACE)
// diagnostic build only: copies the phase stamps out (and clears them)
extern "C" int tetris_debug_trace(unsigned long long* out, int n_words) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(d_trace), (size_t)n_words * 8) != hipSuccess) return -1;
    static unsigned long long zero[2048 * 16];
    return hipMemcpyToSymbol(HIP_SYMBOL(d_trace), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
extern "C" int tetris_debug_chain_trace(unsigned long long* out /*[8][1024][8]*/) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(d_chain_trace), sizeof(unsigned long long) * 8 * 1024 * 8) == hipSuccess ? 0 : -1;
}
#endif

int tetris_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(TETRIS_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int tetris_device_name(int device, char* buf, int len) {
    if (!buf || len < 1) return fail(TETRIS_E_ARG, "buf/len");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return TETRIS_OK;
}

int tetris_record_size(void) { return (int)sizeof(tetris_record); }
int tetris_layout_words(void) { return NWORDS; }
int tetris_snapshot_words(const tetris_batch* b) { return b ? NGWORDS + b->P * b->nw : 0; }
int tetris_table_chunks(const tetris_batch* b) { return b && b->tab ? b->tab->n_chunks : 0; }
void* tetris_device_state(tetris_batch* b) { return b ? b->d_state : nullptr; }
void* tetris_stream(tetris_batch* b) { return b ? (void*)b->stream : nullptr; }

int tetris_destroy(tetris_batch* b) {
    if (!b) return TETRIS_OK;
    (void)hipSetDevice(b->device);
    for (hipStream_t st : b->chain_stream) if (st) (void)hipStreamSynchronize(st);      // (a call that failed half-way may have left launches there)
    chain_release(b);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->tab) tables_release(b->tab);
    (void)hipFree(b->d_shadow); (void)hipFree(b->d_state); (void)hipFree(b->d_gstate); (void)hipFree(b->d_counters);
    if (b->flags) (void)hipHostFree(b->flags);
    if (b->h_counters) (void)hipHostFree(b->h_counters);
    for (hipEvent_t e : b->gate_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->chain_ev) if (e) (void)hipEventDestroy(e);
    for (auto& pair : b->worker_gate_ev) for (hipEvent_t e : pair) if (e) (void)hipEventDestroy(e);
    (void)hipFree(b->d_chain);
    free(b->h_chain);
    if (b->stall_stream) { (void)hipStreamSynchronize(b->stall_stream); (void)hipStreamDestroy(b->stall_stream); }
    if (b->direct_used) { aql::Device* dev = aql::device_for(b->device); if (dev->ok) aql::quiesce(dev->qs); }      // (a test's idle kernel may still sit there)
    Stage* all[] = {&b->s_idx, &b->s_in0, &b->s_in1, &b->s_in2, &b->s_out0, &b->s_out1, &b->s_out2, &b->s_big, &b->s_act0, &b->s_act1, &b->s_act2};
    for (Stage* s : all) s->release();
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
    delete b;
    return TETRIS_OK;
}

// The chain streams of a device, made once and shared by all its batches (chained calls of a device's batches exclude each
// other: chain_acquire).  Per batch they cost three more streams each, and the runtime deals streams onto few hardware
// queues: a batch created after two others ran its chained 20-launch calls at 11-13 us per launch instead of 5, its streams
// sharing queues (profiles/r03/direct_dispatch.txt).
// Every chain stream gets a stream priority of its own.  Not for the priorities' sake: the runtime keeps one hardware queue per
// priority level apart from the four (GPU_MAX_HW_QUEUES) it deals ordinary streams onto in turn, and two chain streams on ONE
// hardware queue do not overlap (measured with GPU_MAX_HW_QUEUES=2: 5.07 us per launch with equal priorities, 3.99 with three
// different ones; with four queues 4.00 either way: profiles/r02/hw_queues.txt).  TETRIS_CHAIN_PRIO=0: equal priorities.
static std::mutex g_dev_streams_mutex;
static hipStream_t g_dev_streams[64][CHAIN_STREAMS] = {};
static hipError_t device_chain_streams(int device, hipStream_t out[CHAIN_STREAMS]) {
    std::lock_guard<std::mutex> lock(g_dev_streams_mutex);
    hipStream_t* st = g_dev_streams[device & 63];
    if (!st[0]) {
        int lo = 0, hi = 0;
        const char* e = getenv("TETRIS_CHAIN_PRIO");
        const bool spread = !(e && e[0] == '0') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo - hi + 1 >= CHAIN_STREAMS;
        hipStream_t made[CHAIN_STREAMS] = {};
        for (int k = 0; k < CHAIN_STREAMS; k++) {
            const hipError_t ce = spread ? hipStreamCreateWithPriority(&made[k], hipStreamNonBlocking, hi + k) : hipStreamCreateWithFlags(&made[k], hipStreamNonBlocking);
            if (ce != hipSuccess) { for (int j = 0; j < k; j++) (void)hipStreamDestroy(made[j]); return ce; }
        }
        for (int k = 0; k < CHAIN_STREAMS; k++) st[k] = made[k];
    }
    for (int k = 0; k < CHAIN_STREAMS; k++) out[k] = st[k];
    return hipSuccess;
}

static int create_impl(tetris_batch** out, int n_games, int n_players, int height, int width, const uint8_t piece_map[7],
                       int device, const int16_t* seeds, int split, int side, int flags = 0) {
    if (!out) return fail(TETRIS_E_ARG, "out is NULL");
    *out = nullptr;
    if (n_games < 1) return fail(TETRIS_E_ARG, "n_games must be >= 1");
    if (n_players < 1 || n_players > TETRIS_MAX_PLAYERS) return fail(TETRIS_E_ARG, "n_players must be 1..4");
    if (n_players > 2 && split) return fail(TETRIS_E_ARG, "split batches are two-player games");
    if ((long long)n_games * n_players > (1ll << 23))       // the state allocation stays below 4 GiB (32-bit buffer offsets): 8M boards x 69 words
        return fail(TETRIS_E_ARG, "n_games * n_players must be <= 2^23");
    if (height < 4 || height > MAX_H) return fail(TETRIS_E_ARG, "height must be in [4, 31]");
    if (width != NCOL) return fail(TETRIS_E_ARG, "width must be 10 (the reference hard-codes 10, gamePlay.cpp:202)");
    if (!piece_map) return fail(TETRIS_E_ARG, "piece_map is NULL");
    for (int i = 0; i < 7; i++)
        if (piece_map[i] > 6) return fail(TETRIS_E_ARG, "piece_map entries must be 0..6");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(TETRIS_E_HIP, std::string("no HIP device available (this library has no CPU path): ") +
                                      (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
    if (device < 0 || device >= ndev) return fail(TETRIS_E_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    tetris_batch* b = new (std::nothrow) tetris_batch();
    if (!b) return fail(TETRIS_E_HIP, "out of host memory");
    b->device = device; b->N = n_games; b->P = n_players; b->H = height;
    b->stride = n_games;
    b->tint = (flags & TETRIS_FLAG_COLOURS) ? 1 : 0;
    { const char* e = getenv("TETRIS_NO_DUO"); b->use_duo = !(e && e[0] == '1'); }
    { const char* e = getenv("TETRIS_NO_CHAIN"); b->use_chain = !(e && e[0] == '1'); }
    b->direct_min = default_direct_min();
    { const char* e = getenv("TETRIS_AFFINE"); b->use_affine = !(e && e[0] == '0'); }
    { const char* e = getenv("TETRIS_GRAPH"); b->use_graph = (e && e[0] == '1'); }
    { const char* e = getenv("TETRIS_CHAIN_SPIN_LIMIT"); if (e && atoll(e) > 0) b->chain_spin_limit = (uint32_t)atoll(e); }
    b->nw = b->tint ? NWORDS_TINT : NWORDS;
#define CREATE_TRY(expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            tetris_destroy(b);                                                              \
            return fail(TETRIS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
        }                                                                                   \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    b->own_stream = b->stream;
    CREATE_TRY(hipEventCreate(&b->ev0));
    CREATE_TRY(hipEventCreate(&b->ev1));
    {   // the chain streams belong to the DEVICE (device_chain_streams below): every batch of a device uses the same three
        hipError_t ce = device_chain_streams(device, b->chain_stream);
        CREATE_TRY(ce);
    }
    for (int k = 0; k < CHAIN_STREAMS + 1; k++) CREATE_TRY(hipEventCreateWithFlags(&b->chain_ev[k], hipEventDisableTiming));
    for (int k = 0; k < CHAIN_STREAMS; k++) for (int j = 0; j < 2; j++) CREATE_TRY(hipEventCreateWithFlags(&b->worker_gate_ev[k][j], hipEventDisableTiming));
    {
        const size_t chain_bytes = (((size_t)n_games + 15) / 16) * sizeof(uint32_t) * CHAIN_STRIDE;       // (one word per wave; at least 16 games per wave)
        CREATE_TRY(hipMalloc((void**)&b->d_chain, chain_bytes));
        CREATE_TRY(hipMemsetAsync(b->d_chain, 0, chain_bytes, b->stream));
    }
    CREATE_TRY(hipEventCreateWithFlags(&b->gate_ev[0], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&b->gate_ev[1], hipEventDisableTiming));
    const size_t state_bytes = state_words((size_t)b->stride, n_players, b->nw) * 4, gstate_bytes = gstate_words((size_t)b->stride) * 4;
    CREATE_TRY(hipMalloc((void**)&b->d_state, state_bytes));
    if (gstate_bytes) CREATE_TRY(hipMalloc((void**)&b->d_gstate, gstate_bytes));
    CREATE_TRY(hipMalloc((void**)&b->d_counters, 8 * sizeof(unsigned long long)));
    CREATE_TRY(hipHostMalloc((void**)&b->h_counters, 8 * sizeof(unsigned long long), hipHostMallocDefault));
    // flag words: host memory the GPU can write (fine-grained, so a store is visible to the host while the kernel runs)
    CREATE_TRY(hipHostMalloc((void**)&b->flags, NFLAGS * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    memset(b->flags, 0, NFLAGS * sizeof(uint32_t));
    if (gstate_bytes) CREATE_TRY(hipMemsetAsync(b->d_gstate, 0, gstate_bytes, b->stream));
    CREATE_TRY(hipMemsetAsync(b->d_state, 0, state_bytes, b->stream));
    int rc = tables_acquire(&b->tab, device, piece_map, b->stream);
    if (rc) { std::string keep = g_err; tetris_destroy(b); return fail(rc, keep); }
    const int16_t* d_seeds = nullptr;
    if (seeds) {
        rc = stage_in(b, b->s_in0, seeds, (size_t)n_games * 2);
        if (rc) { std::string keep = g_err; tetris_destroy(b); return fail(rc, keep); }
        d_seeds = (const int16_t*)b->s_in0.d;
    }
    b->split = split; b->side = side;
    if (split && side == 1) CREATE_TRY(hipMalloc((void**)&b->d_shadow, (size_t)(UNDO_WORDS + b->nw) * (size_t)b->stride * 4));
    KArgs a = base_args(b, n_games, nullptr);
    a.seeds = d_seeds;
    a.steps = side;
    rc = split ? launch_game<M_SPLIT_INIT>(b, a) : launch_game<M_INIT>(b, a);
    if (!rc) rc = finish_call(b);
    if (rc) { std::string keep = g_err; tetris_destroy(b); return fail(rc, keep); }
    *out = b;
    return TETRIS_OK;
}

int tetris_set_chained(tetris_batch* b, int on) {
    int rc = check_batch(b);
    if (rc) return rc;
    if ((rc = finish_call(b))) return rc;
    b->use_chain = on ? 1 : 0;
    return TETRIS_OK;
}

static int default_direct_min() {
    const char* off = getenv("TETRIS_DIRECT");
    if (off && off[0] == '0') return 0;
    const char* e = getenv("TETRIS_DIRECT_MIN");
    const int v = e ? atoi(e) : 16;
    return v < 1 ? 16 : v;
}

int tetris_set_direct_dispatch(tetris_batch* b, int min_launches) {
    int rc = check_batch(b);
    if (rc) return rc;
    if ((rc = finish_call(b))) return rc;
    b->direct_min = min_launches < 0 ? default_direct_min() : min_launches;
    return TETRIS_OK;
}

int tetris_set_xcd_affine(tetris_batch* b, int on) {
    int rc = check_batch(b);
    if (rc) return rc;
    if ((rc = finish_call(b))) return rc;
    b->use_affine = on ? 1 : 0;
    return TETRIS_OK;
}

int tetris_debug_xcd_skew(tetris_batch* b, int skew) {
    int rc = check_batch(b);
    if (rc) return rc;
    if ((rc = finish_call(b))) return rc;
    b->xcd_skew = (uint32_t)skew & 7u;
    return TETRIS_OK;
}

int tetris_debug_code_objects(int* count, uint64_t* bytes) {
    if (!count || !bytes) return fail(TETRIS_E_ARG, "count / bytes is NULL");
    std::vector<std::vector<char>> images;
    std::string why;
    *count = 0; *bytes = 0;
    if (!aql::own_code_objects(images, why)) return TETRIS_OK;      // (none: the caller sees count == 0)
    *count = (int)images.size();
    for (auto& img : images) *bytes += img.size();
    return TETRIS_OK;
}

int tetris_rollout_was_direct(tetris_batch* b) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    return b->last_direct ? (b->last_affine ? 2 : 1) : 0;
}

int tetris_set_chain_spin_limit(tetris_batch* b, uint32_t polls) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if ((rc = finish_call(b))) return rc;
    b->chain_spin_limit = polls ? polls : CHAIN_SPIN_LIMIT;
    return TETRIS_OK;
}

// Test aid (tests of the give-up path of chained launches; nothing in the product calls it): enqueues a kernel that does nothing for
// `microseconds` — which = 0..2: on that chain stream (the launches the next rollout puts there start late); 3: on the batch's
// stream; -1: on a stream of its own, as 1024-thread workgroups that take `percent` % of the device's wave slots meanwhile.
int tetris_debug_stall(tetris_batch* b, int which, int microseconds, int percent) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (which < -1 || which > 3 || microseconds < 0 || microseconds > 2000000 || percent < 0 || percent > 100) return fail(TETRIS_E_ARG, "which / microseconds / percent");
    const unsigned long long ticks = (unsigned long long)microseconds * 100ull;
    if (which >= 0 && which < 3 && b->direct_min > 0) {
        // the device's own queues (tetris_aql.h), if they exist: the idle kernel goes to the one that stands for that stream as well —
        // whichever way the next call's launches go, the ones with that number start late
        aql::Device* dev = aql::device_for(b->device);
        std::string why;
        if (dev->ok && aql::make_queues(dev, why) && dev->blocker.ok) {
            struct { const uint32_t* go; unsigned long long ticks; } args = {nullptr, ticks};
            aql::Pending pd;
            aql::write_dispatch(dev->qs, pd, which, dev->blocker, &args, sizeof args, 1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, hsa_signal_t{});
            aql::ring(dev->qs, pd);
            b->direct_used = true;
        }
    }
    if (which >= 0) {
        hipStream_t st = which == 3 ? b->stream : b->chain_stream[which % CHAIN_STREAMS];
        if (which < 3) b->chain_pending = true;
        hipLaunchKernelGGL(k_blocker, dim3(1), dim3(64), 0, st, (const uint32_t*)nullptr, ticks);
    } else {
        if (!b->stall_stream) HIP_TRY(hipStreamCreateWithFlags(&b->stall_stream, hipStreamNonBlocking));
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device));
        const int blocks = (int)((long long)cus * 2 * percent / 100);        // 2 x 1024 threads = the 32 wave slots of a CU
        if (blocks > 0) hipLaunchKernelGGL(k_blocker, dim3((unsigned)blocks), dim3(1024), 0, b->stall_stream, (const uint32_t*)nullptr, ticks);
    }
    HIP_TRY(hipGetLastError());
    return TETRIS_OK;
}

// Measurement aid: the GPU's shader clock right now, in kHz (a 50 us probe kernel on the batch's stream; synchronous).  The
// chained period follows it: DESIGN.md, section 6.
int tetris_debug_clock_khz(tetris_batch* b, int* khz) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!khz) return fail(TETRIS_E_ARG, "khz is NULL");
    HIP_TRY(hipMemsetAsync(b->d_counters + 4, 0, 2 * sizeof(unsigned long long), b->stream));
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, b->stream, 5000ull, b->d_counters + 4);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_counters + 4, b->d_counters + 4, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
    if ((rc = finish_call(b))) return rc;
    *khz = b->h_counters[5] ? (int)(b->h_counters[4] * 100000ull / b->h_counters[5]) : 0;
    return TETRIS_OK;
}

int tetris_set_game_offset(tetris_batch* b, uint64_t first_game_id) {
    if (!b) return fail(TETRIS_E_ARG, "null batch");
    b->game_offset = (uint32_t)first_game_id;
    return TETRIS_OK;
}

int tetris_create(tetris_batch** out, int n_games, int n_players, int height, int width, const uint8_t piece_map[7],
                  int device, const int16_t* seeds) {
    return create_impl(out, n_games, n_players, height, width, piece_map, device, seeds, 0, 0);
}

int tetris_create_ex(tetris_batch** out, int n_games, int n_players, int height, int width, const uint8_t piece_map[7], int device,
                     const int16_t* seeds, int flags) {
    if (flags & ~TETRIS_FLAG_COLOURS) return fail(TETRIS_E_ARG, "unknown flag");
    return create_impl(out, n_games, n_players, height, width, piece_map, device, seeds, 0, 0, flags);
}

int tetris_create_split(tetris_batch** out, int n_games, int side, int height, int width, const uint8_t piece_map[7], int device,
                        const int16_t* seeds) {
    if (side != 0 && side != 1) return fail(TETRIS_E_ARG, "side must be 0 or 1");
    return create_impl(out, n_games, 1, height, width, piece_map, device, seeds, 1, side);
}

int tetris_set_stream(tetris_batch* b, void* hip_stream, int external) {
    int rc = check_batch(b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->stream = external ? (hipStream_t)hip_stream : b->own_stream;      // NULL + external = the legacy default stream
    return TETRIS_OK;
}

static int split_stage_launch(tetris_batch* b, int stage, KArgs& a, const uint32_t* const d_words[4], uint32_t* d_out) {
    if (!b->split) return fail(TETRIS_E_ARG, "not a split batch (tetris_create_split)");
    if (stage < 0 || stage > 3) return fail(TETRIS_E_ARG, "stage must be 0, 1, 2 or 3 (= 2 of this step + 0 of the next)");
    if (stage != 2 && !d_out) return fail(TETRIS_E_ARG, "stages 0, 1 and 3 need d_out");
    // words a stage reads: stage 1 = both A words (+ player 0's B on side 1); stages 2 and 3 = the opponent's B
    if (stage > 0 && !d_words) return fail(TETRIS_E_ARG, "stages 1, 2 and 3 need d_words");
    if (stage == 1 && (!d_words[0] || !d_words[1] || (b->side == 1 && !d_words[2]))) return fail(TETRIS_E_ARG, "stage 1 needs both A words (and player 0's B words on side 1)");
    if (stage >= 2 && !d_words[b->side == 0 ? 3 : 2]) return fail(TETRIS_E_ARG, "stages 2 and 3 need the opponent's B words");
    b->home_async = true;
    for (int k = 0; k < 4; k++) a.xw[k] = d_words ? d_words[k] : nullptr;
    a.shadow = b->d_shadow; a.xout = d_out; a.split_side = b->side;
    dim3 grid((unsigned)((b->N + 255) / 256)), block(256);
    if (stage == 0) hipLaunchKernelGGL((k_split<0, false>), grid, block, 0, b->stream, a);
    else if (stage == 1) hipLaunchKernelGGL((k_split<1, false>), grid, block, 0, b->stream, a);
    else if (stage == 2) hipLaunchKernelGGL((k_split<2, false>), grid, block, 0, b->stream, a);
    else hipLaunchKernelGGL((k_split<3, false>), grid, block, 0, b->stream, a);
    HIP_TRY(hipGetLastError());
    return TETRIS_OK;
}

int tetris_split_stage_dev(tetris_batch* b, int stage, const uint8_t* d_rot, const uint8_t* d_trans, const uint8_t* d_acting, int ms,
                           const uint32_t* const d_words[4], uint32_t* d_out, uint8_t* d_done, uint8_t* d_lines, uint8_t* d_dead) {
    int rc = check_batch(b);
    if (rc) return rc;
    if ((stage == 0 || stage == 3) && (!d_rot || !d_trans)) return fail(TETRIS_E_ARG, "stages 0 and 3 need rot/trans (stage 3: of the NEXT step)");
    KArgs a = base_args(b, b->N, nullptr);
    a.rot = d_rot; a.trans = d_trans; a.player = d_acting; a.ms = ms;
    a.done = d_done; a.lines = d_lines; a.dead = d_dead;
    return split_stage_launch(b, stage, a, d_words, d_out);
}

int tetris_split_rollout_stage_dev(tetris_batch* b, int stage, uint32_t policy_seed, uint64_t step, int ms, const uint32_t* const d_words[4],
                                   uint32_t* d_out) {
    int rc = check_batch(b);
    if (rc) return rc;
    KArgs a = base_args(b, b->N, nullptr);
    a.ms = ms; a.policy_seed = policy_seed; a.first_step = step; a.steps = 1;
    return split_stage_launch(b, stage, a, d_words, d_out);
}

int tetris_rollout_totals(tetris_batch* b, uint64_t totals[4]) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!totals) return fail(TETRIS_E_ARG, "totals is NULL");
    HIP_TRY(hipMemsetAsync(b->d_counters, 0, 8 * sizeof(unsigned long long), b->stream));
    const int tot_blocks = b->N >= 65536 ? 64 : (b->N + 1023) / 1024;
    hipLaunchKernelGGL(k_totals, dim3(tot_blocks), dim3(256), 0, b->stream, geo_of_batch(b), b->d_counters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_counters, b->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
    if ((rc = finish_call(b))) return rc;
    for (int k = 0; k < 4; k++) totals[k] = b->h_counters[k];
    return TETRIS_OK;
}

int tetris_take_errors(tetris_batch* b, uint32_t* bits) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!bits) return fail(TETRIS_E_ARG, "bits is NULL");
    if ((rc = finish_call(b))) return rc;
    volatile uint32_t* f = b->flags;
    *bits = (f[F_FIFO] ? TETRIS_ERR_FIFO : 0u) | (f[F_EXHAUSTED] ? TETRIS_ERR_STREAM : 0u) | (b->chain_fell_back ? TETRIS_ERR_CHAIN_FELL_BACK : 0u);
    f[F_FIFO] = 0; f[F_EXHAUSTED] = 0; b->chain_fell_back = false;
    return TETRIS_OK;
}

int tetris_sync(tetris_batch* b) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    return finish_call(b);
}

int tetris_reset(tetris_batch* b, const int32_t* idx, int n, const int16_t* seeds) {
    int rc = check_batch(b);
    if (rc) return rc;
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return finish_call(b);          // still a synchronisation point: sticky errors surface, tables get extended
    KArgs a = base_args(b, n, d_idx);
    if (seeds) {
        if ((rc = stage_in(b, b->s_in0, seeds, (size_t)n * 2))) return rc;
        a.seeds = (const int16_t*)b->s_in0.d;
    }
    if ((rc = b->split ? launch_game<M_SPLIT_RESET>(b, a) : launch_game<M_RESET>(b, a))) return rc;
    return finish_call(b);
}

// host keys [n][P][K] -> device [K][P][n]; host lens [n][P] -> device [P][n]
static int stage_keys(tetris_batch* b, int n, const uint8_t* keys, const uint8_t* lens, int max_keys, KArgs& a) {
    if (!keys || !lens || max_keys < 1) return fail(TETRIS_E_ARG, "keys/lens/max_keys");
    const int P = b->P;
    int rc = b->s_in0.ensure((size_t)n * P * max_keys + 4);
    if (rc) return rc;
    if ((rc = b->s_in1.ensure((size_t)n * P + 4))) return rc;
    uint8_t* hk = (uint8_t*)b->s_in0.h;
    uint8_t* hl = (uint8_t*)b->s_in1.h;
    for (int i = 0; i < n; i++)
        for (int p = 0; p < P; p++) {
            const int len = lens[(size_t)i * P + p];
            if (len > max_keys) return fail(TETRIS_E_ARG, "lens[i][p] > max_keys");
            hl[(size_t)p * n + i] = (uint8_t)len;
        }
    // [n][P][K] -> [K][P][n], blocked over games so that reads stay in L1 and every write run is contiguous
    // (the naive order writes with a stride of n bytes: 52 ms instead of ~2 ms for 64k two-player games)
    const int BLK = 512;
    for (int i0 = 0; i0 < n; i0 += BLK) {
        const int i1 = i0 + BLK < n ? i0 + BLK : n;
        for (int k = 0; k < max_keys; k++)
            for (int p = 0; p < P; p++) {
                uint8_t* dst = hk + ((size_t)k * P + p) * n;
                const uint8_t* src = keys + (size_t)p * max_keys + k;
                for (int i = i0; i < i1; i++) dst[i] = src[(size_t)i * P * max_keys];
            }
    }
    HIP_TRY(hipMemcpyAsync(b->s_in0.d, hk, (size_t)n * P * max_keys, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_in1.d, hl, (size_t)n * P, hipMemcpyHostToDevice, b->stream));
    a.keys = (const uint8_t*)b->s_in0.d; a.lens = (const uint8_t*)b->s_in1.d; a.max_keys = max_keys;
    return TETRIS_OK;
}

static int stage_outputs(tetris_batch* b, int n, KArgs& a) {
    int rc;
    if ((rc = b->s_out0.ensure((size_t)n + 4))) return rc;
    if ((rc = b->s_out1.ensure((size_t)n * b->P + 4))) return rc;
    if ((rc = b->s_out2.ensure((size_t)n * b->P + 4))) return rc;
    a.done = (uint8_t*)b->s_out0.d; a.lines = (uint8_t*)b->s_out1.d; a.dead = (uint8_t*)b->s_out2.d;
    return TETRIS_OK;
}

// device [P][n] -> host [n][P]
static int fetch_outputs(tetris_batch* b, int n, uint8_t* done, uint8_t* lines, uint8_t* dead) {
    const int P = b->P;
    HIP_TRY(hipMemcpyAsync(b->s_out0.h, b->s_out0.d, (size_t)n, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out1.h, b->s_out1.d, (size_t)n * P, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out2.h, b->s_out2.d, (size_t)n * P, hipMemcpyDeviceToHost, b->stream));
    int rc = finish_call(b);
    if (rc) return rc;
    if (done) memcpy(done, b->s_out0.h, (size_t)n);
    const uint8_t* hl = (const uint8_t*)b->s_out1.h;
    const uint8_t* hd = (const uint8_t*)b->s_out2.h;
    for (int i = 0; i < n; i++)
        for (int p = 0; p < P; p++) {
            if (lines) lines[(size_t)i * P + p] = hl[(size_t)p * n + i];
            if (dead) dead[(size_t)i * P + p] = hd[(size_t)p * n + i];
        }
    return TETRIS_OK;
}

int tetris_make_actions(tetris_batch* b, const int32_t* idx, int n, const uint8_t* keys, const uint8_t* lens, int max_keys) {
    int rc = check_batch(b);
    if (rc) return rc;
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    KArgs a = base_args(b, n, d_idx);
    if ((rc = stage_keys(b, n, keys, lens, max_keys, a))) return rc;
    if ((rc = launch_game<M_MAKE>(b, a))) return rc;
    return finish_call(b);
}

int tetris_finish_actions(tetris_batch* b, const int32_t* idx, int n, int ms, uint8_t* done, uint8_t* lines, uint8_t* dead) {
    int rc = check_batch(b);
    if (rc) return rc;
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    KArgs a = base_args(b, n, d_idx);
    a.ms = ms;
    if ((rc = stage_outputs(b, n, a))) return rc;
    if ((rc = launch_game<M_FINISH>(b, a))) return rc;
    return fetch_outputs(b, n, done, lines, dead);
}

int tetris_step_keys(tetris_batch* b, const int32_t* idx, int n, const uint8_t* keys, const uint8_t* lens, int max_keys,
                     int ms, uint8_t* done, uint8_t* lines, uint8_t* dead) {
    int rc = check_batch(b);
    if (rc) return rc;
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    KArgs a = base_args(b, n, d_idx);
    a.ms = ms;
    if ((rc = stage_keys(b, n, keys, lens, max_keys, a))) return rc;
    if ((rc = stage_outputs(b, n, a))) return rc;
    if ((rc = launch_game<M_STEP_KEYS>(b, a))) return rc;
    return fetch_outputs(b, n, done, lines, dead);
}

int tetris_step_rt_dev_ex(tetris_batch* b, const uint8_t* d_rot, const uint8_t* d_trans, const uint8_t* d_player, int ms,
                          uint8_t* d_done, uint8_t* d_lines, uint8_t* d_dead, int flags) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!d_rot || !d_trans) return fail(TETRIS_E_ARG, "rot/trans are NULL");
    if (flags & ~TETRIS_STEP_AUTO_RESET) return fail(TETRIS_E_ARG, "unknown flag");
    if ((flags & TETRIS_STEP_AUTO_RESET) && b->split) return fail(TETRIS_E_ARG, "auto-reset is not available on split batches");
    if ((rc = gate_launch(b, GATE_GROUP))) return rc;
    KArgs a = base_args(b, b->N, nullptr);
    a.rot = d_rot; a.trans = d_trans; a.player = d_player; a.ms = ms;
    a.done = d_done; a.lines = d_lines; a.dead = d_dead;
    return (flags & TETRIS_STEP_AUTO_RESET) ? launch_game<M_STEP_RT_AUTO>(b, a) : launch_game<M_STEP_RT>(b, a);
}

int tetris_step_rt_observe_dev(tetris_batch* b, const uint8_t* d_rot, const uint8_t* d_trans, const uint8_t* d_player, int ms,
                               uint8_t* d_done, uint8_t* d_lines, uint8_t* d_dead, int flags, const uint8_t* d_next_player,
                               uint8_t* d_visual, uint8_t* d_vector, uint8_t* d_piece) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!d_rot || !d_trans) return fail(TETRIS_E_ARG, "rot/trans are NULL");
    if (!d_visual || !d_vector || !d_piece) return fail(TETRIS_E_ARG, "visual/vector/piece are NULL");
    if (flags & ~TETRIS_STEP_AUTO_RESET) return fail(TETRIS_E_ARG, "unknown flag");
    if (b->split) return fail(TETRIS_E_ARG, "tetris_step_rt_observe_dev is not available on split batches");
    if (b->P > 2) return fail(TETRIS_E_ARG, "the packed observation is defined for one or two players (own / opponent's board: state_unpack.py:88-137)");
    const bool fused = !b->tint && b->H % 2 == 0 && ((uintptr_t)d_visual & 3u) == 0 && ((uintptr_t)d_vector & 3u) == 0;
    if (!fused) {             // colour batches, odd heights, unaligned outputs: the same two kernels back to back
        if ((rc = tetris_step_rt_dev_ex(b, d_rot, d_trans, d_player, ms, d_done, d_lines, d_dead, flags))) return rc;
        return tetris_observe_packed_dev(b, nullptr, b->N, d_next_player, d_visual, d_vector, d_piece);
    }
    if ((rc = gate_launch(b, GATE_GROUP))) return rc;
    KArgs a = base_args(b, b->N, nullptr);
    a.rot = d_rot; a.trans = d_trans; a.player = d_player; a.ms = ms;
    a.done = d_done; a.lines = d_lines; a.dead = d_dead;
    a.next_player = d_next_player; a.obs_visual = d_visual; a.obs_vector = d_vector; a.obs_piece = d_piece;
    const int nw = b->H * NCOL / 4;
        const size_t lds = ((size_t)SHAPE_WORDS + (size_t)64 * nw) * 4;
    const dim3 grid((unsigned)((b->N + 63) / 64)), block(64);
    const bool autoreset = (flags & TETRIS_STEP_AUTO_RESET) != 0;
    if (b->P == 1) {
        if (autoreset) hipLaunchKernelGGL((k_step_observe<1, M_STEP_RT_AUTO>), grid, block, lds, b->stream, a);
        else hipLaunchKernelGGL((k_step_observe<1, M_STEP_RT>), grid, block, lds, b->stream, a);
    } else if (b->use_duo) {
        // two players: one player per lane (k_duo) — every lane builds the one board it holds
        const size_t lds2 = ((size_t)4 * 64 * nw + 16) * 4;
        if (lds2 > 48 * 1024) {
            const void* fn = autoreset ? (const void*)k_duo<M_STEP_RT_AUTO, false, true> : (const void*)k_duo<M_STEP_RT, false, true>;
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        }
        const dim3 grid2((unsigned)((b->N + 127) / 128)), block2(256);
        if (autoreset) hipLaunchKernelGGL((k_duo<M_STEP_RT_AUTO, false, true>), grid2, block2, lds2, b->stream, a);
        else hipLaunchKernelGGL((k_duo<M_STEP_RT, false, true>), grid2, block2, lds2, b->stream, a);
    } else {
        if (autoreset) hipLaunchKernelGGL((k_step_observe<2, M_STEP_RT_AUTO>), grid, block, lds, b->stream, a);
        else hipLaunchKernelGGL((k_step_observe<2, M_STEP_RT>), grid, block, lds, b->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return TETRIS_OK;
}

int tetris_step_rt_dev(tetris_batch* b, const uint8_t* d_rot, const uint8_t* d_trans, const uint8_t* d_player, int ms,
                       uint8_t* d_done, uint8_t* d_lines, uint8_t* d_dead) {
    return tetris_step_rt_dev_ex(b, d_rot, d_trans, d_player, ms, d_done, d_lines, d_dead, 0);
}

int tetris_reset_dev(tetris_batch* b, const uint8_t* d_mask, const int16_t* d_seeds) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (b->split) return fail(TETRIS_E_ARG, "tetris_reset_dev is not available on split batches");
    if ((rc = gate_launch(b, GATE_GROUP))) return rc;
    KArgs a = base_args(b, b->N, nullptr);
    a.mask = d_mask; a.seeds = d_seeds;
    return d_seeds ? launch_game<M_RESET>(b, a) : launch_game<M_RESET_SCHED>(b, a);
}

int tetris_step_rt(tetris_batch* b, const uint8_t* rot, const uint8_t* trans, const uint8_t* player, int ms, uint8_t* done,
                   uint8_t* lines, uint8_t* dead) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!rot || !trans) return fail(TETRIS_E_ARG, "rot/trans are NULL");
    const int n = b->N;
    if (player)
        for (int i = 0; i < n; i++)
            if (player[i] >= b->P) return fail(TETRIS_E_ARG, "player index out of range");
    KArgs a = base_args(b, n, nullptr);
    a.ms = ms;
    if ((rc = stage_in(b, b->s_in0, rot, (size_t)n))) return rc;
    if ((rc = stage_in(b, b->s_in1, trans, (size_t)n))) return rc;
    a.rot = (const uint8_t*)b->s_in0.d; a.trans = (const uint8_t*)b->s_in1.d;
    if (player) {
        if ((rc = stage_in(b, b->s_in2, player, (size_t)n))) return rc;
        a.player = (const uint8_t*)b->s_in2.d;
    }
    if ((rc = stage_outputs(b, n, a))) return rc;
    if ((rc = launch_game<M_STEP_RT>(b, a))) return rc;
    return fetch_outputs(b, n, done, lines, dead);
}

int tetris_observe_records(tetris_batch* b, const int32_t* idx, int n, tetris_record* records, uint8_t* round_over,
                           int8_t* last_winner) {
    int rc = check_batch(b);
    if (rc) return rc;
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    const size_t rec_bytes = (size_t)n * b->P * sizeof(tetris_record);
    if ((rc = b->s_big.ensure(rec_bytes + 16))) return rc;
    if ((rc = b->s_out0.ensure((size_t)n + 4))) return rc;
    if ((rc = b->s_out1.ensure((size_t)n + 4))) return rc;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    tetris_record* d_rec = (tetris_record*)b->s_big.d;
    HIP_TRY(hipMemsetAsync(d_rec, 0, rec_bytes, b->stream));      // struct padding stays deterministic
#define LAUNCH_OBSERVE(PP, TT)                                                                                              \
    hipLaunchKernelGGL((k_observe<PP, TT>), grid, block, 0, b->stream, geo_of_batch(b), n, d_idx, b->H, d_rec, \
                       (uint8_t*)b->s_out0.d, (int8_t*)b->s_out1.d)
    if (b->P == 1 && !b->tint) LAUNCH_OBSERVE(1, false);
    else if (b->P == 1) LAUNCH_OBSERVE(1, true);
    else if (b->P == 2 && !b->tint) LAUNCH_OBSERVE(2, false);
    else if (b->P == 2) LAUNCH_OBSERVE(2, true);
    else if (b->P == 3 && !b->tint) LAUNCH_OBSERVE(3, false);
    else if (b->P == 3) LAUNCH_OBSERVE(3, true);
    else if (!b->tint) LAUNCH_OBSERVE(4, false);
    else LAUNCH_OBSERVE(4, true);
#undef LAUNCH_OBSERVE
    HIP_TRY(hipGetLastError());
    if (records) HIP_TRY(hipMemcpyAsync(b->s_big.h, b->s_big.d, rec_bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out0.h, b->s_out0.d, (size_t)n, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out1.h, b->s_out1.d, (size_t)n, hipMemcpyDeviceToHost, b->stream));
    if ((rc = finish_call(b))) return rc;
    if (records) memcpy(records, b->s_big.h, rec_bytes);
    if (round_over) memcpy(round_over, b->s_out0.h, (size_t)n);
    if (last_winner) memcpy(last_winner, b->s_out1.h, (size_t)n);
    return TETRIS_OK;
}

int tetris_observe_packed_dev(tetris_batch* b, const int32_t* d_idx, int n, const uint8_t* d_player, uint8_t* d_visual,
                              uint8_t* d_vector, uint8_t* d_piece) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!d_visual || !d_vector || !d_piece) return fail(TETRIS_E_ARG, "visual/vector/piece are NULL");
    if (b->P > 2) return fail(TETRIS_E_ARG, "the packed observation is defined for one or two players (own / opponent's board: state_unpack.py:88-137)");
    if (n < 0 || (!d_idx && n > b->N)) return fail(TETRIS_E_ARG, "n out of range");
    if (n == 0) return TETRIS_OK;
    b->home_async = true;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (b->H % 2 == 0 && ((uintptr_t)d_visual & 3u) == 0 && ((uintptr_t)d_vector & 3u) == 0) {
        // one wave per workgroup (<= 20 KB of LDS): several workgroups per CU overlap their load / build / store phases
        constexpr int OB = 64;
        const int nw = b->H * NCOL / 4;
        const size_t lds = (size_t)OB * nw * 4;
        dim3 ogrid((unsigned)((n + OB - 1) / OB), (unsigned)b->P), oblock(OB);
        if (b->P == 1)
            hipLaunchKernelGGL((k_observe_packed<1, OB>), ogrid, oblock, lds, b->stream, geo_of_batch(b), n, d_idx, d_player, b->H,
                               d_visual, d_vector, d_piece);
        else
            hipLaunchKernelGGL((k_observe_packed<2, OB>), ogrid, oblock, lds, b->stream, geo_of_batch(b), n, d_idx, d_player, b->H,
                               d_visual, d_vector, d_piece);
        HIP_TRY(hipGetLastError());
        return TETRIS_OK;
    }
    const size_t lds = (size_t)256 * b->H * NCOL;             // odd heights (boards are not word-aligned in `visual`): byte tile
    if (lds > 48 * 1024) {      // up to 79 KB of the CU's 160 KB for 31-row boards
        const void* fn = b->P == 1 ? (const void*)k_observe_packed_bytes<1> : (const void*)k_observe_packed_bytes<2>;
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (b->P == 1)
        hipLaunchKernelGGL(k_observe_packed_bytes<1>, grid, block, lds, b->stream, geo_of_batch(b), n, d_idx, d_player, b->H, d_visual,
                           d_vector, d_piece);
    else
        hipLaunchKernelGGL(k_observe_packed_bytes<2>, grid, block, lds, b->stream, geo_of_batch(b), n, d_idx, d_player, b->H, d_visual,
                           d_vector, d_piece);
    HIP_TRY(hipGetLastError());
    return TETRIS_OK;
}

int tetris_observe_packed(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* visual, uint8_t* vector,
                          uint8_t* piece) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!visual || !vector || !piece) return fail(TETRIS_E_ARG, "visual/vector/piece are NULL");
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    const uint8_t* d_player = nullptr;
    if (player) {
        for (int i = 0; i < n; i++)
            if (player[i] >= b->P) return fail(TETRIS_E_ARG, "player index out of range");
        if ((rc = stage_in(b, b->s_in0, player, (size_t)n))) return rc;
        d_player = (const uint8_t*)b->s_in0.d;
    }
    const size_t vis = (size_t)b->P * n * b->H * NCOL, vec = (size_t)b->P * n * 12, pc = (size_t)b->P * n;
    if ((rc = b->s_big.ensure(vis + 16)) || (rc = b->s_out0.ensure(vec + 4)) || (rc = b->s_out1.ensure(pc + 4))) return rc;
    if ((rc = tetris_observe_packed_dev(b, d_idx, n, d_player, (uint8_t*)b->s_big.d, (uint8_t*)b->s_out0.d, (uint8_t*)b->s_out1.d))) return rc;
    HIP_TRY(hipMemcpyAsync(b->s_big.h, b->s_big.d, vis, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out0.h, b->s_out0.d, vec, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out1.h, b->s_out1.d, pc, hipMemcpyDeviceToHost, b->stream));
    if ((rc = finish_call(b))) return rc;
    memcpy(visual, b->s_big.h, vis); memcpy(vector, b->s_out0.h, vec); memcpy(piece, b->s_out1.h, pc);
    return TETRIS_OK;
}

static int snapshot_impl(tetris_batch* b, const int32_t* idx, int n, uint32_t* blob, int restore) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!blob) return fail(TETRIS_E_ARG, "blob is NULL");
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    const int words = NGWORDS + b->P * b->nw;
    const size_t bytes = (size_t)n * words * 4;
    if ((rc = b->s_big.ensure(bytes + 16))) return rc;
    if (restore) {
        memcpy(b->s_big.h, blob, bytes);
        HIP_TRY(hipMemcpyAsync(b->s_big.d, b->s_big.h, bytes, hipMemcpyHostToDevice, b->stream));
    }
    size_t total = (size_t)n * words;
    hipLaunchKernelGGL(k_snapshot, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, b->stream, geo_of_batch(b), n, d_idx,
                       (uint32_t*)b->s_big.d, restore);
    HIP_TRY(hipGetLastError());
    if (!restore) HIP_TRY(hipMemcpyAsync(b->s_big.h, b->s_big.d, bytes, hipMemcpyDeviceToHost, b->stream));
    if ((rc = finish_call(b))) return rc;
    if (!restore) memcpy(blob, b->s_big.h, bytes);
    return TETRIS_OK;
}

int tetris_snapshot(tetris_batch* b, const int32_t* idx, int n, uint32_t* blob) { return snapshot_impl(b, idx, n, blob, 0); }
int tetris_restore(tetris_batch* b, const int32_t* idx, int n, const uint32_t* blob) {
    return snapshot_impl(b, idx, n, (uint32_t*)blob, 1);
}

int tetris_set_dead(tetris_batch* b, const int32_t* idx, int n, const uint8_t* dead) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!dead) return fail(TETRIS_E_ARG, "dead is NULL");
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    if ((rc = stage_in(b, b->s_in0, dead, (size_t)n * b->P))) return rc;
    hipLaunchKernelGGL(k_set_dead, dim3((unsigned)((n * b->P + 255) / 256)), dim3(256), 0, b->stream, geo_of_batch(b), n, d_idx,
                       (const uint8_t*)b->s_in0.d);
    HIP_TRY(hipGetLastError());
    return finish_call(b);
}

int tetris_enumerate_drops_dev_ex(tetris_batch* b, const int32_t* d_idx, int n, const uint8_t* d_player, uint8_t* d_valid,
                                  int8_t* d_land_y, uint8_t* d_cleared, uint32_t* d_after, int flags);

int tetris_enumerate_drops(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* valid, int8_t* land_y,
                           uint8_t* cleared, uint32_t* after) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!valid || !land_y || !cleared) return fail(TETRIS_E_ARG, "valid/land_y/cleared are NULL");
    const int32_t* d_idx;
    if ((rc = stage_idx(b, idx, n, &d_idx))) return rc;
    if (n == 0) return TETRIS_OK;
    const uint8_t* d_player = nullptr;
    if (player) {
        for (int i = 0; i < n; i++)
            if (player[i] >= b->P) return fail(TETRIS_E_ARG, "player index out of range");
        if ((rc = stage_in(b, b->s_in0, player, (size_t)n))) return rc;
        d_player = (const uint8_t*)b->s_in0.d;
    }
    const size_t lanes = (size_t)n * 40;
    if ((rc = b->s_out0.ensure(lanes + 4))) return rc;
    if ((rc = b->s_out1.ensure(lanes + 4))) return rc;
    if ((rc = b->s_out2.ensure(lanes + 4))) return rc;
    if (after && (rc = b->s_big.ensure(lanes * NCOL * 4 + 16))) return rc;
    uint32_t* d_after = after ? (uint32_t*)b->s_big.d : nullptr;
    if ((rc = tetris_enumerate_drops_dev_ex(b, d_idx, n, d_player, (uint8_t*)b->s_out0.d, (int8_t*)b->s_out1.d, (uint8_t*)b->s_out2.d,
                                            d_after, 0))) return rc;
    HIP_TRY(hipMemcpyAsync(b->s_out0.h, b->s_out0.d, lanes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out1.h, b->s_out1.d, lanes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->s_out2.h, b->s_out2.d, lanes, hipMemcpyDeviceToHost, b->stream));
    if (after) HIP_TRY(hipMemcpyAsync(b->s_big.h, b->s_big.d, lanes * NCOL * 4, hipMemcpyDeviceToHost, b->stream));
    if ((rc = finish_call(b))) return rc;
    memcpy(valid, b->s_out0.h, lanes); memcpy(land_y, b->s_out1.h, lanes); memcpy(cleared, b->s_out2.h, lanes);
    if (after) memcpy(after, b->s_big.h, lanes * NCOL * 4);
    return TETRIS_OK;
}

int tetris_enumerate_drops_dev_ex(tetris_batch* b, const int32_t* d_idx, int n, const uint8_t* d_player, uint8_t* d_valid,
                                  int8_t* d_land_y, uint8_t* d_cleared, uint32_t* d_after, int flags) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!d_valid || !d_land_y || !d_cleared) return fail(TETRIS_E_ARG, "valid/land_y/cleared are NULL");
    if (n < 0 || (!d_idx && n > b->N)) return fail(TETRIS_E_ARG, "n out of range");
    if (flags & ~TETRIS_ENUM_PLANAR) return fail(TETRIS_E_ARG, "unknown flag");
    if ((flags & TETRIS_ENUM_PLANAR) && ((((uintptr_t)d_valid | (uintptr_t)d_land_y | (uintptr_t)d_cleared) & 3u) || ((uintptr_t)d_after & 15u)))
        return fail(TETRIS_E_ARG, "planar outputs: valid / land_y / cleared must be 4-byte aligned, after 16-byte aligned");
    if (n == 0) return TETRIS_OK;
    b->home_async = true;
    dim3 grid((unsigned)((n + ENUM_BOARDS - 1) / ENUM_BOARDS)), block(ENUM_BLOCK);
    const Geo geo = geo_of_batch(b);
    const bool planar = (flags & TETRIS_ENUM_PLANAR) != 0;
#define LAUNCH_ENUM(PP, PL) hipLaunchKernelGGL((k_enumerate<PP, PL>), grid, block, 0, b->stream, geo, n, d_idx, d_player, b->H, d_valid, \
                                               d_land_y, d_cleared, d_after)
    if (b->P == 1) { if (planar) LAUNCH_ENUM(1, true); else LAUNCH_ENUM(1, false); }
    else if (b->P == 2) { if (planar) LAUNCH_ENUM(2, true); else LAUNCH_ENUM(2, false); }
    else if (b->P == 3) { if (planar) LAUNCH_ENUM(3, true); else LAUNCH_ENUM(3, false); }
    else { if (planar) LAUNCH_ENUM(4, true); else LAUNCH_ENUM(4, false); }
#undef LAUNCH_ENUM
    HIP_TRY(hipGetLastError());
    return TETRIS_OK;
}

int tetris_enumerate_drops_dev(tetris_batch* b, const int32_t* d_idx, int n, const uint8_t* d_player, uint8_t* d_valid,
                               int8_t* d_land_y, uint8_t* d_cleared, uint32_t* d_after) {
    return tetris_enumerate_drops_dev_ex(b, d_idx, n, d_player, d_valid, d_land_y, d_cleared, d_after, 0);
}

int tetris_timer_start(tetris_batch* b) {
    int rc = check_batch(b);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    return TETRIS_OK;
}

int tetris_timer_stop(tetris_batch* b, float* elapsed_ms) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!elapsed_ms) return fail(TETRIS_E_ARG, "elapsed_ms is NULL");
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
    return TETRIS_OK;
}

int tetris_get_actions(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* keys, uint8_t* lens,
                       int32_t* count, int max_lists, int max_keys) {
    int rc = check_batch(b);
    if (rc) return rc;
    if (!keys || !lens || !count || max_lists < 1 || max_keys < 1 || max_keys > 255) return fail(TETRIS_E_ARG, "keys/lens/count/max_*");
    if (n < 0 || (!idx && n > b->N)) return fail(TETRIS_E_ARG, "n out of range");
    if (player)
        for (int i = 0; i < n; i++)
            if (player[i] >= b->P) return fail(TETRIS_E_ARG, "player index out of range");
    const int LANE_LISTS = 16;                       // lists one (x, rotation) start can produce (<= H/2)
    const int CHUNK_GAMES = 1024;                    // bounds the staging buffers
    Stage &s_cnt = b->s_act0, &s_len = b->s_act1, &s_key = b->s_act2;
    int result = TETRIS_OK;
    for (int g0 = 0; g0 < n && result == TETRIS_OK; g0 += CHUNK_GAMES) {
        const int m = (n - g0 < CHUNK_GAMES) ? n - g0 : CHUNK_GAMES;
        const size_t lanes = (size_t)m * 40;
        const int32_t* d_idx = nullptr;
        std::vector<int32_t> ident;
        const int32_t* h_idx = idx ? idx + g0 : nullptr;
        if (!h_idx && g0 > 0) { ident.resize(m); for (int i = 0; i < m; i++) ident[i] = g0 + i; h_idx = ident.data(); }
        if ((rc = stage_idx(b, h_idx, m, &d_idx))) { result = rc; break; }
        const uint8_t* d_player = nullptr;
        if (player) {
            if ((rc = stage_in(b, b->s_in0, player + g0, (size_t)m))) { result = rc; break; }
            d_player = (const uint8_t*)b->s_in0.d;
        }
        if ((rc = s_cnt.ensure(lanes + 4)) || (rc = s_len.ensure(lanes * LANE_LISTS + 4)) ||
            (rc = s_key.ensure(lanes * LANE_LISTS * max_keys + 4))) { result = rc; break; }
        dim3 grid((unsigned)((lanes + 63) / 64)), block(64);
        if (b->P == 1)
            hipLaunchKernelGGL(k_actions<1>, grid, block, 0, b->stream, geo_of_batch(b), m, d_idx, d_player, b->H, (uint8_t*)s_cnt.d,
                               (uint8_t*)s_len.d, (uint8_t*)s_key.d, LANE_LISTS, max_keys, b->flags);
        else if (b->P == 2)
            hipLaunchKernelGGL(k_actions<2>, grid, block, 0, b->stream, geo_of_batch(b), m, d_idx, d_player, b->H, (uint8_t*)s_cnt.d,
                               (uint8_t*)s_len.d, (uint8_t*)s_key.d, LANE_LISTS, max_keys, b->flags);
        else if (b->P == 3)
            hipLaunchKernelGGL(k_actions<3>, grid, block, 0, b->stream, geo_of_batch(b), m, d_idx, d_player, b->H, (uint8_t*)s_cnt.d,
                               (uint8_t*)s_len.d, (uint8_t*)s_key.d, LANE_LISTS, max_keys, b->flags);
        else
            hipLaunchKernelGGL(k_actions<4>, grid, block, 0, b->stream, geo_of_batch(b), m, d_idx, d_player, b->H, (uint8_t*)s_cnt.d,
                               (uint8_t*)s_len.d, (uint8_t*)s_key.d, LANE_LISTS, max_keys, b->flags);
        if (hipGetLastError() != hipSuccess) { result = fail(TETRIS_E_HIP, "k_actions launch failed"); break; }
        (void)hipMemcpyAsync(s_cnt.h, s_cnt.d, lanes, hipMemcpyDeviceToHost, b->stream);
        (void)hipMemcpyAsync(s_len.h, s_len.d, lanes * LANE_LISTS, hipMemcpyDeviceToHost, b->stream);
        (void)hipMemcpyAsync(s_key.h, s_key.d, lanes * LANE_LISTS * max_keys, hipMemcpyDeviceToHost, b->stream);
        if ((rc = finish_call(b))) { result = rc; break; }
        const uint8_t* hc = (const uint8_t*)s_cnt.h;
        const uint8_t* hl = (const uint8_t*)s_len.h;
        const uint8_t* hk = (const uint8_t*)s_key.h;
        for (int i = 0; i < m && result == TETRIS_OK; i++) {
            int total = 0;
            for (int xi = 0; xi < 10; xi++)               // the reference enumerates x-major, rotation-minor
                for (int r = 0; r < 4; r++) {
                    const size_t lane = (size_t)i * 40 + r * 10 + xi;
                    for (int k = 0; k < hc[lane]; k++) {
                        if (total >= max_lists) { result = fail(TETRIS_E_ARG, "more than max_lists key lists for one game"); break; }
                        const int len = hl[lane * LANE_LISTS + k];
                        lens[(size_t)(g0 + i) * max_lists + total] = (uint8_t)len;
                        memcpy(keys + ((size_t)(g0 + i) * max_lists + total) * max_keys, hk + (lane * LANE_LISTS + k) * max_keys, (size_t)len);
                        total++;
                    }
                }
            count[g0 + i] = total;
        }
    }
    return result;
}

// Chained launches of two batches at once would need room for four launches; only one batch per device chains at a time
// (another one that comes along meanwhile puts its launches on one stream).
static std::mutex g_chain_mutex;
static tetris_batch* g_chain_owner[64] = {nullptr};
static bool chain_acquire(tetris_batch* b) {
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    tetris_batch*& owner = g_chain_owner[b->device & 63];
    if (owner && owner != b) return false;
    owner = b;
    return true;
}
static void chain_release(tetris_batch* b) {
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    tetris_batch*& owner = g_chain_owner[b->device & 63];
    if (owner == b) owner = nullptr;
}

static bool rollout_chained(tetris_batch* b, int steps_per_launch) {
    return b->use_chain && !b->tint && !b->split && b->stream == b->own_stream &&
           (b->P == 1 || (b->P == 2 && steps_per_launch == 1 && b->use_duo)) && chain_fits(b);
}

int tetris_rollout_is_chained(tetris_batch* b, int steps_per_launch) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    return rollout_chained(b, steps_per_launch) ? 1 : 0;
}

// Waves of a chained call gave up waiting (F_CHAIN): with every stream drained, each wave's epoch word names the last launch that
// finished its games (chain_wait).  The games of the waves that are behind are stepped on to the call's last launch by the
// un-chained kernel on the batch's stream, group by group (waves that stopped at the same launch), then the epoch words are set
// to the final epoch and chaining is switched off for this batch (a device that starved a launch once is shared with something:
// tetris_set_chained(b, 1) switches it back on).  Bit-exact: a wave that gives up has not touched its games.
static int chain_recover(tetris_batch* b) {
    volatile uint32_t* f = b->flags;
    const int lanes = b->P == 1 ? CHAIN_LANES : 32;
    const int waves = (b->N + lanes - 1) / lanes;
    hipStream_t const home = b->own_stream;
    struct StreamGuard { tetris_batch* b; hipStream_t keep; ~StreamGuard() { b->stream = keep; } } guard{b, b->stream};
    b->stream = home;
    const size_t bytes = (size_t)waves * CHAIN_STRIDE * sizeof(uint32_t);
    if (!b->h_chain && !(b->h_chain = (uint32_t*)malloc((((size_t)b->N + 15) / 16) * sizeof(uint32_t) * CHAIN_STRIDE))) return fail(TETRIS_E_HIP, "out of host memory");
    HIP_TRY(hipMemcpyAsync(b->h_chain, b->d_chain, bytes, hipMemcpyDeviceToHost, home));
    HIP_TRY(hipStreamSynchronize(home));
    const uint32_t last = b->chain_epoch, epoch0 = b->chain_call.epoch0;
    const int S = b->chain_call.steps_per_launch;
    std::vector<uint32_t> stops;                                  // distinct epochs at which waves stopped, ascending
    for (int w = 0; w < waves; w++) {
        const uint32_t c = b->h_chain[(size_t)w * CHAIN_STRIDE] & ~CHAIN_ABANDONED;
        if (c < epoch0 || c > last) return fail(TETRIS_E_HIP, "chained launches: an epoch word is outside the range of the call that gave up; state is invalid");
        if (c < last) stops.push_back(c);
    }
    std::sort(stops.begin(), stops.end());
    stops.erase(std::unique(stops.begin(), stops.end()), stops.end());
    const uint32_t saved_margin = b->margin;
    struct MarginGuard { tetris_batch* b; uint32_t saved; ~MarginGuard() { b->margin = saved; } } margin_guard{b, saved_margin};
    const int FUSE = 16;                                          // env-steps per recovery launch (the flag words are read after every launch)
    if (b->margin < (uint32_t)(2 * FUSE * 2 + 16)) b->margin = (uint32_t)(2 * FUSE * 2 + 16);
    std::vector<int32_t> idx;
    for (uint32_t c : stops) {
        idx.clear();
        for (int w = 0; w < waves; w++)
            if ((b->h_chain[(size_t)w * CHAIN_STRIDE] & ~CHAIN_ABANDONED) == c)
                for (int g = w * lanes; g < (w + 1) * lanes && g < b->N; g++) idx.push_back(g);
        unsigned long long step = b->chain_call.first_step + (unsigned long long)(c - epoch0) * (unsigned long long)S;
        unsigned long long todo = (unsigned long long)(last - c) * (unsigned long long)S;
        while (todo > 0) {
            const int steps = todo < (unsigned long long)FUSE ? (int)todo : FUSE;
            const int32_t* d_idx = nullptr;
            int rc = stage_idx(b, idx.data(), (int)idx.size(), &d_idx);
            if (rc) return rc;
            KArgs a = base_args(b, (int)idx.size(), d_idx);
            a.ms = b->chain_call.ms; a.steps = steps; a.policy_seed = b->chain_call.policy_seed; a.first_step = step;
            if ((rc = launch_game<M_ROLLOUT>(b, a))) return rc;
            HIP_TRY(hipStreamSynchronize(home));
            if ((rc = service_flags(b))) return rc;                // (an extension of the RNG tables is enqueued before the next launch)
            step += (unsigned long long)steps; todo -= (unsigned long long)steps;
        }
    }
    // every wave's word = the call's last epoch, without the bit (hipMemsetD32: the lines' other words are never read)
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b->d_chain, (int)last, (size_t)waves * CHAIN_STRIDE, home));
    HIP_TRY(hipStreamSynchronize(home));
    f[F_CHAIN] = 0;
    b->chain_fell_back = true;
    b->use_chain = 0;
    return TETRIS_OK;
}

// The launches of one chained call through the batch's own queues (tetris_aql.h).  Returns with every packet retired.
// `group`: run-ahead of the whole call in launches (as for gate_launch); b->chain_epoch has not been advanced yet.
static int rollout_direct(tetris_batch* b, aql::Device* dev, int launches, int steps_per_launch, uint32_t policy_seed, uint64_t first_step,
                          int ms, int group, float* elapsed_ms) {
    aql::Queues& qs = dev->qs;
    const int depth = b->chain_depth;
    // one-player batches on queues that deal their blocks round-robin over the XCDs: the XCD-affine kernel (k_chain_affine), whole groups
    // of eight workgroups, no cache maintenance between a queue's launches
    const bool affine = b->use_affine && qs.affine_ok && (b->P == 1 ? dev->chain1_affine.ok && CHAIN_LANES == 64 : dev->duo_affine.ok);
    b->last_affine = affine;
    const aql::Kernel& kern = affine ? (b->P == 1 ? dev->chain1_affine : dev->duo_affine) : (b->P == 1 ? dev->chain1 : dev->duo);
    uint32_t blocks = b->P == 1 ? (uint32_t)((b->N + CHAIN_LANES - 1) / CHAIN_LANES) : (uint32_t)((b->N + 31) / 32);
    if (affine) blocks = (blocks + 7u) & ~7u;
    const int wgroup = std::min(aql::SLOTS / 2 - 2, std::max(8, group / depth));
    static const bool timing = getenv("TETRIS_TIMING") != nullptr;
    // (experiment knob: the packets between a queue's first and last with fence scope "none" instead of "agent")
    static const int mid_scope = [] { const char* e = getenv("TETRIS_DIRECT_FENCE"); return e && !strcmp(e, "none") ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT; }();
    const auto t_begin = std::chrono::steady_clock::now();
    uint64_t h_begin = 0, h_seen = 0;
    if (timing) (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &h_begin);
    const uint32_t epoch0 = b->chain_epoch;
    b->chain_epoch = epoch0 + (uint32_t)launches;       // (a launch that never gets enqueued leaves waves waiting: they give up, chain_recover finishes the call)
    aql::Pending pd;
    bool gate_pending[CHAIN_STREAMS][2] = {}, done_armed[CHAIN_STREAMS] = {};
    int mine[CHAIN_STREAMS] = {}, gate_slot[CHAIN_STREAMS] = {};
    // whatever happens below, every packet that was written is rung in and retired before this function returns
    auto drain = [&]() {
        aql::ring(qs, pd);
        bool ok = true;
        for (int k = 0; k < depth; k++) {
            if (!pd.wrote[k]) continue;
            if (!done_armed[k]) {          // the call ended early: a barrier packet behind what queue k holds carries its completion signal
                hsa_signal_store_relaxed(qs.done[k], 1);
                aql::write_barrier(qs, pd, k, qs.done[k]);
                done_armed[k] = true;
                aql::ring(qs, pd);
            }
            ok = aql::wait_signal(qs.done[k]) && ok;
        }
        return ok;
    };
    struct DrainGuard { decltype(drain)& d; bool armed; ~DrainGuard() { if (armed) (void)d(); } } guard{drain, true};
    // (An empty barrier packet rung in ahead of the first launch, to wake the idle queues while the host prepares, bought nothing:
    // 5.12-5.38 us per launch with it, 5.15-5.22 without, driver's flags, alternating processes — profiles/r03/direct_dispatch.txt.)
    hsa_signal_t start_sig = qs.first, end_sig = qs.done[(launches - 1) % depth];
    int rc = TETRIS_OK, unflushed = 0;
    for (int l = 0; l < launches; l++) {
        const int k = l % depth;
        if ((rc = service_flags(b))) return rc;
        hsa_signal_t sig{};
        const bool first_on_queue = l < depth, last_on_queue = l >= launches - depth;
        if (last_on_queue) { hsa_signal_store_relaxed(qs.done[k], 1); sig = qs.done[k]; done_armed[k] = true; if (l == 0) start_sig = sig; }
        else if (l == 0) { hsa_signal_store_relaxed(qs.first, 1); sig = qs.first; }
        else if (mine[k] >= wgroup) {                    // at most 2 * wgroup + 1 packets of this queue outstanding
            const int sl = gate_slot[k];
            if (gate_pending[k][sl]) { aql::ring(qs, pd); if (!aql::wait_signal(qs.gate[k][sl])) return fail(TETRIS_E_HIP, "direct dispatch: a flow-control signal never came"); }
            hsa_signal_store_relaxed(qs.gate[k][sl], 1);
            sig = qs.gate[k][sl]; gate_pending[k][sl] = true; gate_slot[k] = sl ^ 1; mine[k] = 0;
        }
        mine[k]++;
        KArgs a = base_args(b, b->N, nullptr);
        a.ms = ms; a.steps = steps_per_launch; a.policy_seed = policy_seed;
        a.first_step = first_step + (uint64_t)l * (uint64_t)steps_per_launch;
        a.chain = b->d_chain; a.epoch = epoch0 + (uint32_t)l + 1u; a.chain_spin_limit = b->chain_spin_limit;
        // Fences.  The first packet of every queue ACQUIRES at system scope: whatever the host or an earlier kernel wrote (a restore
        // through the copy engines, a step on the batch's stream) must not be met as a stale line in some XCD's L2.  No packet needs
        // more than an agent-scope RELEASE: the chained kernels leave no dirty line behind — state and epoch words are written
        // through (sc1), the counters are atomics performed at the memory side, the flag words live in host memory — and the
        // host waits for the completion signals.  (TETRIS_DIRECT_EDGE=system / agent: both edges at that scope, experiment knob.)
        static const int edge_knob = [] { const char* e = getenv("TETRIS_DIRECT_EDGE"); return !e ? -1 : (!strcmp(e, "agent") ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_SYSTEM); }();
        // The affine kernel keeps its state in the XCDs' L2s: its queue's LAST packet releases at system scope (memory is current when the
        // call returns), and between first and last there is NO cache maintenance at all — except that a packet which re-uses the first
        // slot of the argument ring acquires at agent scope (the scalar caches may still hold what the slot held 4096 launches ago).
        const int acq_edge = edge_knob < 0 ? HSA_FENCE_SCOPE_SYSTEM : edge_knob, rel_edge = edge_knob < 0 ? (affine ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT) : edge_knob;
        const bool ring_wraps = qs.issued[k].load() % aql::SLOTS == 0;
        const int mid = affine ? HSA_FENCE_SCOPE_NONE : mid_scope;
        a.xcd_base = qs.xcd_base[k] + b->xcd_skew; a.xcd_slot = (uint32_t)k;
        aql::write_dispatch(qs, pd, k, kern, &a, sizeof a, blocks, first_on_queue ? acq_edge : (ring_wraps ? HSA_FENCE_SCOPE_AGENT : mid), last_on_queue ? rel_edge : mid, sig);
        // the first launches go out one by one (the GPU is idle), later ones eight at a time (one fence + read back per eight)
        if (++unflushed >= 8 || l < 2 * depth || l == launches - 1) { aql::ring(qs, pd); unflushed = 0; }
    }
    b->chain_pending = true;
    const auto t_enq = std::chrono::steady_clock::now();
    guard.armed = false;
    if (!drain()) return fail(TETRIS_E_HIP, "direct dispatch: a completion signal never came");
    b->chain_pending = false;
    const auto t_seen = std::chrono::steady_clock::now();
    const double seen_us = std::chrono::duration<double>(t_seen - t_begin).count() * 1e6;
    if (timing) (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &h_seen);
    float ms_events = 0.0f;
    hsa_amd_profiling_dispatch_time_t t0{}, t1{};
    {
        if (hsa_amd_profiling_get_dispatch_time(dev->gpu, start_sig, &t0) == HSA_STATUS_SUCCESS &&
            hsa_amd_profiling_get_dispatch_time(dev->gpu, end_sig, &t1) == HSA_STATUS_SUCCESS && t1.end >= t0.start)
            ms_events = (float)((double)(t1.end - t0.start) * 1e3 / (double)dev->ts_freq);
    }
    if (affine)                                 // where the queues dealt from in this call (block 0 of every launch says): what the next call expects
        for (int k = 0; k < depth; k++) {
            const uint32_t seen = ((volatile uint32_t*)b->flags)[F_XCC0 + k];
            if (seen & 0x100u) { qs.xcd_base[k] = seen & 7u; ((volatile uint32_t*)b->flags)[F_XCC0 + k] = 0; }
        }
    const bool was_misplaced = ((volatile uint32_t*)b->flags)[F_PLACE] != 0;
    if ((rc = finish_call(b, true))) return rc;
    if (affine && !was_misplaced) b->affine_failures = 0;
    if (elapsed_ms) *elapsed_ms = ms_events;
    if (timing) {
        const double enq = std::chrono::duration<double>(t_enq - t_begin).count();
        const double tick_us = 1e6 / (double)dev->ts_freq;
        fprintf(stderr, "[tetris timing] %d launches, direct dispatch: host enqueue %.2f us/launch, last signal seen %.1f us after the first enqueue, between first start and last end %.1f us; "
                "first enqueue -> first start %.1f us, last end -> seen %.1f us, after that %.1f us\n",
                launches, enq * 1e6 / launches, seen_us, ms_events * 1e3, (double)(int64_t)(t0.start - h_begin) * tick_us, (double)(int64_t)(h_seen - t1.end) * tick_us,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_seen).count() * 1e6);
    }
    return TETRIS_OK;
}

int tetris_rollout_launch(tetris_batch* b, int launches, int steps_per_launch, uint32_t policy_seed, uint64_t first_step,
                          int ms, float* elapsed_ms) {
    const auto t_entry = std::chrono::steady_clock::now();
    int rc = check_batch(b);
    if (rc) return rc;
    if (launches < 1 || steps_per_launch < 0) return fail(TETRIS_E_ARG, "launches must be >= 1, steps_per_launch >= 0");
    if (steps_per_launch > 256) return fail(TETRIS_E_ARG, "steps_per_launch must be <= 256");
    // A step may consume 2 piece draws per player, and the host learns of a board that came close to the end of the RNG
    // tables only through the flag words, up to 2 * GATE_GROUP + 1 launches late (gate_launch): the low-water margin
    // covers what those launches can consume.  Nothing in this loop waits for the GPU unless the host runs that far ahead.
    const uint32_t saved_margin = b->margin;
    struct MarginGuard {                          // every return path below puts the margin back
        tetris_batch* b; uint32_t saved;
        ~MarginGuard() { b->margin = saved; }
    } margin_guard{b, saved_margin};
    int group = steps_per_launch ? GATE_GROUP / steps_per_launch : GATE_GROUP;       // fused launches: fewer of them in flight
    group = group < 1 ? 1 : group;
    const uint32_t need = (uint32_t)(2 * steps_per_launch * (2 * group + 2) + 16);
    if (b->margin < need) b->margin = need;
    // batches on their own stream: chained launches (k_chain / k_duo<.., true>) — consecutive launches rotate over CHAIN_STREAMS
    // streams and each wave waits for its own predecessor only, not for the slowest wave of the whole previous launch.
    // Only when CHAIN_STREAMS launches fit on the device together (chain_fits): a waiting wave keeps its slot, so a launch whose waves
    // wait must never be able to keep its predecessor's waves from being dispatched.
    const bool chained = rollout_chained(b, steps_per_launch) && chain_acquire(b);
    hipStream_t const home = b->stream;
    // Every way out of a chained call — errors included — first waits until the chain streams are idle and only then gives up the
    // device's chaining slot: another batch that took it while launches of this one were still in flight would put more than
    // chain_depth launches on the device (chain_fits counts on that bound).
    struct ChainGuard {
        tetris_batch* b; bool held; hipStream_t home;
        ~ChainGuard() {
            b->stream = home;                     // (base_args / launch_game / gate_launch work on b->stream)
            if (!held) return;
            if (b->chain_pending)
                for (int k = 0; k < CHAIN_STREAMS; k++) (void)hipStreamSynchronize(b->chain_stream[k]);
            chain_release(b);
        }
    } chain_guard{b, chained, home};
    if (!chained && b->use_graph && steps_per_launch >= 1) {
        // TETRIS_GRAPH=1 (profiling aid): the launches are captured into HIP graphs of up to 128 kernel nodes and replayed, so the
        // host makes one call per 128 launches.  Under rocprofv3 a plain launch costs the host ~8 us — more than the kernel
        // takes — and the kernels of a profiled run are then no longer back to back; replayed from a graph they are.
        HIP_TRY(hipEventRecord(b->ev0, home));
        for (int l0 = 0; l0 < launches; l0 += 128) {
            const int m = launches - l0 < 128 ? launches - l0 : 128;
            if ((rc = gate_launch(b, 1))) return rc;                      // flags are looked at between graphs: the margin below covers 128 launches
            const uint32_t keep = b->margin;
            if (b->margin < (uint32_t)(2 * steps_per_launch * 3 * 128 + 16)) b->margin = (uint32_t)(2 * steps_per_launch * 3 * 128 + 16);
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            HIP_TRY(hipStreamBeginCapture(home, hipStreamCaptureModeThreadLocal));
            for (int l = l0; l < l0 + m && !rc; l++) {
                KArgs a = base_args(b, b->N, nullptr);
                a.ms = ms; a.steps = steps_per_launch; a.policy_seed = policy_seed;
                a.first_step = first_step + (uint64_t)l * (uint64_t)steps_per_launch;
                rc = launch_game<M_ROLLOUT>(b, a);
            }
            b->margin = keep;
            const hipError_t ce = hipStreamEndCapture(home, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            HIP_TRY(ce);
            HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            const hipError_t le = hipGraphLaunch(exec, home);
            const hipError_t se = hipStreamSynchronize(home);          // (the graph objects must outlive the launch)
            (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
            HIP_TRY(le); HIP_TRY(se);
        }
        HIP_TRY(hipEventRecord(b->ev1, home));
        if ((rc = finish_call(b))) return rc;
        if (elapsed_ms) HIP_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
        return TETRIS_OK;
    }
    if (chained) {
        if (b->chain_epoch + (uint32_t)launches >= CHAIN_EPOCH_MAX || b->chain_epoch + (uint32_t)launches < b->chain_epoch) {
            // epoch numbers stay below the CHAIN_ABANDONED bit: restart the numbering (every chain stream is idle between calls)
            HIP_TRY(hipMemsetAsync(b->d_chain, 0, (((size_t)b->N + 15) / 16) * sizeof(uint32_t) * CHAIN_STRIDE, home));
            b->chain_epoch = 0;
            b->home_async = true;
        }
        b->chain_call.epoch0 = b->chain_epoch; b->chain_call.first_step = first_step; b->chain_call.steps_per_launch = steps_per_launch;
        b->chain_call.policy_seed = policy_seed; b->chain_call.ms = ms;
    }
    if (chained && b->home_async) {
        // the chain streams start behind the asynchronous work the batch's stream still holds (after a synchronous call it
        // is empty and nothing has to be ordered)
        HIP_TRY(hipEventRecord(b->chain_ev[CHAIN_STREAMS], home));
        for (int k = 0; k < CHAIN_STREAMS; k++) HIP_TRY(hipStreamWaitEvent(b->chain_stream[k], b->chain_ev[CHAIN_STREAMS], 0));
    }
    // Measurement aid (TETRIS_PREQUEUE=1, read per call; bench.py's `launch_us_gpu_paced`): the chain streams are parked behind a
    // blocker kernel that runs until the host, having enqueued every launch of the call, sets a flag word in pinned memory (or 200 ms
    // have passed), so every launch is queued before the first one starts and the host's launch cost — 2.5-5 us per launch, ~8 us
    // under rocprofv3 — does not enter the period.  No run-ahead gate in such a call (it would wait for the blocker): <= 600 launches.
    const bool prequeue = chained && getenv("TETRIS_PREQUEUE") != nullptr && launches <= 600;
    if (prequeue) {
        ((volatile uint32_t*)b->flags)[F_GO] = 0;
        hipLaunchKernelGGL(k_blocker, dim3(1), dim3(64), 0, home, (const uint32_t*)(b->flags + F_GO), 20000000ull);      // <= 200 ms
        HIP_TRY(hipEventRecord(b->chain_ev[CHAIN_STREAMS], home));
        for (int k = 0; k < CHAIN_STREAMS; k++) HIP_TRY(hipStreamWaitEvent(b->chain_stream[k], b->chain_ev[CHAIN_STREAMS], 0));
        group = 1 << 20;
    }
    b->last_direct = false; b->last_affine = false;
    if (chained && !prequeue && b->direct_min > 0) {
        // the device's own queues (tetris_aql.h); anything that keeps them from being set up switches them off for this batch.  They
        // are made by the first chained call of ANY length (loading the code object takes ~15 ms: a warm-up call's business,
        // not that of the first long call).
        aql::Device* dev = aql::device_for(b->device);
        std::string why = dev->why;
        if (!(dev->ok && aql::make_queues(dev, why))) {
            b->direct_min = 0;
            if (getenv("TETRIS_TIMING")) fprintf(stderr, "[tetris] direct dispatch is off: %s\n", why.c_str());
        } else if (aql::calibrate_affine(dev), launches >= b->direct_min) {       // (measured once, by whichever chained call comes first)
            if (b->home_async) { HIP_TRY(hipStreamSynchronize(home)); }      // what the batch's stream still holds comes first
            b->last_direct = true; b->direct_used = true;
            if (getenv("TETRIS_TIMING")) fprintf(stderr, "[tetris timing] call entry -> direct dispatch %.1f us\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count() * 1e6);
            return rollout_direct(b, dev, launches, steps_per_launch, policy_seed, first_step, ms, group, elapsed_ms);
        }
    }
    struct GoGuard {                              // (no return path leaves the blocker waiting for its flag)
        tetris_batch* b; bool armed;
        ~GoGuard() { if (armed) ((volatile uint32_t*)b->flags)[F_GO] = 1; }
    } go_guard{b, prequeue};
    // Chained calls attach their events to the kernels themselves (hipExtLaunchKernel: the first launch carries the start event,
    // the last launch of every stream its end event) instead of recording them as packets of their own in front of and behind the
    // launches: a marker packet ahead of the first kernel and one behind the last cost the driver's 20-launch region a few
    // microseconds each.  TETRIS_EXT_EVENTS=0: plain event records.
    static const bool ext_events = [] { const char* e = getenv("TETRIS_EXT_EVENTS"); return !(e && e[0] == '0'); }();
    const bool attach = chained && ext_events;
    if (!attach) HIP_TRY(hipEventRecord(b->ev0, chained ? b->chain_stream[0] : home));
    static const bool timing = getenv("TETRIS_TIMING") != nullptr;         // debug aid: host-side cost of this loop on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    double gate_s = 0.0;
    // Long chained calls are enqueued by one host thread PER CHAIN STREAM.  A launch costs the host 2.7-4.0 us depending on the
    // process (the core its thread sits on, what the box's other tenants do) and the GPU needs one every 4.03 us: in the slower
    // processes one thread could not keep up and the rollout ran at the host's pace, 4.2-4.6 us per launch
    // (profiles/r03/host_pace.txt).  Every stream's launches are independent of the others' on the host side — epoch and step of
    // launch l follow from l — so each thread enqueues every depth-th launch on its own stream with its own run-ahead gate; this
    // thread keeps looking at the flag words (RNG-table extensions) meanwhile.  TETRIS_ENQUEUE_THREADS_MIN=0: never; n: from n launches.
    static const int thread_min = [] { const char* e = getenv("TETRIS_ENQUEUE_THREADS_MIN"); return e ? atoi(e) : 256; }();
    const bool threaded = chained && !prequeue && thread_min > 0 && launches >= thread_min;
    // (this thread enqueues the first launches of every stream itself, so the GPU has work while the other threads start up,
    // and then serves stream 0)
    const int prefix = threaded ? std::min(launches, 8 * b->chain_depth) : launches;
    const uint32_t call_epoch0 = b->chain_epoch;
    for (int l = 0; l < prefix; l++) {
        if (chained) { b->stream = b->chain_stream[l % b->chain_depth]; b->chain_pending = true; }
        const auto t_gate = std::chrono::steady_clock::now();
        if ((rc = gate_launch(b, group))) return rc;
        if (timing) gate_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gate).count();
        KArgs a = base_args(b, b->N, nullptr);
        a.ms = ms; a.steps = steps_per_launch; a.policy_seed = policy_seed;
        a.first_step = first_step + (uint64_t)l * (uint64_t)steps_per_launch;
        if (chained) {
            a.chain = b->d_chain; a.epoch = ++b->chain_epoch; a.chain_spin_limit = b->chain_spin_limit;
            // (attached events: start of the call's first kernel; end of the last kernel of each stream — the call's last launch
            // carries the timing event ev1, the last launches of the other streams their join events)
            hipEvent_t start_ev = (attach && l == 0) ? b->ev0 : nullptr, stop_ev = nullptr;
            if (attach && l >= launches - b->chain_depth) stop_ev = l == launches - 1 ? b->ev1 : b->chain_ev[l % b->chain_depth];
            if (b->P == 1) hipExtLaunchKernelGGL((k_chain<1>), dim3((unsigned)((b->N + CHAIN_LANES - 1) / CHAIN_LANES)), dim3(64), 0, b->stream, start_ev, stop_ev, 0, a);
            else hipExtLaunchKernelGGL((k_duo<M_ROLLOUT, true>), dim3((unsigned)((b->N + 31) / 32)), dim3(64), 0, b->stream, start_ev, stop_ev, 0, a);
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                // this launch does not exist: without it no wave could ever publish its epoch, so the numbering steps back
                // (the launches already enqueued run to their end; ChainGuard waits for them)
                b->chain_epoch--;
                return fail(TETRIS_E_HIP, std::string("chained launch: ") + hipGetErrorString(le));
            }
        } else if ((rc = launch_game<M_ROLLOUT>(b, a)))
            return rc;
    }
    if (threaded && prefix < launches) {
        const int depth = b->chain_depth, wgroup = std::max(8, group / depth);
        const uint32_t epoch0 = call_epoch0;
        b->chain_epoch = epoch0 + (uint32_t)launches;           // (a launch that fails to be enqueued leaves waves waiting: they give up, chain_recover finishes the call)
        b->chain_pending = true;
        b->stream = home;
        std::atomic<int> failed{0}, finished{0};
        hipError_t worker_err[CHAIN_STREAMS] = {};
        int flag_rc = TETRIS_OK;
        const bool main_works_flag = [] { const char* e = getenv("TETRIS_ENQUEUE_MAIN_WORKS"); return !(e && e[0] == '0'); }();
        auto worker = [&](int k) {
            hipError_t err = hipSetDevice(b->device);
            int mine = 0, slot = 0;
            bool pending[2] = {false, false};
            hipStream_t const st = b->chain_stream[k];
            for (int l = prefix + k; l < launches && err == hipSuccess && !failed.load(std::memory_order_relaxed); l += depth) {
                if (k == 0 && main_works_flag && !flag_rc && (flag_rc = service_flags(b))) { failed.store(1); break; }      // (this thread: the flag words, as ever)
                if (mine >= wgroup) {                            // at most 2 * wgroup + 1 launches of this stream in flight
                    if (pending[slot]) {
                        hipError_t qe = hipErrorNotReady;
                        for (int spin = 0; spin < 2000000 && qe == hipErrorNotReady; spin++) qe = hipEventQuery(b->worker_gate_ev[k][slot]);
                        if (qe == hipErrorNotReady) qe = hipEventSynchronize(b->worker_gate_ev[k][slot]);
                        if (qe != hipSuccess) { err = qe; break; }
                    }
                    if ((err = hipEventRecord(b->worker_gate_ev[k][slot], st)) != hipSuccess) break;
                    pending[slot] = true; slot ^= 1; mine = 0;
                }
                mine++;
                KArgs a = base_args(b, b->N, nullptr);           // (table pointer and size under the tables' mutex: this thread may be extending them)
                a.ms = ms; a.steps = steps_per_launch; a.policy_seed = policy_seed;
                a.first_step = first_step + (uint64_t)l * (uint64_t)steps_per_launch;
                a.chain = b->d_chain; a.epoch = epoch0 + (uint32_t)l + 1u; a.chain_spin_limit = b->chain_spin_limit;
                hipEvent_t start_ev = (attach && l == 0) ? b->ev0 : nullptr, stop_ev = nullptr;
                if (attach && l >= launches - depth) stop_ev = l == launches - 1 ? b->ev1 : b->chain_ev[k];
                if (b->P == 1) hipExtLaunchKernelGGL((k_chain<1>), dim3((unsigned)((b->N + CHAIN_LANES - 1) / CHAIN_LANES)), dim3(64), 0, st, start_ev, stop_ev, 0, a);
                else hipExtLaunchKernelGGL((k_duo<M_ROLLOUT, true>), dim3((unsigned)((b->N + 31) / 32)), dim3(64), 0, st, start_ev, stop_ev, 0, a);
                err = hipGetLastError();
            }
            if (err != hipSuccess) { worker_err[k] = err; failed.store(1); }
            finished.fetch_add(1);
        };
        std::thread th[CHAIN_STREAMS];
        bool inline_k[CHAIN_STREAMS] = {};
        // (TETRIS_ENQUEUE_MAIN_WORKS=0, experiment knob: a thread for stream 0 too, this one only watches the flag words — same
        // periods, a few more outliers: profiles/r03/host_pace.txt)
        static const bool main_works = [] { const char* e = getenv("TETRIS_ENQUEUE_MAIN_WORKS"); return !(e && e[0] == '0'); }();
        for (int k = main_works ? 1 : 0; k < depth; k++) {
            try { th[k] = std::thread(worker, k); } catch (...) { inline_k[k] = true; }      // (no thread to be had: this one does that stream too)
        }
        if (main_works) worker(0);
        for (int k = 0; k < depth; k++) if (inline_k[k]) worker(k);
        while (finished.load() < depth) {
            if (!flag_rc && (flag_rc = service_flags(b))) failed.store(1);
            for (int spin = 0; spin < 200 && finished.load(std::memory_order_relaxed) < depth; spin++) __builtin_ia32_pause();
        }
        for (int k = 0; k < depth; k++) if (th[k].joinable()) th[k].join();
        if (flag_rc) return flag_rc;
        for (int k = 0; k < depth; k++)
            if (worker_err[k] != hipSuccess) return fail(TETRIS_E_HIP, std::string("chained launch (enqueue thread): ") + hipGetErrorString(worker_err[k]));
    }
    // Every chain stream gets an event behind its last launch (the last launch's stream carries the timing event ev1); the host
    // waits for these events by polling them — an interrupt wakes a blocked thread 10-20 us late, and querying a STREAM makes the
    // runtime push a marker through its queue and wait for it, ~6 us per stream even when it is idle.  Nothing on the GPU waits
    // for another stream here (a cross-stream join costs the rollout ~15 us at its end); the batch's own stream is ordered behind
    // the events for whatever comes next.
    hipStream_t const last = chained ? b->chain_stream[(launches - 1) % b->chain_depth] : b->stream;
    const int used = chained ? (launches < b->chain_depth ? launches : b->chain_depth) : 0;
    if (!attach) {
        HIP_TRY(hipEventRecord(b->ev1, last));
        for (int k = 0; k < used; k++)
            if (b->chain_stream[k] != last) HIP_TRY(hipEventRecord(b->chain_ev[k], b->chain_stream[k]));
    }
    for (int k = 0; k < used; k++) HIP_TRY(hipStreamWaitEvent(home, b->chain_stream[k] != last ? b->chain_ev[k] : b->ev1, 0));
    b->stream = home;
    if (prequeue) { ((volatile uint32_t*)b->flags)[F_GO] = 1; go_guard.armed = false; }      // everything is queued: go
    const auto t_enq = std::chrono::steady_clock::now();
    {
        auto poll = [](hipEvent_t e) {
            hipError_t qe = hipErrorNotReady;
            for (int spin = 0; spin < 200000 && qe == hipErrorNotReady; spin++) qe = hipEventQuery(e);
            return qe == hipErrorNotReady ? hipEventSynchronize(e) : qe;
        };
        HIP_TRY(poll(b->ev1));
        for (int k = 0; k < used; k++)
            if (b->chain_stream[k] != last) HIP_TRY(poll(b->chain_ev[k]));
    }
    const double ev1_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e6;
    if ((rc = finish_call(b, true))) return rc;   // the end event is behind everything this call enqueued: no stream drains
    if (elapsed_ms) HIP_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
    if (timing) {
        const double enq = std::chrono::duration<double>(t_enq - t_begin).count(), all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        fprintf(stderr, "[tetris timing] %d launches: host enqueue %.2f us/launch (of which gate + flags %.2f), until drained %.2f us/launch; "
                "call entry -> first enqueue %.1f us, end event seen %.1f us after the first enqueue, whole call %.1f us, between the events %.1f us\n",
                launches, enq * 1e6 / launches, gate_s * 1e6 / launches, all * 1e6 / launches,
                std::chrono::duration<double>(t_begin - t_entry).count() * 1e6, ev1_us, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count() * 1e6,
                elapsed_ms ? *elapsed_ms * 1e3 : 0.0);
    }
    return TETRIS_OK;
}

int tetris_rollout_random(tetris_batch* b, int launches, int steps_per_launch, uint32_t policy_seed, uint64_t first_step,
                          int ms, uint64_t counters[4], float* elapsed_ms) {
    uint64_t before[4] = {0, 0, 0, 0}, after[4] = {0, 0, 0, 0};
    int rc;
    if (counters && (rc = tetris_rollout_totals(b, before))) return rc;
    if ((rc = tetris_rollout_launch(b, launches, steps_per_launch, policy_seed, first_step, ms, elapsed_ms))) return rc;
    if (counters) {
        if ((rc = tetris_rollout_totals(b, after))) return rc;
        // per-game words are uint32 and wrap; a single call stays far below 2^32 per game
        for (int k = 0; k < 4; k++) counters[k] += after[k] - before[k];
    }
    return TETRIS_OK;
}

}  // extern "C"
