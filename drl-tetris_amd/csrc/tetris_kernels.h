// Kernel bodies of libtetris_hip.so, one call = the work of one lane.
//
// `__host__ __device__` like tetris_engine.h: tetris_hip.hip wraps each body in a gfx950 kernel
// (the product); tests/cpu_harness wraps the same bodies in plain loops so that the logic can be
// checked against the oracle on a machine without a GPU.  Nothing in the product calls these on
// the host.
#pragma once
#include "../../include/tetris_hip.h"
#include "tetris_engine.h"

namespace te {

struct KArgs {
    uint32_t* state;
    uint32_t* gstate;
    uint32_t* status;         // the batch's flag words (te::Flag; pinned host memory in the product)
    const uint8_t* table;     // [(chunk * 65536 + seed16) * 624 + r]
    const uint64_t* start;    // [65536] per-seed start entries
    const double* combo_pow;
    uint32_t n_draws, margin;
    int H;
    int n_games;              // N
    int n_stride;             // games per row of the state arrays (>= N)
    int n_players;            // P
    int nw;                   // words per player-board (NWORDS, or NWORDS_TINT with colour planes)
    int n;                    // lanes with work
    const int32_t* idx;       // NULL: identity
    const uint8_t* keys;      // [max_keys][P][n]
    const uint8_t* lens;      // [P][n]
    int max_keys;
    const uint8_t* rot;       // [n]
    const uint8_t* trans;     // [n]
    const uint8_t* player;    // [n] or NULL
    const int16_t* seeds;     // [n] or NULL
    const uint8_t* mask;      // [n] or NULL: lanes with mask[i] == 0 do nothing (device-side reset of finished games)
    int ms;
    uint8_t* done;            // [n]
    uint8_t* lines;           // [P][n]
    uint8_t* dead;            // [P][n]
    int steps;
    uint32_t policy_seed;
    unsigned long long first_step;
    unsigned long long* counters;
    uint32_t game_offset;     // global id of slot 0 (rollout policy / seed schedule)
    uint32_t* chain;          // chained launches: one epoch word per wave (NULL = launches are ordered by the stream)
    uint32_t epoch;           // chained launches: this launch's number; its waves wait for epoch - 1 and publish epoch
    uint32_t chain_spin_limit; // chained launches: polls of the epoch word before a wave gives up (tetris_set_chain_spin_limit)
    uint32_t xcd_base;        // XCD-affine chained launches (k_chain_affine): the XCD block 0 of this launch's queue is expected on
    uint32_t xcd_slot;        // ... and the number of that queue (block 0 reports where it really is: F_XCC0 + xcd_slot)
    uint32_t* shadow;         // split mode, side 1: undo record of the speculative loop-1 pass, UNDO_WORDS + nw rows of n_stride words
    int split_side;           // split mode: the player index this batch holds
    const uint32_t* xw[4];    // split mode: exchange words [n] each: my A, the opponent's A, player 0's B, player 1's B (separate buffers:
                              // the words a kernel wrote and the rows an all-gather delivered are read where they lie, no copies)
    uint32_t* xout;           // split mode: this stage's word per board [n]
    // step + observation in one launch (k_step_observe): the packed observation of the stepped state, from the perspective of
    // next_player[i] (NULL: player 0), in the layout of tetris_observe_packed_dev
    const uint8_t* next_player;
    uint8_t* obs_visual;      // [P][n][H][10]
    uint8_t* obs_vector;      // [P][n][12]
    uint8_t* obs_piece;       // [P][n]
};

enum Mode { M_INIT, M_RESET, M_MAKE, M_FINISH, M_STEP_KEYS, M_STEP_RT, M_ROLLOUT, M_SPLIT_INIT, M_SPLIT_RESET,
            M_STEP_RT_AUTO,      // M_STEP_RT + device-side auto-reset of finished games (seed schedule of the rollout)
            M_RESET_SCHED };     // reset with the next seed of the schedule (no seed array)

struct LaneCounters { unsigned long long steps, episodes, lines, sent; };   // used by the CPU test harness only

// Device-resident index arrays (the *_dev entry points) cannot be validated on the host: an out-of-range entry is clamped
// into the batch instead of becoming an out-of-bounds access (host arrays are rejected with TETRIS_E_ARG before launch).
TE_HD size_t safe_slot(const int32_t* idx, int i, int n_games) {
    if (!idx) return (size_t)i;
    const uint32_t v = (uint32_t)idx[i];
    return (size_t)(v < (uint32_t)n_games ? v : (uint32_t)n_games - 1u);
}
TE_HD int safe_player_value(int v, int P) { return v < P ? v : P - 1; }
TE_HD int safe_player(const uint8_t* player, int i, int P) { return safe_player_value(player ? (int)player[i] : 0, P); }

// a lane has work: inside the launch and, for masked launches, selected by the device-side mask
TE_HD bool lane_active(const KArgs& a, int i) { return i < a.n && (!a.mask || a.mask[i] != 0); }

TE_HD Geo geo_of(const KArgs& a, uint32_t* state = nullptr) {
    Geo g = {state ? state : a.state, a.gstate, (size_t)a.n_games, a.n_players, a.nw, (size_t)a.n_stride};
    return g;
}

// A lane's collected status bits -> the batch's flag words (plain stores; rare).  F_EXTEND carries the table size the
// kernel worked with, so the host can tell a fresh request from one that an extension has already answered.
TE_HD void report_status(const KArgs& a, uint32_t st) {
    if (!st) return;
    volatile uint32_t* f = a.status;
    if (st & ST_NEED_EXTEND) f[F_EXTEND] = a.n_draws;
    if (st & ST_STREAM_EXHAUSTED) f[F_EXHAUSTED] = 1u;
    if (st & ST_FIFO_OVERFLOW) f[F_FIFO] = 1u;
    if (st & ST_BAD_ARGUMENT) f[F_BADARG] = 1u;
}

// ---- chained launches (tetris_hip.hip: k_chain): the hand-over of a wave's 64 games from launch E - 1 to launch E
// A wave's epoch word holds the number of the last launch that finished this wave's games (bits 0..30).  CHAIN_ABANDONED (bit 31)
// is OR-ed into it by a wave that gave up waiting: the waves of every later launch that see the bit pass it on — they leave their
// games untouched too — and the host, once everything has drained, finishes the games of such waves un-chained from the epoch the
// low bits name (tetris_hip.hip: chain_recover).  The bit is set with an atomic OR, never by storing a value the wave has read
// earlier: the predecessor may publish between the last poll and the give-up, and its epoch must survive.  (Should the predecessor
// publish AFTER the OR, the bit is overwritten; the next launch's wave then waits for an epoch that never comes, gives up on its
// own bound and sets the bit again; the word still names the right epoch at every moment.)
constexpr uint32_t CHAIN_ABANDONED = 0x80000000u;
constexpr uint32_t CHAIN_EPOCH_MAX = 0x7FFF0000u;    // the host restarts the epoch numbering (at a drained point) before it gets here
// The epoch words of consecutive waves lie CHAIN_STRIDE words apart: one 128-byte line per wave.  Packed (32 waves' words in
// one line) every publication hit a line that 31 other waves were polling: GPU-paced 4.72 us per launch against 4.50 with one
// line each at the same poll interval, and with the lines private a SHORT poll interval pays (s_sleep 32: 4.50, 8: 4.24,
// 0-2: 4.13-4.15 us; packed it had been the other way round: profiles/r02/chain_sleep_ab.txt, chain_stride_ab.txt).
#ifndef TE_CHAIN_STRIDE
#define TE_CHAIN_STRIDE 32
#endif
constexpr int CHAIN_STRIDE = TE_CHAIN_STRIDE;
#ifndef TE_CHAIN_SLEEP
#define TE_CHAIN_SLEEP 1            // s_sleep between two polls of the epoch word (x 64 cycles)
#endif
constexpr uint32_t CHAIN_SPIN_LIMIT = 1u << 22;     // default polls before a wave gives up (each ~0.5 us: an agent-scope load + a short sleep): ~2 s

#if defined(__HIP_DEVICE_COMPILE__)
TE_HD void or_agent(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
TE_HD void or_agent(uint32_t* p, uint32_t v) { *p |= v; }
#endif

// true when the state of this wave's games as launch E - 1 left it is visible (their stores were `sc1` and drained before the
// epoch word was written, and the word is polled with an `sc1` load: MI355X_MICROARCH.md, valid forms of an inter-workgroup hand-off).
// false: the wave must leave its games untouched and publish nothing (`marker`: the one lane that records a give-up).
TE_HD bool chain_wait(const KArgs& a, uint32_t wave, bool marker) {
    const uint32_t want = a.epoch - 1u;
    uint32_t* word = a.chain + (size_t)wave * CHAIN_STRIDE;
    for (uint32_t spin = 0; spin < a.chain_spin_limit; spin++) {
        const uint32_t v = ld_agent(word);
        if (v == want) return true;
        if (v & CHAIN_ABANDONED) return false;              // an earlier launch's wave gave up: nothing to add
        // an XCD-affine launch found workgroups misplaced (they touched nothing; their games' epochs will never come): every waiting
        // wave learns it from the flag word within ~0.1 ms instead of waiting out its bound
        if ((spin & 255u) == 255u && ((volatile uint32_t*)a.status)[F_PLACE]) break;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_sleep(TE_CHAIN_SLEEP);
#endif
    }
    if (marker) { or_agent(word, CHAIN_ABANDONED); ((volatile uint32_t*)a.status)[F_CHAIN] = 1u; }
    return false;
}

// Residency census of a chained kernel (a launch with a.steps < 0, the kernel's ordinary resource footprint): every wave counts
// itself in and then waits until a.epoch waves have done so.  They can only all see the full count if that many waves of THIS
// kernel are resident on the device at the same time — which is what chaining `depth` launches of it presumes (tetris_hip.hip:
// chain_fits; MI355X_MICROARCH.md: the occupancy API can be off, verify with a census).  a.chain: [0] count, [CHAIN_STRIDE] != 0 = a
// wave gave up.
TE_HD void chain_census(const KArgs& a, bool marker) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (marker) (void)__hip_atomic_fetch_add(a.chain, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t spin = 0; spin < a.chain_spin_limit; spin++) {
        if (ld_agent(a.chain) >= a.epoch) return;
        __builtin_amdgcn_s_sleep(8);
    }
    if (marker) st_agent(a.chain + CHAIN_STRIDE, 1u);
#else
    (void)a; (void)marker;
#endif
}

TE_HD Ctx make_ctx(const KArgs& a, const uint32_t* shapes, bool tint = false, bool queue = true) {
    Ctx cx;
    cx.shapes = shapes;
    cx.table = a.table;
    cx.start = a.start;
    cx.combo_pow = a.combo_pow;
    cx.n_draws = a.n_draws;
    cx.margin = a.margin;
    cx.H = a.H;
    cx.floor_bits = ~0u << a.H;
    cx.tint = tint;
    cx.queue = queue;
    return cx;
}

template <int P>
TE_HD void write_outputs(const KArgs& a, int i, const Game<P>& g, int done) {
    if (a.done) a.done[i] = (uint8_t)done;
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        if (a.lines) a.lines[(size_t)p * a.n + i] = (uint8_t)g.pl[p].reward;
        if (a.dead) a.dead[(size_t)p * a.n + i] = (uint8_t)g.pl[p].dead;
    }
}

// PythonHandle.cpp:138-147 make_actions
template <int P>
TE_HD void make_keys(const Ctx& cx, const KArgs& a, int i, Game<P>& g) {
    if (g.round_over) return;
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        Player& q = g.pl[p];
        int len = a.lens[(size_t)p * a.n + i];
        if (q.dead) len = 0;
        for (int k = 0; k < len; k++) press_key(cx, q, a.keys[((size_t)k * P + p) * a.n + i]);
    }
}

// tetris_environment.py:102-116 perform_action with the (r,t) encoding for `player`, [0] for the others
template <int P>
TE_HD void make_rt(const Ctx& cx, Game<P>& g, int player, int r, int t) {
    if (g.round_over) return;
    TE_UNROLL
    for (int p = 0; p < P; p++)
        if (p == player && !g.pl[p].dead) play_rt(cx, g.pl[p], r, t);
}

// Phase 1 of a lane's work: issue the state loads (nothing here needs the LDS shape table, so the
// kernel runs this BEFORE its table-init barrier and both memory round-trips overlap).
// SURVEY.md §8(d) synthetic policy: words 0 and 1 of Philox4x32-10 keyed by (policy_seed, global game id, step)
TE_HD void policy_draw(const KArgs& a, uint32_t slot, unsigned long long step, uint32_t& w0, uint32_t& w1) {
#if defined(TE_ABLATE) && (TE_ABLATE & 1)
    w0 = slot + (uint32_t)step; w1 = slot * 7u + (uint32_t)step;      // diagnostic build: no Philox
#else
    uint32_t w[4];
    philox4x32_10(a.policy_seed, a.game_offset + slot, (uint32_t)step, (uint32_t)(step >> 32), w);
    w0 = w[0]; w1 = w[1];
#endif
}

template <int P, int MODE, bool TINT = false, int MEM = MEM_STREAM>
TE_HD void game_load(const KArgs& a, int i, Game<P>& g) {
    const size_t slot = a.idx ? (size_t)a.idx[i] : (size_t)i;
    if (MODE != M_INIT && MODE != M_SPLIT_INIT)
        load_game<P>(geo_of(a), slot, g, TINT, P > 1 || MODE == M_SPLIT_RESET, MODE == M_ROLLOUT, MEM, !a.idx);   // split batches: 1-player layout WITH a queue
    // the first step's draw depends on kernel arguments only: its 40 dependent multiplies run while the state loads are in flight
    if (MODE == M_ROLLOUT) policy_draw(a, (uint32_t)slot, a.first_step, g.draw0, g.draw1);
    // (r, t) actions: the three action bytes are requested together with the state, not after it has arrived
    if (MODE == M_STEP_RT || MODE == M_STEP_RT_AUTO) { g.draw0 = a.rot[i]; g.draw1 = (uint32_t)a.trans[i] | ((a.player ? (uint32_t)a.player[i] : 0u) << 8); }
}

// Phase 2: step and store.
template <int P, int MODE, bool TINT = false, int MEM = MEM_STREAM>
TE_HD void game_run(const KArgs& a, int i, const uint32_t* shapes, Game<P>& g, LaneCounters& cnt) {
    const size_t slot = a.idx ? (size_t)a.idx[i] : (size_t)i;
    Ctx cx = make_ctx(a, shapes, TINT, P > 1 || MODE == M_SPLIT_INIT || MODE == M_SPLIT_RESET);
    if (MODE == M_INIT) init_game<P>(cx, g, (uint32_t)(a.seeds ? (uint16_t)a.seeds[i] : 0));
    if (MODE == M_SPLIT_INIT) {           // a.steps carries the side this batch holds
        init_game<P>(cx, g, (uint32_t)(a.seeds ? (uint16_t)a.seeds[i] : 0));
        g.flags = SPLIT_ON | (a.steps ? SPLIT_SIDE : 0u);
    } else if (MODE == M_SPLIT_RESET) {
        if (P == 1) reset_split(cx, *reinterpret_cast<Game<1>*>(&g), (uint32_t)(a.seeds ? (uint16_t)a.seeds[i] : 0));
    } else if (MODE == M_RESET) {
        reset_game<P>(cx, g, (uint32_t)(a.seeds ? (uint16_t)a.seeds[i] : 0));
    } else if (MODE == M_RESET_SCHED) {     // worker.py:157-166 with the SURVEY §8(d) seed schedule, decided on the device
        g.episode++;
        reset_game<P>(cx, g, episode_seed(a.game_offset + (uint32_t)slot, g.episode));
    } else if (MODE == M_MAKE) {
        make_keys<P>(cx, a, i, g);
    } else if (MODE == M_FINISH) {
        TE_UNROLL
        for (int p = 0; p < P; p++) prefetch_next(cx, g.pl[p], g.seed16, g.status);
        int done = finish_game<P>(cx, g, a.ms);
        write_outputs<P>(a, i, g, done);
    } else if (MODE == M_STEP_KEYS) {
        TE_UNROLL
        for (int p = 0; p < P; p++) prefetch_next(cx, g.pl[p], g.seed16, g.status);
        make_keys<P>(cx, a, i, g);
        int done = finish_game<P>(cx, g, a.ms);
        write_outputs<P>(a, i, g, done);
    } else if (MODE == M_STEP_RT || MODE == M_STEP_RT_AUTO) {
        TE_UNROLL
        for (int p = 0; p < P; p++) prefetch_next(cx, g.pl[p], g.seed16, g.status);
        ResetPrefetch rpf;
        rpf.ok = 0; rpf.seed16 = 0; rpf.word = 0;
        if (MODE == M_STEP_RT_AUTO) prefetch_reset(cx, episode_seed(a.game_offset + (uint32_t)slot, g.episode + 1), rpf);
        make_rt<P>(cx, g, (int)(g.draw1 >> 8), (int)(g.draw0 & 3u), (int)(g.draw1 & 0xFFu));
        int done = finish_game<P>(cx, g, a.ms);
        write_outputs<P>(a, i, g, done);                 // done / lines / dead as they stand BEFORE the reset
        if (MODE == M_STEP_RT_AUTO && done) {            // worker.py:157-166 reset_envs, without the host round trip
            g.episode++;
            reset_game<P>(cx, g, episode_seed(a.game_offset + (uint32_t)slot, g.episode), &rpf);
        }
    } else if (MODE == M_ROLLOUT) {
        // SURVEY.md §8(d) synthetic workload: worker.py:91-118 with a counter-based random policy
        for (int s = 0; s < a.steps; s++) {
            unsigned long long step = a.first_step + (unsigned long long)s;
            TE_UNROLL
            for (int p = 0; p < P; p++) prefetch_next(cx, g.pl[p], g.seed16, g.status);
            ResetPrefetch rpf;
            prefetch_reset(cx, episode_seed(a.game_offset + (uint32_t)slot, g.episode + 1), rpf);
            int r = (int)(g.draw0 & 3u), t = (int)(g.draw1 % 10u);
            int player = P > 1 ? (int)(step % (unsigned long long)P) : 0;
            TE_STAMP(4);
            uint32_t sent_before = 0;
            TE_UNROLL
            for (int p = 0; p < P; p++) sent_before += g.pl[p].lines_sent;
            make_rt<P>(cx, g, player, r, t);
            TE_STAMP(5);
            int done = finish_game<P>(cx, g, a.ms);
            TE_STAMP(7);
            cnt.steps++;
            g.steps++;
            uint32_t sent_after = 0;
            TE_UNROLL
            for (int p = 0; p < P; p++) {
                sent_after += g.pl[p].lines_sent;
                if (!g.pl[p].dead) { cnt.lines += (unsigned)g.pl[p].reward; g.add_lines += (unsigned)g.pl[p].reward; }
            }
            cnt.sent += (sent_after - sent_before) & 0xFFFFu;
            g.add_sent += (sent_after - sent_before) & 0xFFFFu;
#if defined(TE_ABLATE) && (TE_ABLATE & 16)
            done = 0;                                // diagnostic build: no auto-reset
#endif
            if (done) {
                cnt.episodes++;
                g.episode++;
                reset_game<P>(cx, g, episode_seed(a.game_offset + (uint32_t)slot, g.episode), &rpf);
            }
            if (s + 1 < a.steps) policy_draw(a, (uint32_t)slot, step + 1, g.draw0, g.draw1);
            TE_STAMP(8);
        }
    }
    store_game<P>(geo_of(a), slot, g, TINT, P > 1 || MODE == M_SPLIT_INIT || MODE == M_SPLIT_RESET,
                  MODE == M_ROLLOUT || MODE == M_INIT || MODE == M_SPLIT_INIT, MEM, !a.idx);   // 1-player: FIFO words stay as zeroed at creation
    report_status(a, g.status);
}

// Split mode (opponents on different GPUs, tetris_engine.h): one stage of a step for the side this batch holds (a.split_side).
//   STAGE 0 = A, 1 = B, 2 = C (tetris_engine.h "split mode");  STAGE 3 = C of step s followed by A of step s + 1 in one pass over
//   the state: C needs only the peer's last exchange word and A of the next step only what C leaves behind, both local once the
//   gather has landed — a step of a device-driven loop is then TWO kernels (B, C+A) and two state round trips instead of three.
// Player 1's loop-1 pass (stage A) is speculative: the reference skips it when player 0 died in loop 1 (PythonHandle.cpp:153-156),
// which side 1 only learns from the exchange.  Side 1 continues from the main state and keeps an UNDO RECORD in the shadow array:
// what the pass changes when it clears no row and the new piece fits — nearly always — is the piece (kind, rotation, position,
// next), the draw counter, now and then the piece group, and 200 ms of combo time: four words per board instead of a second
// copy of the state.  Where rows were cleared or the spawn failed (rare), the board as it was after the key interpreter is
// rebuilt from memory (stage A has not stored anything yet) and kept whole behind the record.  Stage B reads the record only
// for games whose player 0 did die.
constexpr int UNDO_WORDS = 4;        // shadow rows: 0 pose, 1 piece group, 2 piece draws, 3 flags; the full copy follows from row UNDO_WORDS
constexpr uint32_t UNDO_RAN = 1u, UNDO_SIMPLE = 2u;
TE_HD Ref undo_ref(const KArgs& a, int i) { Ref r = {a.shadow, (uint32_t)i * 4u, (size_t)a.n_stride}; return r; }
TE_HD Ref undo_copy_ref(const KArgs& a, int i) { Ref r = {a.shadow + (size_t)UNDO_WORDS * a.n_stride, (uint32_t)i * 4u, (size_t)a.n_stride}; return r; }
TE_HD uint32_t pose_pack(const Player& q) {
    return (uint32_t)q.kind | ((uint32_t)q.rot << 3) | ((uint32_t)(q.x + 4) << 5) | ((uint32_t)q.y << 9) | ((uint32_t)q.next << 14);
}
// settle() with no row cleared and a spawn that fitted (gamePlay.cpp:54-59,71-88,160-171; Combo.cpp:50-52), undone in place
TE_HD void undo_simple_settle(Player& q, uint32_t pose, uint32_t group, uint32_t draws) {
    q.kind = (int)(pose & 7u); q.rot = (int)((pose >> 3) & 3u); q.x = (int)((pose >> 5) & 15u) - 4;
    q.y = (int)((pose >> 9) & 31u); q.next = (int)((pose >> 14) & 7u);
    q.pgroup = group; q.piece_draws = draws; q.pf_ok = 0;
    q.combo_time += 200;
}

template <bool TINT>
TE_HD void split_stage_a(const KArgs& a, const Ctx& cx, int i, Game<1>& g, int side, unsigned long long step) {
    Player& q = g.pl[0];
    const uint32_t sent_before = q.lines_sent;
    int acting, r, t;
    if (a.rot) { acting = a.player ? a.player[i] : 0; r = a.rot[i] & 3; t = a.trans[i]; }
    else {          // built-in synthetic policy, identical on both sides of a game
        uint32_t w[4];
        philox4x32_10(a.policy_seed, a.game_offset + (uint32_t)i, (uint32_t)step, (uint32_t)(step >> 32), w);
        r = (int)(w[0] & 3u); t = (int)(w[1] % 10u); acting = (int)(step % 2ull);
    }
    prefetch_next(cx, q, g.seed16, g.status);
    const bool plays = !g.round_over && !q.dead && acting == side;
    if (plays) play_rt(cx, q, r, t);
    const uint32_t pose = pose_pack(q), group = q.pgroup, draws = q.piece_draws, cleared = q.lines_cleared;
    const bool ran = !g.round_over && !q.dead;
    if (side == 1) {
        // Will the pass be the simple kind?  Known BEFORE it runs: no row in the scanned range is full (gameField.cpp:120-145)
        // and, the board then staying as it is, the next piece fits where it appears (gamePlay.cpp:71-88).  A board for which
        // that does not hold goes to the shadow whole, now, while it still is what the key interpreter left — a store by the few
        // lanes concerned instead of a copy kept by all.
        uint32_t full = (~0u << q.y) & ~cx.floor_bits;
        for (int c = 0; c < NCOL; c++) full &= q.col[c];
        const bool simple = ran && !full && fits_spawn_col(cx, q, shape_of(cx, q.next, spawn_rot(q.next)), 0);
        const Ref ur = undo_ref(a, i);
        word_at(ur, 0) = pose; word_at(ur, 1) = group; word_at(ur, 2) = draws;
        word_at(ur, 3) = (ran ? UNDO_RAN : 0u) | (simple ? UNDO_SIMPLE : 0u);
        if (ran && !simple) {
            const Ref cr = undo_copy_ref(a, i);
            store_player(cr.s, cr.o, cr.ws, q, TINT);
        }
        a.xout[i] = split_settle(cx, g);
        if (simple && (q.lines_cleared != cleared || q.dead)) TE_COUNT(PC_UNDO_MISPREDICT);      // (tests assert this never counts)
    } else
        a.xout[i] = split_settle(cx, g);
    // rollout counter: side 1's loop-1 lines are counted at stage B, once it knows the pass is committed
    if (side == 0) g.add_sent += (q.lines_sent - sent_before) & 0xFFFFu;
}

template <int STAGE, bool TINT = false>
TE_HD void split_body(const KArgs& a, int i, const uint32_t* shapes) {
    Ctx cx = make_ctx(a, shapes, TINT);
    Game<1> g;
    const int side = a.split_side;
    const bool counting = (STAGE == 2 || STAGE == 3) && a.steps;         // built-in rollout: G_STEPS is kept by stage C
    load_game<1>(geo_of(a), (size_t)i, g, TINT, true, counting);
    Player& q = g.pl[0];
    if (STAGE == 0) {
        split_stage_a<TINT>(a, cx, i, g, side, a.first_step);
        store_game<1>(geo_of(a), (size_t)i, g, TINT);
    } else if (STAGE == 1) {
        const uint32_t my_a = a.xw[0][i], opp_a = a.xw[1][i];
        uint32_t w;
        if (side == 0) {
            const uint32_t sent_before = q.lines_sent;
            const bool i_died = (my_a & XW_DIED) != 0;
            const int in = (!i_died && (opp_a & XW_RAN) && !(opp_a & XW_DIED)) ? xw_sent(opp_a) : 0;
            w = split_tick(cx, g, a.ms, in);
            g.add_sent += (q.lines_sent - sent_before) & 0xFFFFu;
        } else {
            const uint32_t opp_b = a.xw[2][i];
            const bool committed = !(opp_a & XW_DIED);
            if (!committed) {                             // player 0 died in loop 1: player 1 did not run (:153-156) — undo its pass
                const Ref ur = undo_ref(a, i);
                const uint32_t flags = word_at(ur, 3);
                if (flags & UNDO_SIMPLE) { TE_COUNT(PC_UNDO_SIMPLE); undo_simple_settle(q, word_at(ur, 0), word_at(ur, 1), word_at(ur, 2)); }
                else if (flags & UNDO_RAN) {
                    TE_COUNT(PC_UNDO_FULL);
                    const Ref cr = undo_copy_ref(a, i);
                    load_player(cr.s, cr.o, cr.ws, q, TINT);
                }
            }
            const uint32_t sent_before = q.lines_sent;
            const int in1 = ((opp_a & XW_RAN) && !(opp_a & XW_DIED)) ? xw_sent(opp_a) : 0;
            if (!g.round_over && in1 > 0) q.incoming = q.incoming + (float)in1 / 1.0f;   // loop 1, PythonHandle.cpp:121
            const int in2 = (opp_b & XW_DIED) ? 0 : xw_sent(opp_b);                      // loop 2, :175
            if (committed && (my_a & XW_RAN) && !(my_a & XW_DIED)) g.add_sent += (uint32_t)xw_sent(my_a);   // committed loop-1 lines
            w = split_tick(cx, g, a.ms, in2);
            g.add_sent += (q.lines_sent - sent_before) & 0xFFFFu;
        }
        a.xout[i] = w;
        store_game<1>(geo_of(a), (size_t)i, g, TINT);
    } else {
        const uint32_t opp_b = a.xw[side == 0 ? 3 : 2][i];
        const int in = (side == 0 && !(opp_b & XW_DIED)) ? xw_sent(opp_b) : 0;
        const int done = split_finish(g, in, (opp_b & XW_DEAD_NOW) != 0, (opp_b & XW_ERR) != 0);
        if (a.done) a.done[i] = (uint8_t)done;
        if (a.lines) a.lines[i] = (uint8_t)q.reward;
        if (a.dead) a.dead[i] = (uint8_t)q.dead;
        if (a.steps) {          // built-in rollout: counters and auto-reset, the same decision on both sides of the game
            g.steps++;
            if (!q.dead) g.add_lines += (unsigned)q.reward;
            if (done) { g.episode++; reset_split(cx, g, episode_seed(a.game_offset + (uint32_t)i, g.episode)); }
        }
        if (STAGE == 3) split_stage_a<TINT>(a, cx, i, g, side, a.first_step + 1ull);      // stage A of the NEXT step
        store_game<1>(geo_of(a), (size_t)i, g, TINT, true, counting);
    }
    report_status(a, g.status);
}

template <int P, int MODE, bool TINT = false>
TE_HD void game_body(const KArgs& a, int i, const uint32_t* shapes, LaneCounters& cnt) {
    Game<P> g;
    game_load<P, MODE, TINT>(a, i, g);
    game_run<P, MODE, TINT>(a, i, shapes, g, cnt);
}

// State views / __getstate__ (PythonHandle.h:54-82,123-308) of one game -> P records
template <int P, bool TINT = false>
TE_HD void observe_body(const Geo& geo, int i, const int32_t* idx, int H,
                        const uint32_t* shapes, tetris_record* rec, uint8_t* round_over, int8_t* last_winner) {
    size_t slot = idx ? (size_t)idx[i] : (size_t)i;
    Game<P> g;
    load_game<P>(geo, slot, g, TINT);
    if (round_over) round_over[i] = (uint8_t)g.round_over;
    if (last_winner) last_winner[i] = (int8_t)g.last_winner;
    if (!rec) return;
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        const Player& q = g.pl[p];
        tetris_record* r = &rec[(size_t)i * P + p];
        uint16_t* f16 = (uint16_t*)&r->field[0][0];
        for (int y = 0; y < TETRIS_MAX_H; y++)
            for (int c = 0; c < NCOL; c += 2) {
                uint32_t lo = (y < H) ? ((q.col[c] >> y) & 1u) : 0u;
                uint32_t hi = (y < H) ? ((q.col[c + 1] >> y) & 1u) : 0u;
                if (TINT) {        // cell value = 1 + 3-bit plane value for occupied squares (State.field as the reference shows it)
                    lo *= 1u + (((q.tint[0][c] >> y) & 1u) | (((q.tint[1][c] >> y) & 1u) << 1) | (((q.tint[2][c] >> y) & 1u) << 2));
                    hi *= 1u + (((q.tint[0][c + 1] >> y) & 1u) | (((q.tint[1][c + 1] >> y) & 1u) << 1) | (((q.tint[2][c + 1] >> y) & 1u) << 2));
                }
                f16[(y * NCOL + c) >> 1] = (uint16_t)(lo | (hi << 8));
            }
        uint32_t shape = shapes[((q.kind & 7) << 2) | (q.rot & 3)];
        int val = shape_value(q.kind);
        for (int gy = 0; gy < 4; gy++) {
            uint32_t word = 0;
            for (int gx = 0; gx < 4; gx++)
                if ((shape >> (4 * gx + gy)) & 1u) word |= (uint32_t)val << (8 * gx);
            *(uint32_t*)&r->grid[gy][0] = word;
        }
        r->x = (int8_t)q.x; r->y = (int8_t)q.y;
        r->piece = (uint8_t)q.kind; r->tile = (uint8_t)(q.kind + 1);
        r->spawn_rot = (uint8_t)(q.kind <= 6 ? spawn_rot(q.kind) : 0); r->cur_rot = (uint8_t)q.rot;
        r->big = (uint8_t)(q.kind == 4 || q.kind == 6);
        r->next = (uint8_t)q.next; r->dead = (uint8_t)q.dead; r->reward = (uint8_t)q.reward;
        r->inc_count = (uint8_t)q.inc_count; r->combo_count = (uint8_t)q.combo_count;
        r->combo_remaining = (uint16_t)q.combo_remaining;
        r->lock_armed = (uint8_t)q.lock_armed; r->fifo_len = (uint8_t)q.qlen; r->line_count = (uint8_t)q.line_count;
        r->fifo_overflow = (uint8_t)q.q_overflow;
        r->time_ms = q.time_ms; r->incoming = q.incoming;
        r->drop_delay = q.drop_delay; r->drop_time = q.drop_time; r->speedup_time = q.speedup_time;
        r->lock_time = q.lock_time; r->min_remaining = q.min_remaining;
        r->combo_start = q.combo_start; r->combo_time = q.combo_time;
        for (int k = 0; k < TETRIS_FIFO_CAP; k++) { r->fifo_delay[k] = 0; r->fifo_count[k] = 0; }
        for (int k = 0; k < FIFO_CAP; k++)
            if (k < q.qlen) { r->fifo_delay[k] = q.qdelay[k]; r->fifo_count[k] = (int16_t)q.qcount[k]; }
        r->lines_sent = (uint16_t)q.lines_sent; r->lines_cleared = (uint16_t)q.lines_cleared;
        r->lines_blocked = (uint16_t)q.lines_blocked; r->garbage_cleared = (uint16_t)q.garbage_cleared;
        r->max_combo = (uint16_t)q.max_combo; r->lines_cleared_seen = (uint16_t)q.lines_seen;
        for (int k = 0; k < 7; k++) r->weights[k] = 0.0f;
        r->piece_draws = q.piece_draws; r->hole_draws = q.hole_draws;
    }
}

// state_dict + unpacker for one player-board (state_processors.py:23-54, state_unpack.py:88-137):
// `cells` receives H*10 bytes (field > 0, row-major), `vec` 12 bytes, returns the piece index
TE_HD int observe_board(const Geo& geo, size_t slot, int p, int H, uint8_t* cells, uint8_t* vec) {
    const Ref br = board_ref(geo, p, slot);
    uint32_t col[NCOL];
    for (int c = 0; c < NCOL; c++) col[c] = word_at(br, W_COL0 + c);
    for (int y = 0; y < H; y++)
        for (int c = 0; c < NCOL; c++) cells[y * NCOL + c] = (uint8_t)((col[c] >> y) & 1u);
    const uint32_t w = word_at(br, W_PIECE);
    const uint32_t m = word_at(br, W_MISC);
    const uint32_t dc = word_at(br, W_DROPCOMBO);
    const int x = (int)((w >> 5) & 15) - 4, y = (w >> 9) & 31, next = (w >> 14) & 7;
    uint32_t t = ((dc >> 16) + 50u) & 0xFFFFu;            // uint16 array + 50 wraps like numpy (state_processors.py:38)
    if (t > 25000u) t = 25000u;
    vec[0] = (uint8_t)x; vec[1] = (uint8_t)y; vec[2] = (uint8_t)(m & 255); vec[3] = (uint8_t)(t / 100u); vec[4] = (uint8_t)((m >> 8) & 255);
    for (int k = 0; k < 7; k++) vec[5 + k] = (uint8_t)(next == k);
    return (int)(w & 7);
}

// ---------------------------------------------------------------- drop enumeration (BASELINE config 4)
// TestField.cpp:64-125 (drop placements) + simulate_actions(finalize=False) (tetris_environment.py:87-100): 40 placements
// (rotation r = 0..3, column index xi = 0..9, x = xi - 1) per board.  Everything that is the same for the 40 placements
// of a board is computed ONCE per board into a `BoardPre` (on the GPU: cooperatively by the board's 40 lanes, in LDS):
//   col[10]   the board's columns          band     the collision window of row 0 (see tetris_engine.h)
//   strip     per-column free depth from row 0 as bytes [wall wall d0..d9 wall wall wall wall] (hard drop, see drop_distance_bytes)
//   pre/suf   pre[i] = AND of columns < i, suf[i] = AND of columns >= i: the full-row mask of a stamped board is
//             pre[x] & suf[x + 4] & (the four columns under the piece | its cells) — no pass over all ten columns
// A placement lane then needs ~10 LDS reads and no loop over the board.
constexpr int PRE_COL = 0, PRE_PIECE = 10, PRE_BAND = 11, PRE_STRIP = 13, PRE_PRE = 17, PRE_SUF = 28, PRE_WORDS = 40;
#ifndef TE_ENUM_BOARDS
#define TE_ENUM_BOARDS 6
#endif
// boards / threads per workgroup of k_enumerate (10 lanes per board).  6 boards = one WAVE per workgroup (60 working lanes, 4 idle):
// the two phases of the per-board precompute are ordered within the wave and need no workgroup barrier — 7.65 -> 7.25 us on
// C4's 16 384 boards against 32 boards (five waves, two __syncthreads) per workgroup, same box
constexpr int ENUM_BOARDS = TE_ENUM_BOARDS, ENUM_BLOCK = ENUM_BOARDS == 6 ? 64 : ENUM_BOARDS * 10;
constexpr bool ENUM_ONE_WAVE = ENUM_BOARDS == 6;

// element functions of the per-board precompute (lane j of a board calls the ones its index selects)
TE_HD uint32_t pre_band_bits(uint32_t col, uint32_t floor_bits, int c, int& word) {      // nibble c+2 of the 64-bit band
    const uint32_t nib = (col | floor_bits) & 0xFu;
    word = c >= 6;
    return nib << (c >= 6 ? 4 * (c - 6) : 4 * c + 8);
}
TE_HD uint32_t pre_depth(uint32_t col, uint32_t floor_bits) { return (uint32_t)ctz32(col | floor_bits); }   // H <= 31: never zero
TE_HD uint32_t pre_and_below(const uint32_t* col, int i) {      // AND of columns < i (i = 0..10)
    uint32_t v = ~0u;
    for (int c = 0; c < NCOL; c++) v &= (c < i) ? col[c] : ~0u;
    return v;
}
TE_HD uint32_t pre_and_from(const uint32_t* col, int i) {       // AND of columns >= i (i = 0..10)
    uint32_t v = ~0u;
    for (int c = 0; c < NCOL; c++) v &= (c >= i) ? col[c] : ~0u;
    return v;
}

// serial form of the precompute (CPU harness; the GPU kernel spreads the same element functions over a board's lanes)
TE_HD void enum_prepare(const Geo& geo, size_t slot, int p, int H, uint32_t* pre) {
    const Ref br = board_ref(geo, p, slot);
    const uint32_t floor_bits = ~0u << H;
    for (int c = 0; c < NCOL; c++) pre[PRE_COL + c] = word_at(br, W_COL0 + c);
    pre[PRE_PIECE] = word_at(br, W_PIECE);
    pre[PRE_BAND] = 0xFFu; pre[PRE_BAND + 1] = 0xFFFF0000u;
    for (int k = 0; k < 4; k++) pre[PRE_STRIP + k] = 0;
    for (int c = 0; c < NCOL; c++) {
        int word;
        const uint32_t bits = pre_band_bits(pre[PRE_COL + c], floor_bits, c, word);
        pre[PRE_BAND + word] |= bits;
        pre[PRE_STRIP + ((c + 2) >> 2)] |= pre_depth(pre[PRE_COL + c], floor_bits) << (8 * ((c + 2) & 3));
    }
    for (int i = 0; i <= NCOL; i++) { pre[PRE_PRE + i] = pre_and_below(pre + PRE_COL, i); pre[PRE_SUF + i] = pre_and_from(pre + PRE_COL, i); }
}

// What the four rotations of one column index share (hoisted out of the rotation loop): the piece, the depths under its 4x4
// box, the four board columns there and the AND of all the others.
struct ColumnCtx {
    int kind, cur_rot, n_rot, x;
    unsigned xs;                     // x + 2: the piece's position in the band window
    uint64_t band;                   // collision window of row 0
    uint32_t under;                  // free depth of the four board columns under the box, a byte each (walls: 0)
    uint32_t colk[4];                // board columns x .. x + 3 (all ones outside the board: neutral in the full-row AND)
    uint32_t rest;                   // AND of the board's other columns
};
struct Placement { int ok, y, cleared; uint64_t placed; };      // placed: the shape's nibbles at the piece's band position

TE_HD ColumnCtx enum_column(const uint32_t* pre, int xi) {
    ColumnCtx cc;
    const uint32_t w = pre[PRE_PIECE];
    cc.kind = w & 7; cc.cur_rot = (w >> 3) & 3;
    cc.n_rot = cc.kind == 6 ? 1 : (cc.kind == 4 || cc.kind == 2 || cc.kind == 3) ? 2 : 4;     // TestField.cpp:71-109
    cc.x = xi - 1;
    cc.xs = (unsigned)(cc.x + 2);                                                            // 1..10
    cc.band = ((uint64_t)pre[PRE_BAND + 1] << 32) | pre[PRE_BAND];
    const uint32_t lo = pre[PRE_STRIP + (cc.xs >> 2)], hi = (cc.xs >> 2) < 3 ? pre[PRE_STRIP + (cc.xs >> 2) + 1] : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    cc.under = __builtin_amdgcn_alignbyte(hi, lo, cc.xs & 3u);
#else
    cc.under = (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (cc.xs & 3u)));
#endif
    for (int k = 0; k < 4; k++) {
        const int c = cc.x + k;
        cc.colk[k] = (c >= 0 && c < NCOL) ? pre[PRE_COL + c] : ~0u;
    }
    cc.rest = pre[PRE_PRE + imax(0, imin(NCOL, cc.x))] & pre[PRE_SUF + imax(0, imin(NCOL, cc.x + 4))];
    return cc;
}

// one placement (rotation r of the lane's column) against a prepared board (`pre` = its BoardPre; LDS on the GPU)
TE_HD Placement enum_place(const uint32_t* pre, const ColumnCtx& cc, const uint32_t* shapes, int H, int r) {
    Placement out;
    const uint32_t floor_bits = ~0u << H;
    const int rot = cc.kind == 6 ? cc.cur_rot : r;                                         // O is used as it stands
    const uint32_t nibs = shapes[((cc.kind & 7) << 2) | rot] & 0xFFFFu;
    out.placed = (uint64_t)nibs << (4 * cc.xs);
    out.ok = cc.kind <= 6 && r < cc.n_rot && cc.x <= NCOL - 2 && (out.placed & cc.band) == 0;   // gameField.cpp:10-20 at (x, 0)
    out.y = 0; out.cleared = 0;
    if (!out.ok) { out.placed = 0; return out; }
    // hard drop (gameField.cpp:49-53): the four depths under the piece + the shape's drop word, byte minimum
    const uint32_t sum = cc.under + shapes[32 + (((cc.kind & 7) << 2) | rot)];
    const uint32_t b0 = sum & 0xFFu, b1 = (sum >> 8) & 0xFFu, b2 = (sum >> 16) & 0xFFu, b3 = sum >> 24;
    const uint32_t m01 = b0 < b1 ? b0 : b1, m23 = b2 < b3 ? b2 : b3;
    const uint32_t m = m01 < m23 ? m01 : m23;
    int y;
    if (m - 0x40u > 0x3Eu) {          // an obstacle above a piece column's top cell inside the 4x4 box: exact closed form per column
        int dist = 64;
        for (int gx = 0; gx < 4; gx++) {
            const uint32_t nib = (nibs >> (4 * gx)) & 0xFu;
            if (nib) {                // (an occupied piece column of a piece that fits lies inside the board)
                const int top = ctz32(nib), bottom = 31 - clz32(nib);
                const uint32_t below = (cc.colk[gx] | floor_bits) >> (top + 1);
                const int first = below ? ctz32(below) + top + 1 : 32;
                dist = imin(dist, imax(0, first - bottom - 1));
            }
        }
        y = dist == 64 ? 0 : dist;
    } else
        y = (int)(m - 0x40u);
    out.y = y;
    // stamp (gameField.cpp:105-110) and the rows a finalize would clear (gameField.cpp:120-145)
    uint32_t full = cc.rest;
    for (int k = 0; k < 4; k++) full &= cc.colk[k] | (((nibs >> (4 * k)) & 0xFu) << y);
    const uint32_t range = (~0u << y) & ~floor_bits;
    if (full & ~range & ~floor_bits) {
        // a full row ABOVE the piece (unreachable by play; only a crafted restore could hold one): the reference re-scans rows
        // that shift into range, so count exactly like clear_rows
        uint32_t colv[NCOL];
        for (int c = 0; c < NCOL; c++) colv[c] = pre[PRE_COL + c] | (((uint32_t)(out.placed >> (4 * c + 8)) & 0xFu) << y);
        int cleared = 0;
        for (;;) {
            uint32_t f = range;
            for (int c = 0; c < NCOL; c++) f &= colv[c];
            if (!f) break;
            const int rr = 31 - clz32(f);
            const uint32_t above = (1u << rr) - 1u, keep = ~((above << 1) | 1u);
            for (int c = 0; c < NCOL; c++) colv[c] = (colv[c] & keep) | ((colv[c] & above) << 1);
            cleared++;
        }
        out.cleared = cleared;
    } else
        out.cleared = __builtin_popcount(full & range);
    return out;
}

// column c (a compile-time constant after unrolling) of the stamped board of a placement: nibble c + 2 of `placed`, at row y
TE_HD uint32_t enum_after_col(uint32_t board_col, const Placement& pl, int c) {
    return board_col | (((uint32_t)(pl.placed >> (4 * c + 8)) & 0xFu) << pl.y);
}

// serial driver of one board's 40 placements (CPU harness)
// `planar`: rotation-minor planes, valid / land_y / cleared [n][10][4] and after [10][n][10][4] (see k_enumerate)
template <int P>
TE_HD void enumerate_body(const Geo& geo, int i, int n, const int32_t* idx, const uint8_t* player, int H,
                          const uint32_t* shapes, uint8_t* valid, int8_t* land_y, uint8_t* cleared, uint32_t* after, bool planar) {
    uint32_t pre[PRE_WORDS];
    enum_prepare(geo, safe_slot(idx, i, (int)geo.n_games), safe_player(player, i, P), H, pre);
    for (int r = 0; r < 4; r++)
        for (int xi = 0; xi < NCOL; xi++) {
            const Placement pl = enum_place(pre, enum_column(pre, xi), shapes, H, r);
            const size_t t = planar ? ((size_t)i * 10 + xi) * 4 + r : ((size_t)i * 4 + r) * 10 + xi;
            valid[t] = (uint8_t)pl.ok;
            land_y[t] = (int8_t)pl.y;
            cleared[t] = (uint8_t)pl.cleared;
            if (after)
                for (int c = 0; c < NCOL; c++)
                    after[planar ? (size_t)c * n * 40 + t : t * NCOL + c] = enum_after_col(pre[PRE_COL + c], pl, c);
        }
}

// PythonHandle.cpp:190 get_actions -> TestField.cpp:64-111 getMask(2): lane t = (game i, rotation r, column xi);
// slab of lane t: count[t], lens[t][max_lists], keys[t][max_lists][max_keys]
template <int P>
TE_HD void actions_body(const Geo& geo, size_t t, const int32_t* idx, const uint8_t* player, int H,
                        const uint32_t* shapes, uint8_t* count, uint8_t* lens, uint8_t* keys, int max_lists, int max_keys,
                        uint32_t* status) {
    const int i = (int)(t / 40), j = (int)(t % 40), r = j / 10, xi = j % 10;
    const size_t slot = safe_slot(idx, i, (int)geo.n_games);
    const int p = safe_player(player, i, P);
    const Ref br = board_ref(geo, p, slot);
    Probe pr;
    for (int c = 0; c < NCOL; c++) pr.q.col[c] = word_at(br, W_COL0 + c);
    const uint32_t w = word_at(br, W_PIECE);
    const int kind = w & 7, cur_rot = (w >> 3) & 3;
    Ctx cx;
    cx.shapes = shapes; cx.H = H; cx.floor_bits = ~0u << H; cx.tint = false; cx.queue = true;
    const int n_rot = kind == 6 ? 1 : (kind == 4 || kind == 2 || kind == 3) ? 2 : 4;
    pr.q.kind = kind; pr.q.rot = kind == 6 ? cur_rot : r; pr.q.x = xi - 1; pr.q.y = 0;
    pr.spawn = spawn_rot(kind);
    pr.path_len = 0; pr.best_len = 0; pr.best_x = 0; pr.best_rot = 0;
    pr.keys = keys + t * (size_t)max_lists * max_keys; pr.lens = lens + t * (size_t)max_lists;
    pr.max_lists = max_lists; pr.max_keys = max_keys; pr.n_lists = 0; pr.overflow = 0;
    if (kind <= 6 && r < n_rot && pr.q.x <= NCOL - 2 && probe_fits(cx, pr)) probe_column(cx, pr);
    count[t] = (uint8_t)imin(pr.n_lists, 255);
    if (pr.overflow || pr.n_lists > max_lists) ((volatile uint32_t*)status)[F_BADARG] = 1u;
}

// per-game cumulative rollout counters of one game -> {env-steps, episodes, lines, sent}
TE_HD void totals_of_game(const Geo& geo, int i, unsigned long long out[4]) {
    const Ref gr = game_ref(geo, (size_t)i);
    out[0] = word_at(gr, G_STEPS);
    out[1] = word_at(gr, G_EPISODE);
    out[2] = word_at(gr, G_LINES);
    out[3] = word_at(gr, G_SENT);
}

// PythonHandle.cpp:36-42 copy / set: raw words, blob[i][NGWORDS + P*nw]; t = lane = (game i, word)
TE_HD void snapshot_body(const Geo& geo, size_t t, const int32_t* idx, uint32_t* blob, int restore) {
    const int words = NGWORDS + geo.P * geo.nw;
    int i = (int)(t / (size_t)words), w = (int)(t % (size_t)words);
    size_t slot = idx ? (size_t)idx[i] : (size_t)i;
    uint32_t* cell;
    if (w < NGWORDS) cell = &word_at(game_ref(geo, slot), w);
    else {
        int pw = w - NGWORDS, p = pw / geo.nw, ww = pw % geo.nw;
        cell = &word_at(board_ref(geo, p, slot), ww);
    }
    if (restore) *cell = blob[t];
    else blob[t] = *cell;
}

// data_types/state.py:11,16: Python writes State.dead
TE_HD void set_dead_body(const Geo& geo, int t, const int32_t* idx, const uint8_t* dead) {
    int i = t / geo.P, p = t % geo.P;
    size_t slot = idx ? (size_t)idx[i] : (size_t)i;
    uint32_t* cell = &word_at(board_ref(geo, p, slot), W_PIECE);
    *cell = (*cell & ~(1u << 17)) | ((uint32_t)(dead[t] ? 1u : 0u) << 17);
}

}  // namespace te
