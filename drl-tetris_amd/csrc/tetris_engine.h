// Per-game step logic of the MI355X batched Tetris environment — one game per lane, boards as
// packed uint32 column bitboards.  Written from the behavioural spec in SURVEY.md App. A/B; every
// function cites the reference file:line (relative to
// /root/reference/environment/game_backend/source/) whose observable behaviour it must reproduce.
//
// This header is `__host__ __device__`.  The product compiles it with hipcc for gfx950 only
// (tetris_hip.hip); tests/cpu_harness compiles the very same source with g++ so the logic can be
// checked against the oracle without a GPU.  There is no CPU path in the shipped library.
//
// Bitboard conventions
//   col[c]            bit y = 1  <=>  square (row y, column c) is occupied; row 0 is the top.
//   floor             bits >= H of every column read as 1 (added in registers, never stored).
//   shape "colnibs"   16 bits, nibble gx = the piece's column gx of its 4x4 grid, bit gy = row gy.
//   band window B(y)  64 bits, nibble (c+2) = rows y..y+3 of column c; nibbles 0,1,12..15 = 0xF
//                     (walls).  A piece at (x,y) fits  <=>  ((colnibs << 4(x+2)) & B(y)) == 0.
//                     One 64-bit AND per collision test; sideways moves only change the shift.
#pragma once
#include "tetris_layout.h"
#include "tetris_tables.h"

#if defined(__clang__)
#define TE_UNROLL _Pragma("unroll")
#else
#define TE_UNROLL _Pragma("GCC unroll 8")
#endif

namespace te {

// ---------------------------------------------------------------- small helpers
#if defined(__HIP_DEVICE_COMPILE__)
TE_HD int ctz32(uint32_t v) { return __builtin_ctz(v); }
TE_HD int clz32(uint32_t v) { return __builtin_clz(v); }
#else
TE_HD int ctz32(uint32_t v) { return __builtin_ctz(v); }
TE_HD int clz32(uint32_t v) { return __builtin_clz(v); }
#endif
// State words are touched exactly once per launch and next read by another launch (possibly on another
// XCD): stream them past the caches (`nt`), which also leaves no dirty L2 lines for the end-of-kernel
// write-back to drain.
#ifndef TE_LD_NT
#define TE_LD_NT 1
#endif
#ifndef TE_ST_NT
#define TE_ST_NT 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
TE_HD uint32_t ld_stream(const uint32_t* p) { return TE_LD_NT ? __builtin_nontemporal_load(p) : *p; }
TE_HD void st_stream(uint32_t* p, uint32_t v) { if (TE_ST_NT) __builtin_nontemporal_store(v, p); else *p = v; }
#else
TE_HD uint32_t ld_stream(const uint32_t* p) { return *p; }
TE_HD void st_stream(uint32_t* p, uint32_t v) { *p = v; }
#endif

// Diagnostic build only (-DTE_PHASE_TRACE, profiles/phase_trace.py): the first active lane of a wave writes the shader
// clock at phase boundaries into a device buffer.  Product builds compile TE_STAMP to nothing.
#if defined(TE_PHASE_TRACE) && defined(__HIPCC__)
static __device__ unsigned long long d_trace[2048 * 16];            // (one copy per translation unit; read out by tetris_hip.hip)
static __device__ unsigned long long d_chain_trace[8 * 1024 * 8];
#endif
#if defined(TE_PHASE_TRACE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void te_stamp(int k, bool realtime = false) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = realtime ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    const unsigned wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int first = __ffsll((unsigned long long)__ballot(1)) - 1;
    if ((int)(threadIdx.x & 63) == first && wv < 2048) d_trace[wv * 16 + k] = t;
    __builtin_amdgcn_sched_barrier(0);
}
#define TE_STAMP(k) te_stamp(k)
#define TE_STAMP_RT(k) te_stamp(k, true)
// chained launches (k_chain): 100 MHz real-time clock (comparable across CUs and launches), stamps of the last 8 epochs kept
__device__ __forceinline__ void te_stamp_chain(uint32_t epoch, int k) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) d_chain_trace[(((size_t)(epoch & 7u) * 1024) + blockIdx.x) * 8 + k] = t;
    __builtin_amdgcn_sched_barrier(0);
}
// slot 7: where the wave runs — HW_ID (wave[3:0] simd[5:4] cu[11:8] sh[12] se[15:13]) | XCC_ID << 32
__device__ __forceinline__ void te_stamp_place(uint32_t epoch) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)
        d_chain_trace[(((size_t)(epoch & 7u) * 1024) + blockIdx.x) * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
}
#define TE_STAMP_CHAIN(e, k) te_stamp_chain(e, k)
#define TE_STAMP_PLACE(e) te_stamp_place(e)
#else
#define TE_STAMP(k) do {} while (0)
#define TE_STAMP_RT(k) do {} while (0)
#endif
#if !defined(TE_STAMP_CHAIN)
#define TE_STAMP_PLACE(e) do {} while (0)
#define TE_STAMP_CHAIN(e, k) do {} while (0)
#endif

// Test-only path counters (tests/cpu_harness defines TE_PATH_COUNTERS): how often the rare branches of the key interpreter
// ran, so the parity tests can assert that they reached them.  Nothing in device or product builds.
#if defined(TE_PATH_COUNTERS) && !defined(__HIPCC__)
enum PathCounter { PC_KICK = 0, PC_KICK_2ND, PC_KICK_3RD, PC_KICK_FAILED, PC_KICK_DOWN, PC_DROP_EXACT, PC_RT_OFF_SPAWN,
                   PC_GARBAGE_ROW, PC_GARBAGE_LIFT2, PC_DEATH_GARBAGE, PC_DEATH_SPAWN, PC_TIMER_LOCK, PC_KEY_KICK, PC_KEY_KICK_FAILED,
                   PC_UNDO_SIMPLE, PC_UNDO_FULL, PC_UNDO_MISPREDICT, PC_NCOUNTERS };
extern unsigned long long te_path_count[PC_NCOUNTERS];
#define TE_COUNT(i) (te_path_count[i]++)
#else
#define TE_COUNT(i) do {} while (0)
#endif

// State word `word_off` (in words, uniform across the wave) of the board at byte offset `o` (per lane, < 4 GiB) of a
// uniform base: on the GPU the base + word offset stay in SGPRs and the lane offset is the 32-bit VGPR offset of the
// global_load/store, so the ~60 state accesses of a step need no per-lane 64-bit address arithmetic.
// Memory mode of a state access.  MEM_STREAM: non-temporal (the default: a state word is touched once per launch).
// MEM_AGENT: agent-scope (`sc1`) — the word is handed from one launch to the next WHILE both are running (chained launches,
// tetris_hip.hip): the store is written through to memory and the load bypasses the non-coherent per-XCD L2
// (MI355X_MICROARCH.md, inter-workgroup visibility: every store and every load of the handed-off bytes must be `sc1`).
// MEM_AFFINE (experiment only, -DTE_EXPERIMENT_AFFINE): `sc1` loads, PLAIN stores — the line stays in the storing XCD's L2, which
// only the same XCD may then read; measured slower than MEM_AGENT in the chained kernel (profiles/r03/handoff_experiments.txt).
enum MemMode : int { MEM_STREAM = 0, MEM_AGENT = 1, MEM_AFFINE = 2 };
#if defined(__HIP_DEVICE_COMPILE__)
TE_HD uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TE_HD void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
TE_HD uint32_t ld_agent(const uint32_t* p) { return *p; }
TE_HD void st_agent(uint32_t* p, uint32_t v) { *p = v; }
#endif
// On the GPU these are BUFFER accesses: the (wave-uniform) row base goes into a buffer resource in SGPRs, the row offset
// `word_off * 4` into the instruction's scalar offset and the lane's byte offset into its vector offset — one scalar add per
// access.  (As global_load/store the same accesses cost a 64-bit scalar add pair or, for about half of them, 64-bit VECTOR
// address arithmetic: ~75 instructions of a ~1000-instruction step.)  Cache policy bits of the instruction: nt for streamed
// state, sc1 for agent-scope hand-offs.  Requires the state allocation to stay below 4 GiB (checked by tetris_create).
#if defined(__HIP_DEVICE_COMPILE__)
// (stride 0, unbounded num_records: offsets are validated by the host; 0x00020000 = raw 32-bit data format of gfx90a/94x/950)
TE_HD __amdgpu_buffer_rsrc_t te_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, -1, 0x00020000);
}
constexpr int TE_AUX_NT = 2, TE_AUX_SC1 = 16;     // cache-policy bits of the buffer instructions on gfx940+
TE_HD uint32_t ldw(const uint32_t* base, uint32_t o, size_t word_off, int mem = MEM_STREAM) {
    const __amdgpu_buffer_rsrc_t r = te_rsrc(base);
    if (mem == MEM_AGENT || mem == MEM_AFFINE) return __builtin_amdgcn_raw_buffer_load_b32(r, (int)o, (int)(uint32_t)(word_off * 4u), TE_AUX_SC1);
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)o, (int)(uint32_t)(word_off * 4u), TE_LD_NT ? TE_AUX_NT : 0);
}
TE_HD void stw(uint32_t* base, uint32_t o, size_t word_off, uint32_t v, int mem = MEM_STREAM) {
    const __amdgpu_buffer_rsrc_t r = te_rsrc(base);
    if (mem == MEM_AGENT) __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)o, (int)(uint32_t)(word_off * 4u), TE_AUX_SC1);
    else if (mem == MEM_AFFINE) __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)o, (int)(uint32_t)(word_off * 4u), 0);
    else __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)o, (int)(uint32_t)(word_off * 4u), TE_ST_NT ? TE_AUX_NT : 0);
}
#else
TE_HD uint32_t ldw(const uint32_t* base, uint32_t o, size_t word_off, int mem = MEM_STREAM) {
    const uint32_t* p = (const uint32_t*)((const char*)(base + word_off) + o);
    return mem != MEM_STREAM ? ld_agent(p) : ld_stream(p);
}
TE_HD void stw(uint32_t* base, uint32_t o, size_t word_off, uint32_t v, int mem = MEM_STREAM) {
    uint32_t* p = (uint32_t*)((char*)(base + word_off) + o);
    if (mem != MEM_STREAM) st_agent(p, v); else st_stream(p, v);
}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
TE_HD void add_word(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
TE_HD void add_word(uint32_t* p, uint32_t v) { *p += v; }
#endif
// a * b for a, b < 2^24 whose product fits 32 bits: a full-rate multiply on the GPU (a 32 x 32 one takes four passes)
#if defined(__HIP_DEVICE_COMPILE__)
TE_HD uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#else
TE_HD uint32_t mul24(uint32_t a, uint32_t b) { return a * b; }
#endif
TE_HD int imin(int a, int b) { return a < b ? a : b; }
TE_HD int imax(int a, int b) { return a > b ? a : b; }
TE_HD uint32_t f2u(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
TE_HD float u2f(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }

// ---------------------------------------------------------------- shapes
// SURVEY.md App. B (probed against the reference for all 28 states): rows y0..y3 as hex nibbles,
// bit x = column x, indexed [piece][current_rotation].  Restates the templates + spawn rotations of
// gamePlay.cpp:116-158 and the raw 3x3/4x4 rotation of pieces.cpp:5-53 as a table.
constexpr uint16_t SHAPE_ROWS[7][4] = {
    {0x2260, 0x0710, 0x3220, 0x4700}, {0x2230, 0x1700, 0x6220, 0x0740}, {0x2640, 0x0630, 0x1320, 0x6300},
    {0x2310, 0x3600, 0x4620, 0x0360}, {0x2222, 0x0f00, 0x4444, 0x00f0}, {0x0720, 0x2320, 0x2700, 0x2620},
    {0x0660, 0x0660, 0x0660, 0x0660},
};
// spawn rotations {3,1,3,1,1,2,0} (gamePlay.cpp:117) and piece-grid values {4,3,5,7,2,1,6}
// (gamePlay.cpp:125-139), packed so that a per-lane lookup is a shift instead of a memory read
TE_HD int spawn_rot(int kind) { return (int)((0x0977u >> (2 * (kind & 7))) & 3u); }
TE_HD int shape_value(int kind) { return (int)((0x06127534u >> (4 * (kind & 7))) & 15u); }

// shape word: colnibs[0:16) | first occupied column[16:18) | last occupied column[18:20)
constexpr uint32_t make_shape(int kind, int rot) {
    uint32_t rows = SHAPE_ROWS[kind][rot];
    uint32_t nibs = 0;
    for (int gy = 0; gy < 4; gy++) {
        uint32_t rown = (rows >> (4 * (3 - gy))) & 0xF;
        for (int gx = 0; gx < 4; gx++)
            if ((rown >> gx) & 1) nibs |= 1u << (4 * gx + gy);
    }
    int minc = 3, maxc = 0;
    for (int gx = 0; gx < 4; gx++)
        if ((nibs >> (4 * gx)) & 0xF) { if (gx < minc) minc = gx; if (gx > maxc) maxc = gx; }
    return nibs | ((uint32_t)minc << 16) | ((uint32_t)maxc << 18);
}
// hard-drop word of a shape (see drop_distance_bytes): byte gx = 0x40 - (lowest cell row of piece column gx + 1), or 0x7F
// for an empty piece column
constexpr uint32_t make_drop_word(uint32_t nibs) {
    uint32_t w = 0;
    for (int gx = 0; gx < 4; gx++) {
        uint32_t n = (nibs >> (4 * gx)) & 0xF;
        int bottom = -1;
        for (int gy = 0; gy < 4; gy++)
            if ((n >> gy) & 1) bottom = gy;
        w |= (uint32_t)(bottom < 0 ? 0x7F : 0x40 - (bottom + 1)) << (8 * gx);
    }
    return w;
}
constexpr int SHAPE_WORDS = 64;                     // [0,32) shape words, [32,64) hard-drop words, index (kind << 2) | rot
struct ShapeTable { uint32_t s[SHAPE_WORDS]; };
constexpr ShapeTable make_shape_table() {
    ShapeTable t{};
    for (int k = 0; k < 7; k++)
        for (int r = 0; r < 4; r++) {
            t.s[k * 4 + r] = make_shape(k, r);
            t.s[32 + k * 4 + r] = make_drop_word(t.s[k * 4 + r] & 0xFFFFu);
        }
    for (int r = 0; r < 4; r++) { t.s[28 + r] = 0; t.s[60 + r] = 0x7F7F7F7Fu; }   // kind 7 = "no piece" (gameField.cpp:147-151)
    return t;
}
constexpr ShapeTable SHAPES = make_shape_table();

// ---------------------------------------------------------------- context (read-only per launch)
struct Ctx {
    const uint32_t* shapes;          // 32 shape words (LDS on the GPU)
    const uint8_t* table;            // RNG table, one allocation: table[(chunk * 65536 + seed16) * 624 + r]
    const uint64_t* start;           // [65536] first_ok | piece(j) << 8 | piece(j+1) << 16 | group(j+2) << 32 (tetris_tables.h)
    const double* combo_pow;         // [256] pow(c, 1.4 + 0.01 c) from the host libm (Combo.cpp:41)
    uint32_t n_draws;                // draws available per seed = n_chunks * 624
    uint32_t margin;                 // ST_NEED_EXTEND when a draw counter comes this close to n_draws
    int H;
    uint32_t floor_bits;             // ~0u << H
    bool tint;                       // colour planes are tracked (compile-time constant in every kernel)
    bool queue;                      // garbage queue / incoming lines can exist (false in 1-player kernels: nothing ever sends)
};

// Capacity errors are confined to the game they happen in (the reference's queue and generators are unbounded: Garbage.h:27,
// randomizer.h:44-50): the board gets an error bit (W_MISC[28:30), tetris_record.fifo_overflow), the game's round is over
// (`done`), every other game of the batch goes on.  A reset clears the bits.
constexpr int ERR_FIFO = 1;      // more than FIFO_CAP garbage packets were pending: one was dropped
constexpr int ERR_STREAM = 2;    // the episode ran past the RNG tables (MAX_CHUNKS * 624 draws): the dealt pieces are wrong

// ---------------------------------------------------------------- one player-board in registers
struct Raw8 { uint32_t lo, hi; };   // 8 raw RNG-table bytes
struct Player {
    uint32_t col[NCOL];
    uint32_t tint[3][NCOL];          // colour planes (only touched when Ctx::tint)
    int kind, rot, x, y, next;
    int dead, lock_armed, reward;
    int inc_count, combo_count, line_count, qlen;
    int q_overflow;                  // per-board error bits (ERR_*): the game is over, the board's state is not the reference's
    int32_t time_ms, drop_delay, drop_time, speedup_time, lock_time;
    int32_t combo_start, combo_time, min_remaining;
    uint32_t combo_remaining;
    float incoming;
    uint32_t piece_draws, hole_draws;
    uint32_t pgroup;                 // dealt pieces of the aligned group of 8 draws holding draw `piece_draws`, nibble each
    uint32_t lines_sent, lines_cleared, lines_blocked, max_combo, lines_seen, garbage_cleared;
    int32_t qcount[FIFO_CAP];
    int32_t qdelay[FIFO_CAP];
    int q_loaded;                    // FIFO words were read from memory (qlen > 0 at load time)
    Raw8 pf_raw;                     // prefetched raw table bytes of the group that starts at draw piece_draws + 1 (valid while pf_ok)
    int pf_ok;
};

// table bytes a reset will need, fetched ahead of time (see prefetch_reset)
struct ResetPrefetch { uint32_t seed16; uint64_t word; int ok; };

template <int P>
struct Game {
    Player pl[P];
    uint32_t seed16;                 // low 16 bits of the seed (table row)
    int round_over, last_winner;
    uint32_t flags;                  // split mode (cross-device opponents): side[0] opp_dead[1] split[2]
    uint32_t episode;
    uint32_t steps;                  // G_STEPS (cumulative; loaded and stored by the built-in rollout only)
    uint32_t add_lines, add_sent;    // this launch's additions to G_LINES / G_SENT (memory is touched only when non-zero)
    uint32_t status;                 // te::Status bits raised while stepping this game
    uint32_t draw0, draw1;           // rollout kernels: the synthetic policy's words for the NEXT step (registers only)
};

// ---------------------------------------------------------------- load / store (SoA, coalesced)
// state[(w * P + p) * n + slot]; game words at gstate[w * n + slot]
// one player-board: word w of this board lives at s[w * ws]
TE_HD void load_player(const uint32_t* s, uint32_t o, size_t ws, Player& q, bool tint, bool queue = true, int mem = MEM_STREAM) {
    for (int c = 0; c < NCOL; c++) q.col[c] = ldw(s, o, (size_t)(W_COL0 + c) * ws, mem);
    if (tint)
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < NCOL; c++) q.tint[k][c] = ldw(s, o, (size_t)(W_TINT0 + 10 * k + c) * ws, mem);
    uint32_t w = ldw(s, o, (size_t)W_PIECE * ws, mem);
    q.kind = w & 7; q.rot = (w >> 3) & 3; q.x = (int)((w >> 5) & 15) - 4; q.y = (w >> 9) & 31;
    q.next = (w >> 14) & 7; q.dead = (w >> 17) & 1; q.lock_armed = (w >> 18) & 1; q.reward = (w >> 19) & 255;
    w = ldw(s, o, (size_t)W_MISC * ws, mem);
    q.inc_count = w & 255; q.combo_count = (w >> 8) & 255; q.line_count = (w >> 16) & 255;
    q.qlen = (w >> 24) & 15; q.q_overflow = (w >> 28) & 3;
    if (!queue) { q.inc_count = 0; q.qlen = 0; q.q_overflow = 0; }      // invariants of a game without opponents
    q.time_ms = (int32_t)ldw(s, o, (size_t)W_TIME * ws, mem);
    w = ldw(s, o, (size_t)W_DROPCOMBO * ws, mem);
    q.drop_delay = w & 0xFFFF; q.combo_remaining = w >> 16;
    q.drop_time = (int32_t)ldw(s, o, (size_t)W_DROP_TIME * ws, mem);
    q.speedup_time = (int32_t)ldw(s, o, (size_t)W_SPEEDUP_TIME * ws, mem);
    q.lock_time = (int32_t)ldw(s, o, (size_t)W_LOCK_TIME * ws, mem);
    q.combo_start = (int32_t)ldw(s, o, (size_t)W_COMBO_START * ws, mem);
    q.combo_time = (int32_t)ldw(s, o, (size_t)W_COMBO_TIME * ws, mem);
    // nobody can send garbage to a single player: incoming lines, hole draws and the queue timer never leave their
    // reset values (0, 0, 1000), so 1-player kernels neither load nor (for the two zeros) store these words
    q.incoming = queue ? u2f(ldw(s, o, (size_t)W_INCOMING * ws, mem)) : 0.0f;
    q.min_remaining = queue ? (int32_t)ldw(s, o, (size_t)W_MIN_REMAINING * ws, mem) : 1000;
    q.piece_draws = ldw(s, o, (size_t)W_PIECE_DRAWS * ws, mem);
    q.hole_draws = queue ? ldw(s, o, (size_t)W_HOLE_DRAWS * ws, mem) : 0u;
    q.pgroup = ldw(s, o, (size_t)W_PIECE_GROUP * ws, mem);
    w = ldw(s, o, (size_t)W_STATS0 * ws, mem); q.lines_sent = w & 0xFFFF; q.lines_cleared = w >> 16;
    w = ldw(s, o, (size_t)W_STATS1 * ws, mem); q.lines_blocked = w & 0xFFFF; q.max_combo = w >> 16;
    w = ldw(s, o, (size_t)W_STATS2 * ws, mem); q.lines_seen = w & 0xFFFF; q.garbage_cleared = w >> 16;
    q.q_loaded = queue && q.qlen > 0;
    q.pf_ok = 0; q.pf_raw.lo = 0; q.pf_raw.hi = 0;
    if (queue)
        for (int i = 0; i < FIFO_CAP; i++) { q.qcount[i] = 0; q.qdelay[i] = 0; }
    if (q.q_loaded) {
        for (int i = 0; i < FIFO_CAP / 2; i++) {
            uint32_t cw = ldw(s, o, (size_t)(W_FIFO_COUNT0 + i) * ws, mem);
            q.qcount[2 * i] = (int16_t)(cw & 0xFFFF);
            q.qcount[2 * i + 1] = (int16_t)(cw >> 16);
        }
        for (int i = 0; i < FIFO_CAP; i++) q.qdelay[i] = (int32_t)ldw(s, o, (size_t)(W_FIFO_DELAY0 + i) * ws, mem);
    }
}

// the per-game words; `counters`: also G_STEPS (the built-in rollout)
template <int P>
TE_HD void load_game_words(const Ref& gr, Game<P>& g, bool counters = false, int mem = MEM_STREAM) {
    uint32_t meta = ldw(gr.s, gr.o, (size_t)G_META * gr.ws, mem);
    g.seed16 = meta & 0xFFFFu;
    g.round_over = (meta >> 16) & 1;
    g.last_winner = (int)((meta >> 17) & 0xF) - 1;
    g.flags = (meta >> 21) & 7u;
    g.episode = ldw(gr.s, gr.o, (size_t)G_EPISODE * gr.ws, mem);
    g.steps = counters ? ldw(gr.s, gr.o, (size_t)G_STEPS * gr.ws, mem) : 0u;
    g.add_lines = 0; g.add_sent = 0;
    g.status = 0;
}

template <int P>
TE_HD void load_game(const Geo& geo_in, size_t slot, Game<P>& g, bool tint = false, bool queue = true, bool counters = false, int mem = MEM_STREAM,
                     bool uniform = false) {
    Geo geo = geo_in;
    geo.P = P;                           // compile-time stride factor for the hot loads
    load_game_words<P>(game_ref(geo, slot, uniform), g, counters, mem);
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        const Ref r = board_ref(geo, p, slot, uniform);
        load_player(r.s, r.o, r.ws, g.pl[p], tint, queue, mem);
    }
}

TE_HD void store_player(uint32_t* s, uint32_t o, size_t ws, const Player& q, bool tint, bool queue = true, int mem = MEM_STREAM) {
    for (int c = 0; c < NCOL; c++) stw(s, o, (size_t)(W_COL0 + c) * ws, q.col[c], mem);
    if (tint)
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < NCOL; c++) stw(s, o, (size_t)(W_TINT0 + 10 * k + c) * ws, q.tint[k][c], mem);
    stw(s, o, (size_t)W_PIECE * ws, (uint32_t)q.kind | ((uint32_t)q.rot << 3) | ((uint32_t)(q.x + 4) << 5) | ((uint32_t)q.y << 9) |
                              ((uint32_t)q.next << 14) | ((uint32_t)q.dead << 17) | ((uint32_t)q.lock_armed << 18) |
                              ((uint32_t)(q.reward & 255) << 19), mem);
    stw(s, o, (size_t)W_MISC * ws, (uint32_t)(q.inc_count & 255) | ((uint32_t)(q.combo_count & 255) << 8) |
                             ((uint32_t)(q.line_count & 255) << 16) | ((uint32_t)q.qlen << 24) | ((uint32_t)q.q_overflow << 28), mem);
    stw(s, o, (size_t)W_TIME * ws, (uint32_t)q.time_ms, mem);
    stw(s, o, (size_t)W_DROPCOMBO * ws, ((uint32_t)q.drop_delay & 0xFFFF) | (q.combo_remaining << 16), mem);
    stw(s, o, (size_t)W_DROP_TIME * ws, (uint32_t)q.drop_time, mem);
    stw(s, o, (size_t)W_SPEEDUP_TIME * ws, (uint32_t)q.speedup_time, mem);
    stw(s, o, (size_t)W_LOCK_TIME * ws, (uint32_t)q.lock_time, mem);
    stw(s, o, (size_t)W_COMBO_START * ws, (uint32_t)q.combo_start, mem);
    stw(s, o, (size_t)W_COMBO_TIME * ws, (uint32_t)q.combo_time, mem);
    if (queue) stw(s, o, (size_t)W_INCOMING * ws, f2u(q.incoming), mem);
    stw(s, o, (size_t)W_MIN_REMAINING * ws, (uint32_t)q.min_remaining, mem);
    stw(s, o, (size_t)W_PIECE_DRAWS * ws, q.piece_draws, mem);
    if (queue) stw(s, o, (size_t)W_HOLE_DRAWS * ws, q.hole_draws, mem);
    stw(s, o, (size_t)W_PIECE_GROUP * ws, q.pgroup, mem);
    stw(s, o, (size_t)W_STATS0 * ws, (q.lines_sent & 0xFFFF) | (q.lines_cleared << 16), mem);
    stw(s, o, (size_t)W_STATS1 * ws, (q.lines_blocked & 0xFFFF) | (q.max_combo << 16), mem);
    stw(s, o, (size_t)W_STATS2 * ws, (q.lines_seen & 0xFFFF) | (q.garbage_cleared << 16), mem);
    if (queue && (q.q_loaded || q.qlen > 0)) {
        for (int i = 0; i < FIFO_CAP / 2; i++)
            stw(s, o, (size_t)(W_FIFO_COUNT0 + i) * ws, ((uint32_t)q.qcount[2 * i] & 0xFFFF) | ((uint32_t)q.qcount[2 * i + 1] << 16), mem);
        for (int i = 0; i < FIFO_CAP; i++) stw(s, o, (size_t)(W_FIFO_DELAY0 + i) * ws, (uint32_t)q.qdelay[i], mem);
    }
}

template <int P>
TE_HD void store_game_words(const Ref& gr, const Game<P>& g, bool counters = false, int mem = MEM_STREAM) {
    stw(gr.s, gr.o, (size_t)G_META * gr.ws, g.seed16 | ((uint32_t)g.round_over << 16) | ((uint32_t)(g.last_winner + 1) << 17) | (g.flags << 21), mem);
    stw(gr.s, gr.o, (size_t)G_EPISODE * gr.ws, g.episode, mem);
    if (counters) stw(gr.s, gr.o, (size_t)G_STEPS * gr.ws, g.steps, mem);
    // lines are cleared / sent in a few steps per thousand under a random policy: these two words are updated only then, by
    // a fire-and-forget atomic add (no return value: the wave does not wait for the memory round trip at the end of its
    // step — a dependent load + store there made the slowest wave, and with it every launch, ~0.8 us longer)
    if (g.add_lines) add_word(&word_at(gr, G_LINES), g.add_lines);
    if (g.add_sent) add_word(&word_at(gr, G_SENT), g.add_sent);
}

template <int P>
TE_HD void store_game(const Geo& geo_in, size_t slot, const Game<P>& g, bool tint = false, bool queue = true, bool counters = false, int mem = MEM_STREAM,
                      bool uniform = false) {
    Geo geo = geo_in;
    geo.P = P;
    store_game_words<P>(game_ref(geo, slot, uniform), g, counters, mem);
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        const Ref r = board_ref(geo, p, slot, uniform);
        store_player(r.s, r.o, r.ws, g.pl[p], tint, queue, mem);
    }
}

// ---------------------------------------------------------------- board primitives

TE_HD uint32_t shape_of(const Ctx& cx, int kind, int rot) { return cx.shapes[(kind << 2) | rot]; }

// band window B(y): nibble c+2 = rows y..y+3 of column c (floor below H), wall nibbles elsewhere.
TE_HD uint64_t band_window(const Ctx& cx, const Player& q, int y) {
    uint32_t lo = 0xFFu, hi = 0xFFFF0000u;
    for (int c = 0; c < 6; c++) lo |= ((uint32_t)((int32_t)(q.col[c] | cx.floor_bits) >> y) & 0xFu) << (4 * c + 8);
    for (int c = 6; c < NCOL; c++) hi |= ((uint32_t)((int32_t)(q.col[c] | cx.floor_bits) >> y) & 0xFu) << (4 * (c - 6));
    return ((uint64_t)hi << 32) | lo;
}

// gameField.cpp:10-20 BasicField::possible, against a prepared band window
TE_HD bool fits_band(uint64_t band, uint32_t shape, int x) {
    unsigned xs = (unsigned)(x + 2);
    if (xs > 12u) return false;          // x < -2 or x > 10: some occupied piece column is outside the board
    return (((uint64_t)(shape & 0xFFFFu) << (4 * xs)) & band) == 0;
}

TE_HD bool fits_at(const Ctx& cx, const Player& q, uint32_t shape, int x, int y) {
    return fits_band(band_window(cx, q, y), shape, x);
}

// The same test for a piece in the spawn column (x = 3, where every new piece appears and where it still is when the tick of
// the same step runs): only board columns 3..6 can collide — four ANDs instead of a band window over ten columns.
TE_HD bool fits_spawn_col(const Ctx& cx, const Player& q, uint32_t shape, int y) {
    constexpr int X0 = (NCOL - 4) / 2;
    uint32_t hit = 0;
    TE_UNROLL
    for (int gx = 0; gx < 4; gx++) hit |= (q.col[X0 + gx] | cx.floor_bits) & (((shape >> (4 * gx)) & 0xFu) << y);
    return hit == 0;
}

// dynamic column read without dynamic register indexing
TE_HD uint32_t col_at(const Player& q, int i) {
    uint32_t a0 = (i & 1) ? q.col[1] : q.col[0];
    uint32_t a1 = (i & 1) ? q.col[3] : q.col[2];
    uint32_t a2 = (i & 1) ? q.col[5] : q.col[4];
    uint32_t a3 = (i & 1) ? q.col[7] : q.col[6];
    uint32_t a4 = (i & 1) ? q.col[9] : q.col[8];
    uint32_t b0 = (i & 2) ? a1 : a0;
    uint32_t b1 = (i & 2) ? a3 : a2;
    uint32_t d0 = (i & 4) ? b1 : b0;
    return (i & 8) ? a4 : d0;
}

// gameField.cpp:49-53 BasicField::hd — closed form: per occupied piece column, the first occupied
// square (or the floor) strictly below the column's top cell bounds how far its bottom cell can go.
// Equals the step-by-step loop also when the start position itself overlaps (distance 0).
TE_HD int drop_distance(const Ctx& cx, const Player& q, uint32_t shape) {
    int dist = 64;
    for (int gx = 0; gx < 4; gx++) {
        uint32_t nib = (shape >> (4 * gx)) & 0xFu;
        if (nib) {
            int top = q.y + ctz32(nib);
            int bottom = q.y + (31 - clz32(nib));
            uint32_t below = (col_at(q, q.x + gx) | cx.floor_bits) >> (top + 1);
            int first = below ? ctz32(below) + top + 1 : 32;
            dist = imin(dist, imax(0, first - bottom - 1));
        }
    }
    return dist == 64 ? 0 : dist;
}

// gameField.cpp:105-110 addPiece: the four squares become occupied and take the piece's tile value (kind + 1)
// The same distance for a piece that FITS at (x, y), from byte arithmetic: depth[c] = rows from y down to the first
// occupied square (or the floor) of board column c, for the 10 columns held in registers; the four depths under the piece
// are four consecutive bytes of the 16-byte strip [wall wall c0..c9 wall wall wall wall] (one v_alignbyte after a word
// select); adding the shape's hard-drop word (make_drop_word) gives 0x40 + distance per occupied piece column and >= 0x7F
// for empty ones.  A byte below 0x40 means a square above that piece column's top cell inside the 4x4 box (an overhang
// entered by a kick): those lanes take the exact routine.
TE_HD int drop_distance_bytes(const Ctx& cx, const Player& q, uint32_t shape, uint32_t drop_word) {
    uint32_t d[NCOL];
    for (int c = 0; c < NCOL; c++) d[c] = (uint32_t)ctz32((q.col[c] | cx.floor_bits) >> q.y);     // y <= H: never zero
    const uint32_t w0 = (d[0] << 16) | (d[1] << 24);
    const uint32_t w1 = d[2] | (d[3] << 8) | (d[4] << 16) | (d[5] << 24);
    const uint32_t w2 = d[6] | (d[7] << 8) | (d[8] << 16) | (d[9] << 24);
    const unsigned xs = (unsigned)(q.x + 2);           // 0..12 for a piece that fits
    const bool hi2 = (xs & 8u) != 0, hi1 = (xs & 4u) != 0;
    const uint32_t lo = hi2 ? (hi1 ? 0u : w2) : (hi1 ? w1 : w0);
    const uint32_t hi = hi2 ? 0u : (hi1 ? w2 : w1);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t under = __builtin_amdgcn_alignbyte(hi, lo, xs & 3u);
#else
    const uint32_t under = (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (xs & 3u)));
#endif
    const uint32_t sum = under + drop_word;            // bytes <= 32 + 0x7F: no carries between bytes
    const uint32_t b0 = sum & 0xFFu, b1 = (sum >> 8) & 0xFFu, b2 = (sum >> 16) & 0xFFu, b3 = sum >> 24;
    const uint32_t m01 = b0 < b1 ? b0 : b1, m23 = b2 < b3 ? b2 : b3;
    const uint32_t m = m01 < m23 ? m01 : m23;
    if (m - 0x40u > 0x3Eu) { TE_COUNT(PC_DROP_EXACT); return drop_distance(cx, q, shape); }      // overhang (m < 0x40) or no piece at all (m >= 0x7F)
    return (int)(m - 0x40u);
}

TE_HD void stamp(const Ctx& cx, Player& q, uint32_t shape) {
    unsigned xs = (unsigned)(q.x + 2);
    if (xs > 12u) return;
    uint64_t placed = (uint64_t)(shape & 0xFFFFu) << (4 * xs);
    for (int c = 0; c < NCOL; c++) {
        const uint32_t cells = ((uint32_t)(placed >> (4 * c + 8)) & 0xFu) << q.y;
        q.col[c] |= cells;
        if (cx.tint)
            for (int k = 0; k < 3; k++) q.tint[k][c] = (q.tint[k][c] & ~cells) | (((q.kind >> k) & 1) ? cells : 0u);
    }
}

// gameField.cpp:120-145 clearlines (+ removeline :112-118): rows >= piece.posY that are full are
// removed bottom-up, everything above shifts down, and a row that shifts into the scanned range is
// examined too — i.e. "remove the lowest full row in range" until none is left.
TE_HD int clear_rows(const Ctx& cx, Player& q) {
    uint32_t range = (~0u << q.y) & ~cx.floor_bits;
    int cleared = 0;
    for (;;) {
        uint32_t full = range;
        for (int c = 0; c < NCOL; c++) full &= q.col[c];
        if (!full) break;
        int r = 31 - clz32(full);
        uint32_t above = (1u << r) - 1u;            // rows 0..r-1
        uint32_t keep = ~((above << 1) | 1u);       // rows r+1..
        if (cx.tint) {
            uint32_t garbage = 0;                   // squares holding an 8 (value - 1 = 0b111)
            for (int c = 0; c < NCOL; c++) garbage |= q.col[c] & q.tint[0][c] & q.tint[1][c] & q.tint[2][c];
            if ((garbage >> r) & 1u) q.garbage_cleared = (q.garbage_cleared + 1u) & 0xFFFFu;       // gameField.cpp:128-129,140-141
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < NCOL; c++) q.tint[k][c] = (q.tint[k][c] & keep) | ((q.tint[k][c] & above) << 1);
        }
        for (int c = 0; c < NCOL; c++) q.col[c] = (q.col[c] & keep) | ((q.col[c] & above) << 1);
        cleared++;
    }
    return cleared;
}

// ---------------------------------------------------------------- RNG table reads
TE_HD uint32_t table_byte(const Ctx& cx, uint32_t seed16, uint32_t draw, uint32_t& status) {
    if (draw + cx.margin >= cx.n_draws) {
        status |= ST_NEED_EXTEND;
        if (draw >= cx.n_draws) { status |= ST_STREAM_EXHAUSTED; draw = cx.n_draws - 1; }
    }
    uint32_t chunk = draw / (uint32_t)CHUNK;
    uint32_t r = draw - mul24(chunk, (uint32_t)CHUNK);
    return cx.table[mul24((chunk << 16) + seed16, (uint32_t)CHUNK) + r];       // < MAX_CHUNKS * 65536 * 624 < 2^32
}

// raw 8 table bytes of the aligned group of draws that holds `draw` (draw % 8 == 0 expected).  Loading and packing
// are separate on purpose: a prefetch must not touch the loaded registers, or the wave waits for the load on the spot.
TE_HD Raw8 table_group_raw(const Ctx& cx, uint32_t seed16, uint32_t draw, uint32_t& status) {
    if (draw + 8u + cx.margin >= cx.n_draws) {
        status |= ST_NEED_EXTEND;
        if (draw + 8u > cx.n_draws) { status |= ST_STREAM_EXHAUSTED; draw = cx.n_draws - 8u; }
    }
    uint32_t chunk = draw / (uint32_t)CHUNK;
    uint32_t r = draw - mul24(chunk, (uint32_t)CHUNK);                // multiple of 8; 624 = 8 * 78: never straddles
    const uint8_t* p = cx.table + (mul24((chunk << 16) + seed16, (uint32_t)CHUNK) + r);       // (a 32-bit index: < 2^32)
    Raw8 v;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint2 w = *reinterpret_cast<const uint2*>(p);               // 8-byte aligned
    v.lo = w.x; v.hi = w.y;
#else
    v.lo = 0; v.hi = 0;
    for (int k = 0; k < 4; k++) { v.lo |= (uint32_t)p[k] << (8 * k); v.hi |= (uint32_t)p[4 + k] << (8 * k); }
#endif
    return v;
}
// the 8 low nibbles (dealt pieces) of 8 table bytes -> one word, nibble k = piece of draw 8g + k
TE_HD uint32_t pack_group(Raw8 v) {
    const uint32_t a = v.lo & 0x0F0F0F0Fu, b = v.hi & 0x0F0F0F0Fu;
    uint32_t lo = (a | (a >> 4)) & 0x00FF00FFu; lo = (lo | (lo >> 8)) & 0x0000FFFFu;
    uint32_t hi = (b | (b >> 4)) & 0x00FF00FFu; hi = (hi | (hi >> 8)) & 0x0000FFFFu;
    return lo | (hi << 16);
}

// A spawn that consumes the LAST piece of the current group needs the next group afterwards: issue that read at the
// top of the step (only 1 lane in 8 per step does), so its latency hides under the key interpreter.
TE_HD void prefetch_next(const Ctx& cx, Player& q, uint32_t seed16, uint32_t& status) {
    q.pf_ok = 0;
    if ((q.piece_draws & 7u) == 7u) {
        q.pf_raw = table_group_raw(cx, seed16, q.piece_draws + 1u, status);
        q.pf_ok = 1;
    }
}

// ---------------------------------------------------------------- garbage queue (Garbage.cpp)
TE_HD void q_pop_front(Player& q) {
    for (int i = 0; i + 1 < FIFO_CAP; i++) { q.qcount[i] = q.qcount[i + 1]; q.qdelay[i] = q.qdelay[i + 1]; }
    q.qcount[FIFO_CAP - 1] = 0; q.qdelay[FIFO_CAP - 1] = 0;
    q.qlen--;
}

// Garbage.cpp:22-24 add (initialDelay 1000)
TE_HD void q_add(Player& q, int amount, int32_t t, uint32_t& status) {
    if (q.qlen >= FIFO_CAP) { q.q_overflow |= ERR_FIFO; status |= ST_FIFO_OVERFLOW; return; }
    for (int i = 0; i < FIFO_CAP; i++)
        if (i == q.qlen) { q.qcount[i] = (int16_t)amount; q.qdelay[i] = t + 1000; }
    q.qlen++;
}

// Garbage.cpp:26-52 block (freezeDelay 450)
TE_HD int q_block(Player& q, int amount, int32_t t, bool freeze) {
    if (q.qlen == 0) return amount;
    int32_t head_delay = q.qdelay[0];
    int blocked = 0;
    while (amount && q.qlen) {
        // whole head packet at once when it is not larger than what is left to block
        int take = imin(amount, (int)q.qcount[0]);
        if (take < 1) take = 1;                       // count <= 0 never occurs; mirror the decrement loop
        q.qcount[0] -= take; amount -= take; blocked += take;
        if (q.qcount[0] == 0) q_pop_front(q);
    }
    q.lines_blocked = (q.lines_blocked + (uint32_t)blocked) & 0xFFFFu;
    if (q.qlen) {
        if (head_delay > q.qdelay[0]) q.qdelay[0] = head_delay;
        if (freeze) q.qdelay[0] = imin(q.qdelay[0] + 450, t + q.min_remaining + 450);
    } else
        q.min_remaining = 1000;
    return amount;
}

// Garbage.cpp:54-72 check (addDelay 450, Garbage.cpp:7)
TE_HD bool q_release(Player& q, int32_t t) {
    if (q.qlen == 0) return false;
    if (t > q.qdelay[0]) {
        int32_t next_delay = q.qdelay[0] + 450;
        if (--q.qcount[0] == 0) q_pop_front(q);
        if (q.qlen) {
            if (next_delay > q.qdelay[0]) q.qdelay[0] = next_delay;
            q.min_remaining = q.qdelay[0] - t;
        } else
            q.min_remaining = 1000;
        return true;
    }
    q.min_remaining = imin(q.min_remaining, q.qdelay[0] - t);
    return false;
}

// Garbage.cpp:9-14 count (uint16 sum)
TE_HD int q_total(const Player& q) {
    uint32_t total = 0;
    for (int i = 0; i < FIFO_CAP; i++)
        if (i < q.qlen) total += (uint32_t)q.qcount[i];
    return (int)(total & 0xFFFFu);
}

// ---------------------------------------------------------------- combo (Combo.cpp)

// Combo.cpp:15-30 increase: integer quotients accumulate in a float; the float sum is added to the
// int32 comboTime in float arithmetic and truncated back.
TE_HD void combo_gain(Player& q, int32_t t, int amount) {
    if (q.combo_count == 0) { q.combo_start = t; q.combo_time = 0; }
    q.combo_count = (q.combo_count + 1) & 255;
    float line_time = 0.0f;
    for (int i = 0; i < amount; i++) {
        q.line_count = (q.line_count + 1) & 255;
        line_time = line_time + (float)(1000 / imax(1, q.line_count));
    }
    float add = (float)(800 / imax(1, q.combo_count)) + line_time;
    q.combo_time = (int32_t)((float)q.combo_time + add);
    if ((uint32_t)q.combo_count > q.max_combo) q.max_combo = (uint32_t)q.combo_count;
}

// Combo.cpp:32-48 check
TE_HD int combo_expire(const Ctx& cx, Player& q, int32_t t) {
    int32_t left = q.combo_start + q.combo_time - t;
    q.combo_remaining = left < 0 ? 0u : ((uint32_t)left & 0xFFFFu);
    if (t > q.combo_start + q.combo_time && q.combo_count != 0) {
        float duration = 1.f + (float)t / 60000.f * 0.1f;
        double v = cx.combo_pow[q.combo_count] * (double)duration;
        int lines = (int)((uint32_t)(int64_t)v & 0xFFFFu);
        q.combo_count = 0;
        q.line_count = 0;
        return lines;
    }
    return 0;
}

// ---------------------------------------------------------------- player logic (gamePlay.cpp)

// gamePlay.cpp:71-88 makeNewPiece / copyPiece; true when the spawn collides (piece stamped, dead)
TE_HD bool spawn_next(const Ctx& cx, Player& q, uint32_t seed16, uint32_t& status) {
    q.kind = q.next;
    q.rot = spawn_rot(q.kind);
    q.x = (NCOL - 4) / 2;
    q.y = 0;
    if (q.piece_draws + cx.margin >= cx.n_draws) {
        status |= ST_NEED_EXTEND;
        if (q.piece_draws >= cx.n_draws) status |= ST_STREAM_EXHAUSTED;
    }
    q.next = (int)((q.pgroup >> (4u * (q.piece_draws & 7u))) & 7u);
    q.piece_draws++;
    if ((q.piece_draws & 7u) == 0u) {                       // crossed into the next aligned group of 8 draws
        q.pgroup = pack_group(q.pf_ok ? q.pf_raw : table_group_raw(cx, seed16, q.piece_draws, status));
        q.pf_ok = 0;
    }
    uint32_t shape = shape_of(cx, q.kind, q.rot);
    if (!fits_spawn_col(cx, q, shape, 0)) { stamp(cx, q, shape); TE_COUNT(PC_DEATH_SPAWN); return true; }
    return false;
}

// gamePlay.cpp:160-171 sendLines (+ Combo.cpp:50-52 noClear)
TE_HD int score_clears(const Ctx& cx, Player& q, int cleared) {
    q.lines_cleared = (q.lines_cleared + (uint32_t)cleared) & 0xFFFFu;
    if (cleared == 0) { q.combo_time -= 200; return 0; }
    int amount = cx.queue ? q_block(q, cleared - 1, q.time_ms, true) : cleared - 1;
    q.lines_sent = (q.lines_sent + (uint32_t)amount) & 0xFFFFu;
    combo_gain(q, q.time_ms, cleared);
    return amount;
}

// gamePlay.cpp:48-52 hd_make (+ DropDelay.cpp:23-26 reset)
TE_HD void lock_piece(const Ctx& cx, Player& q) {
    uint32_t shape = shape_of(cx, q.kind, q.rot);
    q.y += drop_distance(cx, q, shape);
    stamp(cx, q, shape);
    q.drop_time = q.time_ms;
    q.lock_armed = 0;
}

// gamePlay.cpp:54-59 hd_finish; -1 = died
TE_HD int settle(const Ctx& cx, Player& q, uint32_t seed16, uint32_t& status) {
#if defined(TE_ABLATE) && (TE_ABLATE & 8)
    return 0;                                        // diagnostic build: no clear / spawn
#endif
    int sent = score_clears(cx, q, clear_rows(cx, q));
    if (spawn_next(cx, q, seed16, status)) return -1;
    return sent;
}

// gamePlay.cpp:61-69 GamePlay::mDown (+ DropDelay.cpp:23-26 reset, :37-41 set)
// A piece in the spawn column takes the four-column test: that is the piece of nearly every tick (it follows the spawn of the
// same step; the exception is a player whose loop-1 pass was skipped because an earlier player died in it), so the band-window
// path usually runs for no lane of a wave.
TE_HD bool soft_drop(const Ctx& cx, Player& q) {
    const uint32_t shape = shape_of(cx, q.kind, q.rot);
    if (q.x == (NCOL - 4) / 2 ? fits_spawn_col(cx, q, shape, q.y + 1) : fits_at(cx, q, shape, q.x, q.y + 1)) {
        q.y++;
        q.drop_time = q.time_ms;
        q.lock_armed = 0;
        return true;
    }
    if (!q.lock_armed) q.lock_time = q.time_ms + 400;
    q.lock_armed = 1;
    return false;
}

// DropDelay.cpp:3-21 check
TE_HD bool gravity_due(Player& q, int32_t t) {
    if (t - q.speedup_time > 3000) {
        if (q.drop_delay > 200) q.drop_delay -= 10;
        else if (q.drop_delay > 100) q.drop_delay -= 5;
        else if (q.drop_delay > 50) q.drop_delay -= 2;
        else if (q.drop_delay > 10) q.drop_delay -= 1;
        q.speedup_time = t;
    }
    if (t - q.drop_time > q.drop_delay) { q.drop_time = t; return true; }
    return false;
}

// gamePlay.cpp:179-204 pushGarbage / addGarbageLine; true when the piece dies.
// hole == 10 (float uniform rounding to 1.0) leaves the row without a hole.
TE_HD bool push_garbage(const Ctx& cx, Player& q, uint32_t seed16, uint32_t& status) {
    int hole = (int)(table_byte(cx, seed16, q.hole_draws, status) >> 4);
    q.hole_draws++;
    uint32_t bottom = 1u << (cx.H - 1);
    for (int c = 0; c < NCOL; c++) q.col[c] = (q.col[c] >> 1) | (c == hole ? 0u : bottom);
    if (cx.tint)
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < NCOL; c++) q.tint[k][c] = (q.tint[k][c] >> 1) | (c == hole ? 0u : bottom);
    TE_COUNT(PC_GARBAGE_ROW);
    if (q.y > 0) q.y--;
    if (!fits_at(cx, q, shape_of(cx, q.kind, q.rot), q.x, q.y)) {
        if (q.y > 0) { q.y--; TE_COUNT(PC_GARBAGE_LIFT2); }
        else { TE_COUNT(PC_DEATH_GARBAGE); return true; }
    }
    return false;
}

// gamePlay.cpp:90-114 delayCheck
TE_HD int tick(const Ctx& cx, Player& q, int ms, uint32_t seed16, uint32_t& status) {
    q.time_ms += ms;
#if defined(TE_ABLATE) && (TE_ABLATE & 4)
    return 0;                                        // diagnostic build: no timers / garbage / combo
#endif
    if (gravity_due(q, q.time_ms)) soft_drop(cx, q);
    // Single exit on purpose: with an early `return settle(..)` the compiler kept the board in different registers on the two
    // paths and merged them with ~50 register moves that every wave executed every step.
    int sent = 0;
    bool finished = false;
    if (q.lock_armed && q.time_ms > q.lock_time && !soft_drop(cx, q)) {    // DropDelay.cpp:43-48
        TE_COUNT(PC_TIMER_LOCK);
        lock_piece(cx, q);                                                  // gamePlay.cpp:38-46 hd
        sent = settle(cx, q, seed16, status);                               // "return" of the reference: nothing else happens this tick
        finished = true;
    }
    if (!finished) {
        if (cx.queue) {
            int whole = 0;
            while (q.incoming >= 1.0f) { whole++; q.incoming = q.incoming - 1.f; }
            if (whole) q_add(q, whole, q.time_ms, status);
        }
        sent = combo_expire(cx, q, q.time_ms);
        if (sent) {
            if (cx.queue) sent = q_block(q, sent, q.time_ms, false);
            q.lines_sent = (q.lines_sent + (uint32_t)sent) & 0xFFFFu;
        }
        if (cx.queue && q_release(q, q.time_ms))
            if (push_garbage(cx, q, seed16, status)) sent = -1;
    }
    return sent;
}

// gamePlay.cpp:206-216 restartRound + :218-230 seed, through the RNG tables (tetris_tables.h).
// Not reset (as in the reference): reward, inc_count, combo_remaining.
TE_HD void restart_player(const Ctx& cx, Player& q, uint32_t seed16, uint32_t& status, const ResetPrefetch* pf = nullptr) {
    for (int c = 0; c < NCOL; c++) q.col[c] = 0;
    if (cx.tint)
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < NCOL; c++) q.tint[k][c] = 0;
    q.qlen = 0; q.q_overflow = 0; q.lines_blocked = 0; q.min_remaining = 1000;
    if (cx.queue)
        for (int i = 0; i < FIFO_CAP; i++) { q.qcount[i] = 0; q.qdelay[i] = 0; }
    q.combo_start = 0; q.combo_time = 0; q.max_combo = 0; q.combo_count = 0; q.line_count = 0;
    q.lines_sent = 0; q.lines_cleared = 0; q.garbage_cleared = 0;
    q.speedup_time = 0; q.drop_delay = 1000; q.drop_time = 0; q.lock_time = 0; q.lock_armed = 0;
    q.time_ms = 0; q.incoming = 0.0f; q.lines_seen = 0; q.dead = 0;
    (void)status;
    const uint64_t word = (pf && pf->ok && pf->seed16 == seed16) ? pf->word : cx.start[seed16];
    const uint32_t j = (uint32_t)word & 0xFFu, b0 = ((uint32_t)word >> 8) & 0xFFu, b1 = ((uint32_t)word >> 16) & 0xFFu;
    q.pgroup = (uint32_t)(word >> 32);
    q.kind = (int)(b0 & 7u);
    q.next = (int)(b1 & 7u);
    q.pf_ok = 0;
    q.rot = spawn_rot(q.kind);
    q.x = (NCOL - 4) / 2; q.y = 0;
    q.piece_draws = j + 2; q.hole_draws = 0;
}

// PythonHandle.cpp:49-71 reset + seed
template <int P>
TE_HD void reset_game(const Ctx& cx, Game<P>& g, uint32_t seed16, const ResetPrefetch* pf = nullptr) {
    g.round_over = 0;
    int winner = -1, alive = 0;
    TE_UNROLL
    for (int p = 0; p < P; p++)
        if (!g.pl[p].dead) { alive++; winner = p; }
    g.last_winner = winner;
    if (P == 1) g.last_winner = 0;
    if (alive > 1) g.last_winner = -1;
    g.seed16 = seed16 & 0xFFFFu;
    TE_UNROLL
    for (int p = 0; p < P; p++) restart_player(cx, g.pl[p], g.seed16, g.status, pf);
}

// The rollout knows the seed of a game's NEXT episode one step ahead: read that seed's start word (one
// independent 4-byte load from a 256 KB table that lives in L2) at the top of the step, so an auto-reset at the
// bottom finds it in a register.  No load in the step depends on another load's result.
TE_HD void prefetch_reset(const Ctx& cx, uint32_t next_seed16, ResetPrefetch& pf) {
    pf.seed16 = next_seed16 & 0xFFFFu;
    pf.word = cx.start[pf.seed16];
    pf.ok = 1;
}

// PythonHandle.cpp:5-25 init: fresh GamePlay objects (nextpiece 0, reward 0, ...), restartRound, seed
template <int P>
TE_HD void init_game(const Ctx& cx, Game<P>& g, uint32_t seed16) {
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        Player& q = g.pl[p];
        q.reward = 0; q.inc_count = 0; q.combo_remaining = 0; q.dead = 0; q.next = 0; q.kind = 7; q.rot = 0;
        q.q_loaded = 1;
    }
    g.episode = 0; g.steps = 0; g.add_lines = 0; g.add_sent = 0; g.status = 0; g.flags = 0;
    reset_game(cx, g, seed16);
    g.last_winner = -1;
}

// ---------------------------------------------------------------- key interpreter

// gameField.cpp:55-103 rcw / rccw / r180 with the 7-offset kick test; turn = +1, +3 (ccw), +2.
// `b0` must be band_window(q.y) on entry and is kept equal to band_window(q.y) on exit, so a run of
// rotations and sideways moves at one height shares a single window.  `b1` caches band_window(q.y + 1)
// (valid while b1_ok).
TE_HD bool rotate_shape_band(const Ctx& cx, Player& q, int nr, uint32_t shape, uint64_t& b0, uint64_t& b1, bool& b1_ok) {
    if (fits_band(b0, shape, q.x)) { q.rot = nr; return true; }
    if (!b1_ok) { b1 = band_window(cx, q, q.y + 1); b1_ok = true; }
    // (dx,dy) in the reference's order: (0,+1) (-1,0) (+1,0) (-1,+1) (+1,+1) (-2,0) (+2,0)
    int dx = 99, dy = 0;
    if (fits_band(b1, shape, q.x)) { dx = 0; dy = 1; }
    else if (fits_band(b0, shape, q.x - 1)) { dx = -1; }
    else if (fits_band(b0, shape, q.x + 1)) { dx = 1; }
    else if (fits_band(b1, shape, q.x - 1)) { dx = -1; dy = 1; }
    else if (fits_band(b1, shape, q.x + 1)) { dx = 1; dy = 1; }
    else if (fits_band(b0, shape, q.x - 2)) { dx = -2; }
    else if (fits_band(b0, shape, q.x + 2)) { dx = 2; }
    if (dx == 99) { TE_COUNT(PC_KEY_KICK_FAILED); return false; }
    TE_COUNT(PC_KEY_KICK);
    q.rot = nr; q.x += dx; q.y += dy;
    if (dy) { b0 = b1; b1_ok = false; }
    return true;
}

TE_HD bool rotate_piece_band(const Ctx& cx, Player& q, int turn, uint64_t& b0) {
    uint64_t b1 = 0;
    bool b1_ok = false;
    const int nr = (q.rot + turn) & 3;
    return rotate_shape_band(cx, q, nr, shape_of(cx, q.kind, nr), b0, b1, b1_ok);
}

TE_HD bool rotate_piece(const Ctx& cx, Player& q, int turn) {
    uint64_t b0 = band_window(cx, q, q.y);
    return rotate_piece_band(cx, q, turn, b0);
}

// PythonHandle.cpp:73-112 action_make
TE_HD void press_key(const Ctx& cx, Player& q, int key) {
    if (key >= 1 && key <= 4) {
        uint32_t shape = shape_of(cx, q.kind, q.rot);
        uint64_t band = band_window(cx, q, q.y);
        int dir = (key <= 2) ? -1 : 1;
        bool repeat = (key == 2 || key == 4);
        while (fits_band(band, shape, q.x + dir)) { q.x += dir; if (!repeat) break; }
    } else if (key == 5) {
        soft_drop(cx, q);
    } else if (key == 6) {
        while (soft_drop(cx, q)) {}
    } else if (key == 7) {
        lock_piece(cx, q);
    } else if (key == 8) {
        rotate_piece(cx, q, 1);
    } else if (key == 9) {
        rotate_piece(cx, q, 3);
    } else if (key == 10) {
        rotate_piece(cx, q, 2);
    }
}

// all four rotations of one piece kind in one 128-bit LDS read (table rows are 16-byte aligned)
struct Shapes4 { uint32_t r[4]; };
TE_HD Shapes4 shapes_of_kind(const Ctx& cx, int kind) {
    Shapes4 out;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 v = *reinterpret_cast<const uint4*>(cx.shapes + ((kind & 7) << 2));
    out.r[0] = v.x; out.r[1] = v.y; out.r[2] = v.z; out.r[3] = v.w;
#else
    for (int i = 0; i < 4; i++) out.r[i] = cx.shapes[((kind & 7) << 2) | i];
#endif
    return out;
}
TE_HD Shapes4 drop_words_of_kind(const Ctx& cx, int kind) {
    Shapes4 out;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 v = *reinterpret_cast<const uint4*>(cx.shapes + 32 + ((kind & 7) << 2));
    out.r[0] = v.x; out.r[1] = v.y; out.r[2] = v.z; out.r[3] = v.w;
#else
    for (int i = 0; i < 4; i++) out.r[i] = cx.shapes[32 + (((kind & 7) << 2) | i)];
#endif
    return out;
}
TE_HD uint32_t pick4(const Shapes4& s, int rot) {
    uint32_t lo = (rot & 1) ? s.r[1] : s.r[0];
    uint32_t hi = (rot & 1) ? s.r[3] : s.r[2];
    return (rot & 2) ? hi : lo;
}

// bit xs of the result = 1  <=>  the shape fits at x = xs - 2 (xs = 0..12) in this band window
// the same map from 32-bit tests: positions 0..4 against band nibbles 0..7, 5..8 against nibbles 4..11, 9..12 against 8..15
TE_HD uint32_t free_positions32(uint64_t band, uint32_t shape) {
    const uint32_t lo = (uint32_t)band, hi = (uint32_t)(band >> 32), mid = (lo >> 16) | (hi << 16);
    const uint32_t s0 = shape & 0xFFFFu, s1 = s0 << 4, s2 = s0 << 8, s3 = s0 << 12, s4 = s0 << 16;
    uint32_t free = 0;
    free |= (s0 & lo) == 0 ? 1u << 0 : 0u;
    free |= (s1 & lo) == 0 ? 1u << 1 : 0u;
    free |= (s2 & lo) == 0 ? 1u << 2 : 0u;
    free |= (s3 & lo) == 0 ? 1u << 3 : 0u;
    free |= (s4 & lo) == 0 ? 1u << 4 : 0u;
    free |= (s1 & mid) == 0 ? 1u << 5 : 0u;
    free |= (s2 & mid) == 0 ? 1u << 6 : 0u;
    free |= (s3 & mid) == 0 ? 1u << 7 : 0u;
    free |= (s4 & mid) == 0 ? 1u << 8 : 0u;
    free |= (s1 & hi) == 0 ? 1u << 9 : 0u;
    free |= (s2 & hi) == 0 ? 1u << 10 : 0u;
    free |= (s3 & hi) == 0 ? 1u << 11 : 0u;
    free |= (s4 & hi) == 0 ? 1u << 12 : 0u;
    return free;
}
TE_HD uint32_t free_positions(uint64_t band, uint32_t shape) {
    const uint64_t pc = (uint64_t)(shape & 0xFFFFu);
    uint32_t free = 0;
    for (int xs = 0; xs <= 12; xs++) free |= (((pc << (4 * xs)) & band) == 0 ? 1u : 0u) << xs;
    return free;
}

// sventon_utils.py:9-13: the key list [8]*r + [2] + [3]*t + [7] executed literally
// (PythonHandle.cpp:73-112), but without per-lane loops:
//  * rotations: when the next r raw rotations all fit in place, none of them kicks and the result is
//    rot + r; only lanes where some intermediate rotation collides walk the kick sequence.
//  * key 2 then t x key 3: with the set F of free x positions at this height, "left until blocked"
//    ends just right of the nearest blocked position on the left, and each of the t single steps right
//    succeeds until the first blocked position on the right (a blocked step stays blocked).
// 32 bits (8 nibbles) of a band window starting at nibble position p, -8 <= p <= 16; positions outside the band read
// as wall.  A shape (16 bits) shifted by 4k, k = 0..4, tested against it is the collision test at band position p + k:
// one funnel shift per window, then and + compare per test, and out-of-range positions collide with the walls by
// construction (every shape has a cell in its first three and in its last three columns).
TE_HD uint32_t band_win32(uint64_t band, int p) {
    const int off = 4 * p + 32;                      // bit offset into [~0, lo, hi, ~0]
    const uint32_t lo = (uint32_t)band, hi = (uint32_t)(band >> 32);
    const int k = off >> 5;
    const uint32_t a = k == 0 ? ~0u : (k == 1 ? lo : (k == 2 ? hi : ~0u));
    const uint32_t b = k == 0 ? lo : (k == 1 ? hi : ~0u);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(b, a, (uint32_t)off & 31u);
#else
    const int sh = off & 31;
    return sh ? (a >> sh) | (b << (32 - sh)) : a;
#endif
}

// The ten columns seen from row y down, floor included: v[c] = (col[c] | floor) >> y.  Everything play_rt asks of the board at one
// height — the band window of row y, the band window of row y + 1 (kicks), the free depth under the piece (hard drop) — is a
// function of these ten words, so they are made once (20 instructions) instead of once per question.
struct RowsFrom { uint32_t v[NCOL]; };
TE_HD RowsFrom rows_from(const Ctx& cx, const Player& q, int y) {
    RowsFrom r;
    for (int c = 0; c < NCOL; c++) r.v[c] = (uint32_t)((int32_t)(q.col[c] | cx.floor_bits) >> y);
    return r;
}
TE_HD uint64_t band_of_rows(const RowsFrom& r) {           // = band_window(cx, q, y)
    uint32_t lo = 0xFFu, hi = 0xFFFF0000u;
    for (int c = 0; c < 6; c++) lo |= (r.v[c] & 0xFu) << (4 * c + 8);
    for (int c = 6; c < NCOL; c++) hi |= (r.v[c] & 0xFu) << (4 * (c - 6));
    return ((uint64_t)hi << 32) | lo;
}
// band_window(cx, q, y + 1) from the window of row y: every column nibble moves down one bit and takes row y + 4 in at the top
TE_HD uint64_t band_one_down(uint64_t band, const RowsFrom& r) {
    uint32_t lo = (uint32_t)band, hi = (uint32_t)(band >> 32);
    lo = ((lo >> 1) & 0x77777777u) | 0xFFu;              // (the mask also drops the bit that crosses in from the next nibble)
    hi = ((hi >> 1) & 0x77777777u) | 0xFFFF0000u;
    for (int c = 0; c < 6; c++) lo |= (r.v[c] & 0x10u) << (4 * c + 7);
    hi |= (r.v[6] & 0x10u) >> 1;
    for (int c = 7; c < NCOL; c++) hi |= (r.v[c] & 0x10u) << (4 * (c - 6) - 1);
    return ((uint64_t)hi << 32) | lo;
}
// drop_distance_bytes with the depths taken from the rows (v = the columns from the piece's row down): same result
TE_HD int drop_distance_rows(const Ctx& cx, const Player& q, const RowsFrom& r, uint32_t shape, uint32_t drop_word) {
    uint32_t d[NCOL];
    for (int c = 0; c < NCOL; c++) d[c] = (uint32_t)ctz32(r.v[c]);     // y <= H: never zero
    const uint32_t w0 = (d[0] << 16) | (d[1] << 24);
    const uint32_t w1 = d[2] | (d[3] << 8) | (d[4] << 16) | (d[5] << 24);
    const uint32_t w2 = d[6] | (d[7] << 8) | (d[8] << 16) | (d[9] << 24);
    const unsigned xs = (unsigned)(q.x + 2);           // 0..12 for a piece that fits
    const bool hi2 = (xs & 8u) != 0, hi1 = (xs & 4u) != 0;
    const uint32_t lo = hi2 ? (hi1 ? 0u : w2) : (hi1 ? w1 : w0);
    const uint32_t hi = hi2 ? 0u : (hi1 ? w2 : w1);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t under = __builtin_amdgcn_alignbyte(hi, lo, xs & 3u);
#else
    const uint32_t under = (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (xs & 3u)));
#endif
    const uint32_t sum = under + drop_word;            // bytes <= 32 + 0x7F: no carries between bytes
    const uint32_t b0 = sum & 0xFFu, b1 = (sum >> 8) & 0xFFu, b2 = (sum >> 16) & 0xFFu, b3 = sum >> 24;
    const uint32_t m01 = b0 < b1 ? b0 : b1, m23 = b2 < b3 ? b2 : b3;
    const uint32_t m = m01 < m23 ? m01 : m23;
    if (m - 0x40u > 0x3Eu) { TE_COUNT(PC_DROP_EXACT); return drop_distance(cx, q, shape); }      // overhang (m < 0x40) or no piece at all (m >= 0x7F)
    return (int)(m - 0x40u);
}

TE_HD void play_rt(const Ctx& cx, Player& q, int r, int t) {
#if defined(TE_ABLATE) && (TE_ABLATE & 2)
    lock_piece(cx, q); return;                       // diagnostic build: no rotations / slides
#endif
    const Shapes4 sh = shapes_of_kind(cx, q.kind);
    const Shapes4 dw = drop_words_of_kind(cx, q.kind);
    if (q.y != 0 || q.x != (NCOL - 4) / 2) TE_COUNT(PC_RT_OFF_SPAWN);
    RowsFrom rows = rows_from(cx, q, q.y);
    uint64_t band = band_of_rows(rows);
    uint32_t win = band_win32(band, q.x);            // band positions x .. x+4 (piece x - 2 .. x + 2), piece itself at k = 2
    const bool fit1 = (((pick4(sh, q.rot + 1) & 0xFFFFu) << 8) & win) == 0;
    const bool fit2 = (((pick4(sh, q.rot + 2) & 0xFFFFu) << 8) & win) == 0;
    const bool fit3 = (((pick4(sh, q.rot + 3) & 0xFFFFu) << 8) & win) == 0;
    const bool easy = (r < 1 || fit1) && (r < 2 || fit2) && (r < 3 || fit3);
    TE_STAMP(10);
    if (easy) {
        q.rot = (q.rot + r) & 3;
    } else {
        // Some rotation needs the kick table (gameField.cpp:55-103).  One loop iteration per KICK, not per rotation: the
        // rotations that fit in place around it are taken from no-kick tests at the current position, so a wave normally
        // runs this body once (it runs for the whole wave as soon as one lane needs it).
        int adv = fit1 ? (fit2 ? (fit3 ? 3 : 2) : 1) : 0;      // leading rotations that fit in place (adv < r here)
        q.rot = (q.rot + adv) & 3;
        int left = r - adv;
        int kicks = 0;
        (void)kicks;
        while (left > 0) {
            const int nr = (q.rot + 1) & 3;
            const uint32_t s1 = pick4(sh, nr) & 0xFFFFu;        // does not fit at (x, y): the 7 offsets in the reference's order
            const uint64_t b1 = band_one_down(band, rows);
            const uint32_t win1 = band_win32(b1, q.x);
            const uint32_t sA = s1, sB = s1 << 4, sC = s1 << 8, sD = s1 << 12, sE = s1 << 16;   // dx = -2 .. +2
            // all seven tests, then a priority select (no nested divergent branches)
            const bool k0 = (sC & win1) == 0, k1 = (sB & win) == 0, k2 = (sD & win) == 0;
            const bool k3 = (sB & win1) == 0, k4 = (sD & win1) == 0;
            const bool k5 = (sA & win) == 0, k6 = (sE & win) == 0;
            int dx = k6 ? 2 : 99, dy = 0;
            dx = k5 ? -2 : dx;
            if (k4) { dx = 1; dy = 1; }
            if (k3) { dx = -1; dy = 1; }
            if (k2) { dx = 1; dy = 0; }
            if (k1) { dx = -1; dy = 0; }
            if (k0) { dx = 0; dy = 1; }
            if (dx == 99) { TE_COUNT(PC_KICK_FAILED); break; }  // nothing fits: every further press fails the same way
            kicks++;
            TE_COUNT(kicks == 1 ? PC_KICK : (kicks == 2 ? PC_KICK_2ND : PC_KICK_3RD));
            if (dy) TE_COUNT(PC_KICK_DOWN);
            q.rot = nr; q.x += dx; q.y += dy;
            if (dy) {
                band = b1;
                for (int c = 0; c < NCOL; c++) rows.v[c] >>= 1;       // rows from y + 1 down (the floor keeps a bit below bit 31: H - y <= 31)
            }
            left--;
            if (left > 0) {                                     // rotations that fit in place at the new position
                win = band_win32(band, q.x);
                const bool g1 = (((pick4(sh, q.rot + 1) & 0xFFFFu) << 8) & win) == 0;
                const bool g2 = left > 1 && (((pick4(sh, q.rot + 2) & 0xFFFFu) << 8) & win) == 0;
                const int more = g1 ? (g2 ? 2 : 1) : 0;
                q.rot = (q.rot + more) & 3;
                left -= more;
            }
        }
    }
    TE_STAMP(11);
    const uint32_t shape = pick4(sh, q.rot);
    const uint32_t free = free_positions32(band, shape);
    const int xs0 = q.x + 2;
    const uint32_t blocked_left = ~free & ((1u << xs0) - 1u);
    int xs = blocked_left ? 32 - clz32(blocked_left) : 0;
    const uint32_t blocked_right = (~free >> (xs + 1)) | (1u << 13);
    xs += imin(t, ctz32(blocked_right));
    q.x = xs - 2;
    TE_STAMP(12);
    // gamePlay.cpp:48-52 hd_make
    q.y += drop_distance_rows(cx, q, rows, shape, pick4(dw, q.rot));
    TE_STAMP(13);
    stamp(cx, q, shape);
    q.drop_time = q.time_ms;
    q.lock_armed = 0;
}

// ---------------------------------------------------------------- get_actions (TestField.cpp:64-415)
// "place_block" enumeration for ONE (x, rotation) start: the plain drop, then every deeper resting place in
// that column range that can be reached by sliding under an overhang or by a rotation with wall kick, each
// with the key list that produces it.  The search runs backwards from the landing pose up to row 0,
// recording its path (255/254 = slid left/right, 24x = rotation, 253 = dropped to the floor, n < 240 = n rows),
// which is replayed reversed (TestField.cpp:3-35,127-164).  Poses may stick out above row 0 (y < 0).
constexpr int PROBE_PATH = 96;
struct Probe {
    Player q;                       // board + pose (kind, rot, x, y)
    int spawn;
    uint8_t path[PROBE_PATH]; int path_len;
    uint8_t best[PROBE_PATH]; int best_len, best_x, best_rot;
    uint8_t* keys; uint8_t* lens;   // this lane's slab: keys[max_lists][max_keys], lens[max_lists]
    int max_lists, max_keys, n_lists, overflow;
};
struct Pose { int rot, x, y; };
TE_HD Pose pose_get(const Probe& t) { Pose p = {t.q.rot, t.q.x, t.q.y}; return p; }
TE_HD void pose_set(Probe& t, Pose p) { t.q.rot = p.rot; t.q.x = p.x; t.q.y = p.y; }

// gameField.cpp:10-20 possible() for any y >= -3: cells above row 0 collide
TE_HD bool fits_any(const Ctx& cx, const Player& q, uint32_t shape, int x, int y) {
    if (y >= 0) return fits_at(cx, q, shape, x, y);
    int k = -y;
    if (k > 3) return (shape & 0xFFFFu) == 0;
    uint32_t low = ((1u << k) - 1u) * 0x1111u;
    if (shape & low) return false;
    return fits_at(cx, q, ((shape & 0xFFFFu) >> k) & ((0xFu >> k) * 0x1111u), x, 0);
}
TE_HD bool probe_fits(const Ctx& cx, const Probe& t) { return fits_any(cx, t.q, shape_of(cx, t.q.kind, t.q.rot), t.q.x, t.q.y); }
TE_HD bool probe_shift(const Ctx& cx, Probe& t, int dx) {
    if (fits_any(cx, t.q, shape_of(cx, t.q.kind, t.q.rot), t.q.x + dx, t.q.y)) { t.q.x += dx; return true; }
    return false;
}
TE_HD void probe_drop(const Ctx& cx, Probe& t) { t.q.y += drop_distance(cx, t.q, shape_of(cx, t.q.kind, t.q.rot)); }
TE_HD void path_push(Probe& t, int v) { if (t.path_len < PROBE_PATH) t.path[t.path_len++] = (uint8_t)v; else t.overflow = 1; }

TE_HD int convert_code(int m) {                                   // TestField.cpp:3-35
    return m == 255 ? 3 : m == 254 ? 1 : m == 253 ? 6 : m == 252 ? 5 : m == 241 ? 8 : m == 242 ? 10 : m == 243 ? 9 : m;
}
TE_HD void emit_key(Probe& t, int& n, int code) {
    if (t.n_lists < t.max_lists && n < t.max_keys) t.keys[t.n_lists * t.max_keys + n] = (uint8_t)convert_code(code);
    else t.overflow = 1;
    n++;
}
TE_HD void emit_end(Probe& t, int n) {
    if (t.n_lists < t.max_lists) t.lens[t.n_lists] = (uint8_t)imin(n, t.max_keys);
    t.n_lists++;
}
TE_HD void emit_rotation(Probe& t, int& n, int value) { value &= 3; if (value) emit_key(t, n, 240 + value); }   // :37-42
TE_HD void emit_start_moves(Probe& t, int& n, int x) {                                                             // :44-49
    const int mid = (NCOL - 4) / 2;
    for (int i = 0; i < x - mid; i++) emit_key(t, n, 255);
    for (int i = 0; i < mid - x; i++) emit_key(t, n, 254);
}

// TestField.cpp:392-410 moveUp
TE_HD int probe_move_up(const Ctx& cx, Probe& t) {
    t.q.y++;
    const bool landed = !probe_fits(cx, t);
    t.q.y--;
    int count = 0;
    do { t.q.y--; count++; } while (probe_fits(cx, t));
    count--;
    t.q.y++;
    if (count && landed) return 253;
    return count & 255;
}

TE_HD bool probe_commit(Probe& t) {                                 // TestField.cpp:166-172 setFinesseMove
    for (int i = 0; i < t.path_len; i++) t.best[i] = t.path[i];
    t.best_len = t.path_len; t.best_x = t.q.x; t.best_rot = t.q.rot;
    return true;
}

// TestField.cpp:202-238 tryLeft / tryRight
TE_HD bool probe_slide(const Ctx& cx, Probe& t, int dir, bool clear_first) {
    for (;;) {
        if (!probe_shift(cx, t, dir)) return false;
        if (clear_first) t.path_len = 0;
        clear_first = false;
        path_push(t, dir < 0 ? 255 : 254);
        int up = probe_move_up(cx, t);
        if (up) path_push(t, up);
        if (t.q.y == 0) return probe_commit(t);
    }
}

// TestField.cpp:240-259 tryUp
TE_HD bool probe_up(const Ctx& cx, Probe& t, int turn) {
    t.path_len = 0;
    int up = probe_move_up(cx, t);
    path_push(t, turn + 240);
    if (up) path_push(t, up);
    if (t.q.y == 0) return probe_commit(t);
    if (probe_slide(cx, t, -1, false)) return true;
    t.path_len = 0;
    path_push(t, turn + 240);
    if (up) path_push(t, up);
    return probe_slide(cx, t, +1, false);
}

// gameField.cpp:55-103 rcw/rccw/r180 with kicks, for poses that may stick out above row 0
TE_HD void probe_rotate(const Ctx& cx, Probe& t, int turn) {
    const int nr = (t.q.rot + turn) & 3;
    const uint32_t shape = shape_of(cx, t.q.kind, nr);
    const int x = t.q.x, y = t.q.y;
    int dx = 99, dy = 0;
    if (fits_any(cx, t.q, shape, x, y)) { dx = 0; }
    else if (fits_any(cx, t.q, shape, x, y + 1)) { dx = 0; dy = 1; }
    else if (fits_any(cx, t.q, shape, x - 1, y)) { dx = -1; }
    else if (fits_any(cx, t.q, shape, x + 1, y)) { dx = 1; }
    else if (fits_any(cx, t.q, shape, x - 1, y + 1)) { dx = -1; dy = 1; }
    else if (fits_any(cx, t.q, shape, x + 1, y + 1)) { dx = 1; dy = 1; }
    else if (fits_any(cx, t.q, shape, x - 2, y)) { dx = -2; }
    else if (fits_any(cx, t.q, shape, x + 2, y)) { dx = 2; }
    if (dx == 99) return;
    t.q.rot = nr; t.q.x = x + dx; t.q.y = y + dy;
}

// TestField.cpp:280-356 doWallKick: un-rotate (kick offsets mirrored in y) to a pose from which the forward
// rotation, kicks included, lands exactly on the target pose
TE_HD bool probe_wallkick(const Ctx& cx, Probe& t) {
    const Pose target = pose_get(t);
    int r = 0;
    bool found = false;
    for (r = 0; r < 4 && !found; r++) {
        if (r == target.rot) continue;
        t.q.rot = r;
        if (probe_fits(cx, t)) {
            bool ok = probe_up(cx, t, (target.rot - r) & 3);
            pose_set(t, target);
            return ok;
        }
        const int bx = t.q.x, by = t.q.y;
        for (int k = 0; k < 7 && !found; k++) {
            // (0,-1) (-1,0) (+1,0) (-1,-1) (+1,-1) (-2,0) (+2,0)
            const int ddx = (k == 0) ? 0 : (k == 1 || k == 3) ? -1 : (k == 2 || k == 4) ? 1 : (k == 5) ? -2 : 2;
            const int ddy = (k == 0 || k == 3 || k == 4) ? -1 : 0;
            t.q.x = bx + ddx; t.q.y = by + ddy;
            if (probe_fits(cx, t)) found = true;
        }
        if (found) break;
        t.q.x = bx; t.q.y = by;
    }
    if (!found) { pose_set(t, target); return false; }
    const int turn = (target.rot - r) & 3;
    const Pose from = pose_get(t);
    if (turn == 0) { pose_set(t, target); return false; }
    probe_rotate(cx, t, turn);
    if (t.q.x != target.x || t.q.y != target.y) { pose_set(t, target); return false; }
    pose_set(t, from);
    if (probe_up(cx, t, turn)) { pose_set(t, target); return true; }
    pose_set(t, from);
    t.path_len = 0;
    path_push(t, 240 + turn);
    if (probe_slide(cx, t, -1, false)) { pose_set(t, target); return true; }
    pose_set(t, from);
    if (t.path_len > 1) { t.path_len = 0; path_push(t, 240 + turn); }
    bool ok = probe_slide(cx, t, +1, false);
    pose_set(t, target);
    return ok;
}

// TestField.cpp:358-390 r180KeepPos
TE_HD void probe_flip_keep_pos(Probe& t) {
    t.q.rot = (t.q.rot + 2) & 3;
    const int k = t.q.kind, r = t.q.rot;
    if (k == 4 || k == 3) { if (r == 0) t.q.x++; else if (r == 1) t.q.y++; else if (r == 2) t.q.x--; else t.q.y--; }
    if (k == 2) { if (r == 0) t.q.x--; else if (r == 1) t.q.y--; else if (r == 2) t.q.x++; else t.q.y++; }
}

// TestField.cpp:261-278 reverseWallkick
TE_HD bool probe_reverse_kick(const Ctx& cx, Probe& t) {
    if (t.q.kind == 6) return false;
    if (probe_shift(cx, t, +1)) { probe_shift(cx, t, -1); return false; }
    if (probe_shift(cx, t, -1)) { probe_shift(cx, t, +1); return false; }
    const Pose here = pose_get(t);
    if (probe_wallkick(cx, t)) { pose_set(t, here); return true; }
    if (t.q.kind == 2 || t.q.kind == 3 || t.q.kind == 4) {
        probe_flip_keep_pos(t);
        bool ok = probe_wallkick(cx, t);
        pose_set(t, here);
        return ok;
    }
    pose_set(t, here);
    return false;
}

// TestField.cpp:189-200 finesseIsPossible
TE_HD bool probe_reachable(const Ctx& cx, Probe& t) {
    const Pose here = pose_get(t);
    if (probe_reverse_kick(cx, t)) { pose_set(t, here); return true; }
    pose_set(t, here);
    if (probe_slide(cx, t, -1, true)) { pose_set(t, here); return true; }
    bool ok = probe_slide(cx, t, +1, true);
    pose_set(t, here);
    return ok;
}

// TestField.cpp:113-125 findNextMove + :174-187 tryAllFinesseMoves + :127-164 useFinesseMove
TE_HD void probe_column(const Ctx& cx, Probe& t) {
    int n = 0;
    emit_rotation(t, n, t.q.rot - t.spawn);
    emit_start_moves(t, n, t.q.x);
    emit_key(t, n, 7);
    emit_end(t, n);
    probe_drop(cx, t);
    const Pose landed = pose_get(t);
    for (int y = landed.y + 2; y < cx.H - 1; y++) {
        pose_set(t, landed);
        t.q.y = y;
        if (probe_fits(cx, t)) {
            probe_drop(cx, t);
            y = t.q.y;
            if (probe_reachable(cx, t)) {
                n = 0;
                emit_rotation(t, n, t.best_rot - t.spawn);
                emit_start_moves(t, n, t.best_x);
                for (int i = t.best_len - 1; i >= 0; i--) {
                    int v = t.best[i];
                    if (v < 240) for (int k = 0; k < v; k++) emit_key(t, n, 252);
                    else emit_key(t, n, v);
                }
                emit_key(t, n, 7);
                emit_end(t, n);
            }
        }
    }
    pose_set(t, landed);
}

// ---------------------------------------------------------------- the two-phase step

// PythonHandle.cpp:124-136 distributeLines
template <int P>
TE_HD void share_lines(Game<P>& g, int sender, int amount) {
    if (P < 2) return;
    float each = (float)amount / (float)(P - 1);
    TE_UNROLL
    for (int p = 0; p < P; p++)
        if (p != sender) g.pl[p].incoming = g.pl[p].incoming + each;
}

// a capacity error of this step (status bits) or of an earlier one (board bits) ends the game's round: see ERR_*
template <int P>
TE_HD bool confine_errors(Game<P>& g) {
    int err = 0;
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        if (g.status & ST_STREAM_EXHAUSTED) g.pl[p].q_overflow |= ERR_STREAM;      // the seed, and with it the tables, is the game's
        err |= g.pl[p].q_overflow;
    }
    return err != 0;
}

// PythonHandle.cpp:149-188 finish_actions (+ action_finish :114-122); returns round_over
template <int P>
TE_HD int finish_game(const Ctx& cx, Game<P>& g, int ms) {
    if (g.round_over) return 1;
    bool stop = false;
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        Player& q = g.pl[p];
        if (!stop && !q.dead) {
            const int sent = settle(cx, q, g.seed16, g.status);
            if (sent == -1) { q.dead = 1; stop = true; }           // the reference breaks out of loop 1
            else if (sent) share_lines<P>(g, p, sent);
        }
    }
    TE_STAMP(6);
    int alive = 0;
    TE_UNROLL
    for (int p = 0; p < P; p++) {
        Player& q = g.pl[p];
        if (!q.dead) {
            const int sent = tick(cx, q, ms, g.seed16, g.status);
            if (sent == -1) q.dead = 1;
            else {
                if (sent) share_lines<P>(g, p, sent);
                alive++;
                q.reward = (int)((q.lines_cleared - q.lines_seen) & 0xFFu);
                q.lines_seen = q.lines_cleared;
                q.inc_count = cx.queue ? (q_total(q) & 255) : 0;
            }
        }
    }
    if (confine_errors<P>(g)) { g.round_over = 1; return 1; }
    if ((P > 1 && alive < 2) || !alive) { g.round_over = 1; return 1; }
    return 0;
}

// ---------------------------------------------------------------- split mode: opponents on different GPUs
// One batch holds ONE side (player index `side`) of N two-player games; the other side lives in another batch,
// normally on another GPU.  The only cross-board traffic of the engine is PythonHandle::distributeLines
// (PythonHandle.cpp:124-136) plus the dead flags of the winner logic, so a step is cut into three stages with an
// exchange of one 32-bit word per board after each (an RCCL all-gather in drl-tetris_amd/distributed.py):
//   A  key interpreter + loop 1 of finish_actions (clear, send/block, new piece; PythonHandle.cpp:151-158).
//      Player 1 runs it SPECULATIVELY: the reference skips player 1 when player 0 died in loop 1 (`break`), which
//      side 1 only learns from the exchange, so it keeps both its pre- and post-settle state.
//   B  loop 2 (delayCheck), player 0 first, then (after a second exchange) player 1 — player 0's combo lines reach
//      player 1's garbage queue in the same step (PythonHandle.cpp:160-180).
//   C  after the third exchange: player 1's tick lines reach player 0, winner / round_over (:182-187).
// Exchange word: sent[0:16) | DIED[16] | DEAD_NOW[17] | RAN[18].
constexpr uint32_t XW_DIED = 1u << 16, XW_DEAD_NOW = 1u << 17, XW_RAN = 1u << 18, XW_ERR = 1u << 19;
constexpr uint32_t SPLIT_SIDE = 1u, SPLIT_OPP_DEAD = 2u, SPLIT_ON = 4u;
TE_HD int xw_sent(uint32_t w) { return (int)(w & 0xFFFFu); }

// stage A after the key interpreter: loop-1 body for my player.  Returns my exchange word.
TE_HD uint32_t split_settle(const Ctx& cx, Game<1>& g) {
    Player& q = g.pl[0];
    if (g.round_over || q.dead) return 0u;
    int sent = settle(cx, q, g.seed16, g.status);
    if (sent == -1) { q.dead = 1; return XW_RAN | XW_DIED; }
    return XW_RAN | (uint32_t)(sent & 0xFFFF);
}

// stage B: my player's delayCheck.  `first_in` = lines that reach me before my tick.  Returns my word.
TE_HD uint32_t split_tick(const Ctx& cx, Game<1>& g, int ms, int lines_in) {
    Player& q = g.pl[0];
    if (g.round_over) return (q.dead ? XW_DEAD_NOW : 0u) | (q.q_overflow ? XW_ERR : 0u);
    if (lines_in > 0) q.incoming = q.incoming + (float)lines_in / 1.0f;       // amount / (P - 1), P = 2
    uint32_t w = 0;
    if (!q.dead) {
        int sent = tick(cx, q, ms, g.seed16, g.status);
        if (sent == -1) { q.dead = 1; w = XW_DIED; }
        else {
            w = (uint32_t)(sent & 0xFFFF);
            q.reward = (int)((q.lines_cleared - q.lines_seen) & 0xFFu);
            q.lines_seen = q.lines_cleared;
            q.inc_count = q_total(q) & 255;
        }
    }
    if (g.status & ST_STREAM_EXHAUSTED) q.q_overflow |= ERR_STREAM;
    return w | (q.dead ? XW_DEAD_NOW : 0u) | (q.q_overflow ? XW_ERR : 0u);
}

// stage C: lines that arrive after my tick, then PythonHandle.cpp:182-187 with the opponent's final dead flag
TE_HD int split_finish(Game<1>& g, int lines_in, bool opp_dead, bool opp_err = false) {
    Player& q = g.pl[0];
    if (g.round_over) return 1;
    if (lines_in > 0) q.incoming = q.incoming + (float)lines_in / 1.0f;
    g.flags = (g.flags & ~SPLIT_OPP_DEAD) | (opp_dead ? SPLIT_OPP_DEAD : 0u);
    int alive = (q.dead ? 0 : 1) + (opp_dead ? 0 : 1);
    if (alive < 2 || q.q_overflow || opp_err) { g.round_over = 1; return 1; }      // a capacity error on either side ends the round (ERR_*)
    return 0;
}

// PythonHandle.cpp:49-71 reset for one side of a split game: last_winner needs both dead flags
TE_HD void reset_split(const Ctx& cx, Game<1>& g, uint32_t seed16) {
    const int side = (int)(g.flags & SPLIT_SIDE);
    const bool me_alive = !g.pl[0].dead, opp_alive = !(g.flags & SPLIT_OPP_DEAD);
    int winner = -1;
    if (me_alive && !opp_alive) winner = side;
    if (opp_alive && !me_alive) winner = 1 - side;
    reset_game<1>(cx, g, seed16);
    g.last_winner = winner;                        // both alive: -1 (alive_count > 1); none alive: -1
    g.flags &= ~SPLIT_OPP_DEAD;
}

// seed16 = (12345 + 7919 i + 104729 e) mod 65536
// (only the low 16 bits of the sum count, so the low 16 bits of game and episode do: two full-rate 24-bit multiplies)
TE_HD uint32_t episode_seed(uint32_t game, uint32_t episode) { return (12345u + mul24(7919u, game & 0xFFFFu) + mul24(104729u & 0xFFFFu, episode & 0xFFFFu)) & 0xFFFFu; }

// Philox4x32-10 (Salmon et al., SC'11), key (k0, 0), counter (c0, c1, c2, 0): the synthetic policy
TE_HD void philox4x32_10(uint32_t k0, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t out[4]) {
    uint32_t k1 = 0u, c3 = 0u;
    TE_UNROLL                                        // straight-line: lets the scheduler run it under outstanding loads
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // namespace te
