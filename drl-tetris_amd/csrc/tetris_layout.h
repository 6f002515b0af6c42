// Device state layout of the MI355X batched Tetris environment (shared by host and device code).
//
// One "player-board" = NWORDS uint32 words, stored structure-of-arrays in HBM:
//     state[word][player][game]          (game is the fastest index)
// so that a wavefront whose 64 lanes hold 64 consecutive games reads/writes every word as one
// fully coalesced 256-byte access.  One extra array of NGWORDS words per game holds what the
// reference keeps per PythonHandle (round_over, last_winner) plus the RNG seed, which the
// reference gives to all players and both generators of a game alike (PythonHandle.cpp:68-71).
//
// What is NOT stored: the two std::mt19937 engines (2 x 2.5 KB per player in the reference,
// randomizer.h:44-50) and the 7 float piece weights (randomizer.h:42).  Both are pure functions
// of (seed16, number of draws so far) — see tetris_tables.h — so a board only carries its two
// draw counters.
#pragma once
#include <stdint.h>

namespace te {

constexpr int NCOL = 10;        // board width; the reference hard-codes 10 (gamePlay.cpp:202)
constexpr int MAX_H = 31;       // one uint32 per column; bit 31 is needed as floor sentinel
constexpr int FIFO_CAP = 8;     // pending garbage packets per board (reference: unbounded vector)
constexpr int CHUNK = 624;      // draws per RNG-table chunk = one MT19937 block
constexpr int MAX_CHUNKS = 64;  // 39 936 draws per episode

enum Word : int {
    W_COL0 = 0,            // 10 words: occupancy of column c, bit y = row y (row 0 = top)
    W_PIECE = 10,          // kind[0:3) rot[3:5) x+4[5:9) y[9:14) next[14:17) dead[17] lock_armed[18] reward[19:27)
    W_MISC = 11,           // inc_count[0:8) combo_count[8:16) line_count[16:24) fifo_len[24:28) fifo_overflow[28]
    W_TIME = 12,           // GamePlay.time_ms
    W_DROPCOMBO = 13,      // DropDelay.dropDelay[0:16) | ComboCounter.remaining[16:32)
    W_DROP_TIME = 14,      // DropDelay.dropDelayTime
    W_SPEEDUP_TIME = 15,   // DropDelay.increaseDropDelayTime
    W_LOCK_TIME = 16,      // DropDelay.lockdownTime
    W_COMBO_START = 17,    // ComboCounter.comboStart
    W_COMBO_TIME = 18,     // ComboCounter.comboTime
    W_INCOMING = 19,       // GamePlay.incoming_lines (float bits)
    W_MIN_REMAINING = 20,  // GarbageHandler.minRemaining
    W_PIECE_DRAWS = 21,    // outputs consumed from piece_gen
    W_HOLE_DRAWS = 22,     // outputs consumed from hole_gen
    W_STATS0 = 23,         // linesSent[0:16) | linesCleared[16:32)
    W_STATS1 = 24,         // linesBlocked[0:16) | maxCombo[16:32)
    W_STATS2 = 25,         // GamePlay.linesCleared ("seen")[0:16) | garbageCleared[16:32)
    W_PIECE_GROUP = 26,    // dealt pieces of draws 8*(piece_draws/8) .. +7, one nibble each (cache of the RNG table)
    W_FIFO_COUNT0 = 27,    // 4 words: Garbage.count, two int16 per word
    W_FIFO_DELAY0 = 31,    // 8 words: Garbage.delay
    NWORDS = 39,
    NWORDS_HOT = 27,       // words touched by every step; the FIFO words only when a queue exists
    // optional colour planes (batches created with TETRIS_FLAG_COLOURS): plane k, column c at W_TINT0 + 10 k + c holds
    // bit k of (cell value - 1) for every occupied square: tiles 1..7 (gamePlay.cpp:146) and 8 = garbage (gamePlay.cpp:202)
    W_TINT0 = 39,
    NWORDS_TINT = 69
};

enum GameWord : int {
    G_META = 0,     // seed16[0:16) round_over[16] last_winner+1[17:21) | split mode: side[21] opp_dead[22] split[23]
    G_EPISODE = 1,  // episodes finished by the built-in rollout / device-side auto-reset (SURVEY.md §8d seed schedule)
    G_STEPS = 2,    // env-steps executed on this game by the built-in rollout (cumulative; summed by k_totals)
    G_LINES = 3,    // lines cleared during built-in rollouts (cumulative; touched only when a line was cleared)
    G_SENT = 4,     // garbage lines sent during built-in rollouts (cumulative; touched only when lines were sent)
    NGWORDS = 5,
    NGWORDS_HOT = 3 // read and written by every rollout step
};

// status bits a lane collects while stepping its game; reported through the batch's flag words (below)
enum Status : uint32_t {
    ST_NEED_EXTEND = 1u,        // some board is within `margin` draws of the end of the RNG tables
    ST_STREAM_EXHAUSTED = 2u,   // a draw index ran past the tables (results invalid)
    ST_FIFO_OVERFLOW = 4u,      // more than FIFO_CAP pending garbage packets (results invalid)
    ST_BAD_ARGUMENT = 8u,
};

// Flag words of a batch: pinned host memory that the GPU writes with plain stores (rare) and the host reads without
// enqueuing anything, so the asynchronous entry points (tetris_step_rt_dev, the rollout launches) can service
// "extend the RNG tables" requests without a copy or a stream drain per launch.
enum Flag : int {
    F_EXTEND = 0,      // n_draws as seen by a kernel in which a board came within `margin` of the table end (0 = no request)
    F_EXHAUSTED = 1,   // != 0: a draw index ran past the tables
    F_FIFO = 2,        // != 0: a garbage queue overflowed
    F_BADARG = 3,      // != 0: an output capacity was exceeded (get_actions)
    F_CHAIN = 4,       // != 0: a wave of a chained launch gave up waiting for its predecessor (the host finishes its games un-chained)
    F_GO = 5,          // written by the HOST: releases the blocker kernel of a pre-queued rollout (TETRIS_PREQUEUE)
    F_PLACE = 6,       // != 0: a workgroup of an XCD-affine chained launch found itself on another XCD than the host expected for its queue
    F_XCC0 = 8,        // [3] 0x100 | the XCD that block 0 of the last affine launch on queue k landed on (the host's expectation for the next call)
    NFLAGS = 16
};

// ---------------------------------------------------------------- addressing
// Every access to a batch's state goes through these two functions, so the memory layout is decided here and
// nowhere else.  A `Ref` names one player-board (or one game's game words): word w lives `o` bytes past s[w * ws].
// `s` and `ws` are uniform across a wave whose lanes hold consecutive games; `o` is the lane's 32-bit byte offset.
struct Ref { uint32_t* s; uint32_t o; size_t ws; };

// Geometry of one batch: N games, P players, nw words per player-board; `state` and `gstate` are its allocations.
struct Geo { uint32_t* state; uint32_t* gstate; size_t n_games; int P; int nw; size_t stride; };     // stride: games per row (>= n_games)

#if defined(__HIPCC__)
#define TE_LAYOUT_HD __host__ __device__ __forceinline__
#else
#define TE_LAYOUT_HD static inline
#endif

// Layout "rows" (default): state[(w * P + p) * N + slot], gstate[gw * N + slot] — word w of all games is one contiguous row.
// Layout "tiles" (-DTE_TILED=1, kept for same-box A/B): the batch is cut into tiles of 64 consecutive games — the games of one
// wavefront — and a tile is contiguous: [NGWORDS game words][nw words x P players], 64 lanes each, i.e. word w of player p of
// game g lives at state[(g / 64) * tile_words + (NGWORDS + w * P + p) * 64 + g % 64].  Every access of a wave is one coalesced
// 256-byte row in both layouts; with tiles all rows of a wave sit at COMPILE-TIME offsets from one base (immediate offsets: no
// address arithmetic, -74 scalar instructions per step) and the ~8 KB a wave touches are one contiguous block.  MEASURED
// (profiles/r02/layout_chain_ab.txt, same box, 64k single-player boards, one step per launch): tiles 6.24 us per launch, rows
// 5.79 us (chained launches: 5.25 vs 5.00 us).  The step is bound by the latency of its ~30 loads, not by instruction count:
// thirty rows that lie N * 4 bytes apart are thirty requests to different memory channels, while one contiguous 8 KB block
// queues on a few.  With tiles `gstate` is the batch's main state allocation (`state` may point at a second allocation of the
// same shape — the shadow copy of split batches — whose game words are not used).
#ifndef TE_TILED
#define TE_TILED 0
#endif
constexpr int TILE = 64;
#if defined(__HIP_DEVICE_COMPILE__)
TE_LAYOUT_HD size_t wave_uniform(size_t v) { return (size_t)__builtin_amdgcn_readfirstlane((uint32_t)v); }   // same value in all lanes, told to the compiler
#else
TE_LAYOUT_HD size_t wave_uniform(size_t v) { return v; }
#endif
#if TE_TILED
TE_LAYOUT_HD size_t tile_words(const Geo& g) { return (size_t)(NGWORDS + g.nw * g.P) * TILE; }
// `uniform`: the lanes of the calling wave hold the 64 games of ONE tile (slot = 64 * wave + lane): the base is then a scalar
TE_LAYOUT_HD Ref board_ref(const Geo& g, int p, size_t slot, bool uniform = false) {
    const size_t tile = uniform ? wave_uniform(slot / TILE) : slot / TILE;
    Ref r = {g.state + tile * tile_words(g) + (size_t)(NGWORDS + p) * TILE, (uint32_t)(slot % TILE) * 4u, (size_t)g.P * TILE};
    return r;
}
TE_LAYOUT_HD Ref game_ref(const Geo& g, size_t slot, bool uniform = false) {
    const size_t tile = uniform ? wave_uniform(slot / TILE) : slot / TILE;
    Ref r = {g.gstate + tile * tile_words(g), (uint32_t)(slot % TILE) * 4u, (size_t)TILE};      // (gstate: the batch's main allocation)
    return r;
}
// player words follow the game words: word w of a board_ref = row NGWORDS + w * P + p of the tile
TE_LAYOUT_HD size_t state_words(size_t n_games, int P, int nw) { return ((n_games + TILE - 1) / TILE) * (size_t)(NGWORDS + nw * P) * TILE; }
TE_LAYOUT_HD size_t gstate_words(size_t) { return 0; }
#else
TE_LAYOUT_HD Ref board_ref(const Geo& g, int p, size_t slot, bool = false) {
    Ref r = {g.state + (size_t)p * g.stride, (uint32_t)slot * 4u, (size_t)g.P * g.stride};
    return r;
}
TE_LAYOUT_HD Ref game_ref(const Geo& g, size_t slot, bool = false) {
    Ref r = {g.gstate, (uint32_t)slot * 4u, g.stride};
    return r;
}
TE_LAYOUT_HD size_t state_words(size_t n_games, int P, int nw) { return (size_t)nw * P * n_games; }
TE_LAYOUT_HD size_t gstate_words(size_t n_games) { return (size_t)NGWORDS * n_games; }
#endif
// plain (cached) access to one word, for the kernels that touch a few words of many games
TE_LAYOUT_HD uint32_t& word_at(const Ref& r, int w) { return *(uint32_t*)((char*)(r.s + (size_t)w * r.ws) + r.o); }

}  // namespace te
