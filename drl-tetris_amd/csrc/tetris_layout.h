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
    G_EPISODE = 1,  // episodes finished by the built-in rollout (SURVEY.md §8d seed schedule)
    G_LINES = 2,    // lines cleared during built-in rollouts (cumulative; summed by k_totals)
    G_SENT = 3,     // garbage lines sent during built-in rollouts (cumulative)
    NGWORDS = 4
};

// sticky status bits written by kernels into the batch's device status word
enum Status : uint32_t {
    ST_NEED_EXTEND = 1u,        // some board is within `margin` draws of the end of the RNG tables
    ST_STREAM_EXHAUSTED = 2u,   // a draw index ran past the tables (results invalid)
    ST_FIFO_OVERFLOW = 4u,      // more than FIFO_CAP pending garbage packets (results invalid)
    ST_BAD_ARGUMENT = 8u,
};

}  // namespace te
