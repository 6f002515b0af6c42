/* ==========================================================================
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the DRL-Tetris / SpeedBlocks environment step.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (drl-tetris_amd/, include/tetris_hip.h) never links, imports or calls it.
 *
 * Pinning: every entry point here is checked bit-for-bit against the compiled
 * reference (oracle/_ref, built by oracle/Makefile from the sources in
 * /root/reference) through the golden traces in tests/golden/ — see
 * tests/test_oracle_golden.py.  The reference has no tests of its own
 * (SURVEY.md §4), so those traces are the pin.
 *
 * Representation: one uint8 cell per square (0 empty, 1..7 tile, 8 garbage),
 * exactly the values the reference's State.field view exposes — on purpose a
 * different representation from the packed-column bitboards of the HIP path.
 * ========================================================================== */
#ifndef TETRIS_ORACLE_H
#define TETRIS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_MAX_H 32
#define OR_W 10          /* the reference hard-codes 10 in gamePlay.cpp:202 */
#define OR_FIFO_CAP 16   /* record capacity; the oracle's own queue is larger */

/* Everything the reference can show about one player-board, in one flat
 * record (mirrored by a numpy dtype in oracle/oracle.py and produced in the
 * same layout by the HIP library's tetris_observe_records()).               */
typedef struct or_record {
    uint8_t  field[OR_MAX_H][OR_W];   /* State.field; rows >= H are zero       */
    uint8_t  grid[4][4];              /* State.piece (piece-grid values)       */
    int8_t   x, y;                    /* State.x / State.y                     */
    uint8_t  piece;                   /* basePieces.piece (0..6, 7 = none)     */
    uint8_t  tile;                    /* basePieces.tile                       */
    uint8_t  spawn_rot;               /* basePieces.rotation                   */
    uint8_t  cur_rot;                 /* basePieces.current_rotation           */
    uint8_t  big;                     /* basePieces.lpiece                     */
    uint8_t  next;                    /* State.nextpiece                       */
    uint8_t  dead;                    /* State.dead                            */
    uint8_t  reward;                  /* State.reward                          */
    uint8_t  inc_count;               /* State.inc_lines                       */
    uint8_t  combo_count;             /* State.combo_count                     */
    uint16_t combo_remaining;         /* State.combo_time                      */
    uint8_t  lock_armed;              /* DropDelay.lockdown                    */
    uint8_t  fifo_len;                /* GarbageHandler.garbage.size()         */
    uint8_t  line_count;              /* ComboCounter.lineCount (private)      */
    uint8_t  fifo_overflow;           /* record/fifo capacity exceeded         */
    int32_t  time_ms;                 /* GamePlay.time_ms                      */
    float    incoming;                /* GamePlay.incoming_lines               */
    int32_t  drop_delay;              /* DropDelay.dropDelay                   */
    int32_t  drop_time;               /* DropDelay.dropDelayTime               */
    int32_t  speedup_time;            /* DropDelay.increaseDropDelayTime       */
    int32_t  lock_time;               /* DropDelay.lockdownTime                */
    int32_t  min_remaining;           /* GarbageHandler.minRemaining           */
    int32_t  combo_start;             /* ComboCounter.comboStart (private)     */
    int32_t  combo_time;              /* ComboCounter.comboTime (private)      */
    int32_t  fifo_delay[OR_FIFO_CAP]; /* Garbage.delay                         */
    int16_t  fifo_count[OR_FIFO_CAP]; /* Garbage.count                         */
    uint16_t lines_sent;              /* GameplayData.linesSent                */
    uint16_t lines_cleared;           /* GameplayData.linesCleared             */
    uint16_t lines_blocked;           /* GameplayData.linesBlocked             */
    uint16_t garbage_cleared;         /* GameplayData.garbageCleared           */
    uint16_t max_combo;               /* GameplayData.maxCombo                 */
    uint16_t lines_cleared_seen;      /* GamePlay.linesCleared                 */
    float    weights[7];              /* randomizer.cogP                       */
    uint32_t piece_draws;             /* outputs consumed from piece_gen       */
    uint32_t hole_draws;              /* outputs consumed from hole_gen        */
} or_record;

typedef struct or_batch or_batch;

int         or_record_size(void);

/* PythonHandle(n_players, [H, W]) x n_games, after set_pieces(map)
 * (PythonHandle.h:116-121, PythonHandle.cpp:5-25).  Every game is created
 * seeded with seeds[g] (NULL: 0), i.e. in the state the reference's ctor
 * leaves behind with time()==seed.                                           */
or_batch   *or_create(int n_games, int n_players, int height, int width,
                      const uint8_t piece_map[7], const int16_t *seeds);
void        or_destroy(or_batch *b);
int         or_n_games(const or_batch *b);
int         or_n_players(const or_batch *b);

/* PythonHandle::reset() with time()==seed (PythonHandle.cpp:49-71).
 * idx NULL => games 0..n-1.                                                  */
void        or_reset(or_batch *b, const int32_t *idx, int n, const int16_t *seeds);

/* PythonHandle::make_actions (PythonHandle.cpp:138-147).
 * keys[n][P][max_keys], lens[n][P].                                          */
void        or_make_actions(or_batch *b, const int32_t *idx, int n,
                            const uint8_t *keys, const uint8_t *lens, int max_keys);
/* PythonHandle::finish_actions (PythonHandle.cpp:149-188); done[n] = return. */
void        or_finish_actions(or_batch *b, const int32_t *idx, int n, int ms, uint8_t *done);

/* tetris_environment.perform_action with the SVENton key encoding
 * (sventon_utils.py:9-13: [8]*r + [2] + [3]*t + [7]) for `player`, [0] for the
 * others (tetris_environment.py:102-116); all games.  done/rot/trans/player [N]. */
void        or_step_rt(or_batch *b, const uint8_t *rot, const uint8_t *trans,
                       const uint8_t *player, int ms, uint8_t *done);

/* records[n][P]; round_over[n]; last_winner[n]                                */
void        or_observe(const or_batch *b, const int32_t *idx, int n, or_record *records,
                       uint8_t *round_over, int8_t *last_winner);

/* PythonHandle::copy / set on whole games (PythonHandle.cpp:36-42)            */
void        or_copy_games(or_batch *dst, const int32_t *dst_idx, const or_batch *src,
                          const int32_t *src_idx, int n);

/* Write State fields the way Python does on a live handle (state.py:11,16;
 * scripts/eval.py:156): dead flags.                                           */
void        or_set_dead(or_batch *b, const int32_t *idx, int n, const uint8_t *dead /*[n][P]*/);

/* PythonHandle::get_actions(player) -> masks[player].action
 * (PythonHandle.cpp:190, TestField.cpp:64-415).  Returns the number of key
 * lists; list i has lens[i] keys at keys[i*max_keys ...].  Lists beyond
 * max_lists / keys beyond max_keys are dropped (return value still counts).   */
int         or_get_actions(or_batch *b, int game, int player, uint8_t *keys, uint8_t *lens,
                           int max_lists, int max_keys);

/* Drop part of TestField::getMask (TestField.cpp:64-125): for absolute rotation r = 0..3 and x = -1..W-2
 * (index xi = x+1), in the reference's enumeration order x-major, r-minor: the current piece of `player`
 * placed at (x, 0) with raw rotation r — if it fits there it is hard-dropped (gameField.cpp:49-53) and stamped.
 * Rotations the reference does not enumerate (r >= 1 for O, r >= 2 for I/S/Z) are invalid.
 * valid/land_y/cleared [n][4][10]; after_cells [n][4][10][H*W] (stamped, before line clear; NULL to skip).   */
void        or_enumerate_drops(const or_batch *b, const int32_t *idx, int n, const uint8_t *player,
                               uint8_t *valid, int8_t *land_y, uint8_t *cleared, uint8_t *after_cells);

/* Seeded synthetic rollout used by bench.py's cpu_baseline leg and by the
 * parity tests: SURVEY.md §8(d) policy (Philox4x32-10 keyed (policy_seed, game,
 * step)), auto-reset with seed16 = (12345 + 7919 g + 104729 e) mod 65536.
 * Runs `steps` env-steps on every game with `threads` OpenMP threads and
 * accumulates counters[0..3] = {env_steps, episodes, lines_cleared, garbage_sent}. */
void        or_rollout_random(or_batch *b, uint32_t policy_seed, uint64_t first_step, int steps,
                              int ms, uint32_t *episode /*[N] in/out*/, uint64_t counters[4],
                              int threads);
/* same, for a shard whose game 0 has global id `game_offset` (policy and seed schedule use global ids) */
void        or_rollout_random_shard(or_batch *b, uint32_t policy_seed, uint64_t first_step, int steps,
                                    int ms, uint32_t *episode, uint64_t counters[4], int threads,
                                    uint32_t game_offset);

/* Known-answer helpers                                                         */
void        or_mt19937_block(uint32_t seed, uint32_t *out, int n);  /* first n tempered outputs */
void        or_philox4x32_10(uint32_t key0, uint32_t key1, uint32_t c0, uint32_t c1,
                             uint32_t c2, uint32_t c3, uint32_t out[4]);
double      or_combo_pow(int combo_count);      /* pow(c, 1.4 + c*0.01), Combo.cpp:41 */

#ifdef __cplusplus
}
#endif
#endif
