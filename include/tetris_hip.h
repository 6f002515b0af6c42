/* tetris_hip.h — C ABI of the MI355X-native batched Tetris environment (libtetris_hip.so).
 *
 * This is the drop-in boundary for the environment-step path of mightypirate1/DRL-Tetris.
 * In the reference that boundary is the pybind11 module `tetris_env`
 * (environment/game_backend/source/PythonHandle.h:113-340) consumed by
 * environment/tetris_environment.py and environment/tetris_environment_vector.py; one
 * `PythonHandle` = one game.  Here one `tetris_batch` = N games resident in HBM on one GPU and
 * every entry point works on a list of game indices, so the per-env Python loops of
 * tetris_environment_vector.py:55-120 become one kernel launch.  Each function below names the
 * reference interface it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain C types only; caller owns every host buffer, the library owns all device memory.
 *  - return value: 0 = OK, negative = error (TETRIS_E_*); tetris_last_error() gives the text
 *    (thread-local).  There is no CPU fallback: without a HIP device tetris_create fails.
 *  - `idx` = int32 game indices (unique within one call), NULL = games 0..n-1.
 *  - per-player arrays are [n][P] (game-major) on the host side.
 *  - a batch is not thread-safe; different batches are independent (one HIP stream each).
 *  - functions ending in _dev take DEVICE pointers, enqueue on the batch's stream and return
 *    without synchronising (zero-copy callers, benchmarks); all others are synchronous.
 *    Host index / player arrays are validated (TETRIS_E_ARG); device-resident ones cannot be, so an
 *    out-of-range game index or player in a d_idx / d_player array is clamped into the batch by the kernel.
 *  - board height 4..31, width 10 (the reference hard-codes 10, gamePlay.cpp:202), 1..4 players per game (the reference
 *    takes any n_players, PythonHandle.cpp:5-25; none of its presets or agents uses more than two).  Three and four players
 *    run through the general one-lane-per-game kernel (all of a game's players in one lane); the packed observation
 *    (own / opponent planes), the one-launch step + observation, split batches and chained launches are one- and
 *    two-player features.
 */
#ifndef TETRIS_HIP_H
#define TETRIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TETRIS_OK 0
#define TETRIS_E_ARG (-1)        /* bad argument */
#define TETRIS_E_HIP (-2)        /* a HIP runtime call failed (no device, out of memory, ...) */
#define TETRIS_E_STREAM (-3)     /* (no longer returned: see TETRIS_ERR_STREAM) */
#define TETRIS_E_FIFO (-4)       /* (no longer returned: see TETRIS_ERR_FIFO) */

/* Capacity errors are confined to the game they happen in.  The reference's garbage queue is an unbounded vector
 * (Garbage.h:27) and its generators never run out (randomizer.h:44-50); here a board holds at most 8 pending garbage
 * packets and an episode at most 39 936 piece draws.  A game that exceeds either ends its round in that step (`done`), its
 * boards carry the bits below in tetris_record.fifo_overflow until the game is reset, and every other game of the batch
 * goes on untouched.  tetris_take_errors tells whether any game of the batch was ended this way since the last call.   */
#define TETRIS_ERR_FIFO 1u       /* a 9th garbage packet arrived while 8 were pending: it was dropped */
#define TETRIS_ERR_STREAM 2u     /* the episode ran past the RNG tables: the pieces dealt in that step are wrong */
/* Not a capacity error and nothing is wrong with any game: reported once by tetris_take_errors after a chained rollout call had
 * to finish some of its games un-chained (see tetris_set_chained) — the results are the same, the call was slow, and chained
 * launches are off for the batch from then on.                                                                          */
#define TETRIS_ERR_CHAIN_FELL_BACK 4u

#define TETRIS_MAX_H 32
#define TETRIS_MAX_PLAYERS 4   /* players per game (PythonHandle(n_players, ...): PythonHandle.cpp:5-25) */
#define TETRIS_W 10
#define TETRIS_FIFO_CAP 16

/* Everything the reference can show about one player-board (State views PythonHandle.h:54-82 and
 * the pickled members PythonHandle.h:123-308), flattened.  Same layout as the test oracle's record. */
typedef struct tetris_record {
    uint8_t  field[TETRIS_MAX_H][TETRIS_W]; /* State.field (occupancy: 0 / 1)                  */
    uint8_t  grid[4][4];                    /* State.piece                                      */
    int8_t   x, y;                          /* State.x, State.y                                 */
    uint8_t  piece, tile, spawn_rot, cur_rot, big;   /* basePieces (pieces.h:7-28)             */
    uint8_t  next, dead, reward, inc_count, combo_count;   /* State.*                           */
    uint16_t combo_remaining;               /* State.combo_time                                 */
    uint8_t  lock_armed, fifo_len, line_count, fifo_overflow;
    int32_t  time_ms;
    float    incoming;
    int32_t  drop_delay, drop_time, speedup_time, lock_time;   /* DropDelay.h:6-18              */
    int32_t  min_remaining;
    int32_t  combo_start, combo_time;
    int32_t  fifo_delay[TETRIS_FIFO_CAP];
    int16_t  fifo_count[TETRIS_FIFO_CAP];
    uint16_t lines_sent, lines_cleared, lines_blocked, garbage_cleared, max_combo, lines_cleared_seen;
    float    weights[7];                    /* not tracked on the GPU (always 0)                */
    uint32_t piece_draws, hole_draws;
} tetris_record;

typedef struct tetris_batch tetris_batch;

const char *tetris_last_error(void);
int         tetris_device_count(void);                 /* < 0 on error                          */
int         tetris_device_name(int device, char *buf, int len);   /* "<name> (<arch>, <n> CUs)"  */
int         tetris_record_size(void);
int         tetris_snapshot_words(const tetris_batch *b);   /* uint32 words per game in a snapshot */

/* replaces: tetris_env.set_pieces(map) + PythonHandle(n_players, [H, W]) for n_games games
 * (PythonHandle.h:116-121,309; PythonHandle.cpp:5-25).  seeds[n_games] (host, NULL = all 0) stand
 * in for time(NULL) at construction (PythonHandle.cpp:68-71).                                   */
int tetris_create(tetris_batch **out, int n_games, int n_players, int height, int width,
                  const uint8_t piece_map[7], int device, const int16_t *seeds);
/* same with options.  TETRIS_FLAG_COLOURS: also track the tile value of every square (1..7 = piece index + 1,
 * gamePlay.cpp:146; 8 = garbage, gamePlay.cpp:202) in three extra bit-planes per board, so that tetris_record.field
 * holds exactly what the reference's State.field shows and GameplayData.garbageCleared is produced.  Costs 120 B more
 * state per board; off by default because state_dict only uses field > 0.                                          */
#define TETRIS_FLAG_COLOURS 1
int tetris_create_ex(tetris_batch **out, int n_games, int n_players, int height, int width,
                     const uint8_t piece_map[7], int device, const int16_t *seeds, int flags);
int tetris_destroy(tetris_batch *b);
int tetris_sync(tetris_batch *b);                      /* drain the stream, surface sticky errors */
int tetris_take_errors(tetris_batch *b, uint32_t *bits);   /* synchronises; *bits = TETRIS_ERR_* seen since the last call, then cleared */

/* replaces: PythonHandle.reset() with time(NULL) == seeds[i] (PythonHandle.cpp:49-71)           */
int tetris_reset(tetris_batch *b, const int32_t *idx, int n, const int16_t *seeds);

/* replaces: PythonHandle.make_action(list[list[int]]) (PythonHandle.cpp:138-147).
 * keys[n][P][max_keys] uint8, lens[n][P] uint8 (host).                                          */
int tetris_make_actions(tetris_batch *b, const int32_t *idx, int n, const uint8_t *keys,
                        const uint8_t *lens, int max_keys);
/* replaces: PythonHandle.finish_action(ms) (PythonHandle.cpp:149-188).  done[n]; lines[n][P] =
 * State.reward, dead[n][P] = State.dead after the call (either may be NULL).                    */
int tetris_finish_actions(tetris_batch *b, const int32_t *idx, int n, int ms, uint8_t *done,
                          uint8_t *lines, uint8_t *dead);
/* make_action + finish_action in one launch = tetris_environment.perform_action
 * (tetris_environment.py:102-116).                                                              */
int tetris_step_keys(tetris_batch *b, const int32_t *idx, int n, const uint8_t *keys,
                     const uint8_t *lens, int max_keys, int ms, uint8_t *done, uint8_t *lines,
                     uint8_t *dead);
/* perform_action on ALL games with the SVENton (rotation, translation) encoding
 * [8]*r + [2] + [3]*t + [7] for `player[g]`, [0] for the others (sventon_utils.py:9-13).
 * rot/trans/player/done [N], lines/dead [N][P]; player NULL = player 0.                         */
int tetris_step_rt(tetris_batch *b, const uint8_t *rot, const uint8_t *trans, const uint8_t *player,
                   int ms, uint8_t *done, uint8_t *lines, uint8_t *dead);
/* same, device pointers, asynchronous.  d_lines/d_dead are [P][N] (player-major) on the device. */
int tetris_step_rt_dev(tetris_batch *b, const uint8_t *d_rot, const uint8_t *d_trans,
                       const uint8_t *d_player, int ms, uint8_t *d_done, uint8_t *d_lines,
                       uint8_t *d_dead);
/* same with flags.  TETRIS_STEP_AUTO_RESET: a game whose round ended in this step is reset inside the same launch —
 * the `env.reset(env=[done idxs])` of the worker loop (drl_tetris/worker.py:157-166 -> PythonHandle.cpp:49-71) without
 * a host round trip.  The seed is the next one of the built-in schedule, seed16 = (12345 + 7919 game + 104729 episode)
 * mod 65536 with game = global game id (tetris_set_game_offset) and episode = the game's count of resets so far;
 * d_done / d_lines / d_dead report the step as it ended, BEFORE the reset.
 * The asynchronous entry points never drain the stream: requests to extend the RNG tables reach the host through
 * flag words in pinned memory, and the host lets at most 241 launches run ahead of what it has seen of them.       */
#define TETRIS_STEP_AUTO_RESET 1
int tetris_step_rt_dev_ex(tetris_batch *b, const uint8_t *d_rot, const uint8_t *d_trans,
                          const uint8_t *d_player, int ms, uint8_t *d_done, uint8_t *d_lines,
                          uint8_t *d_dead, int flags);
/* One iteration of the agent loop in ONE launch (drl_tetris/worker.py:91-118: perform_action, then get_state + the unpacker
 * for the next decision): tetris_step_rt_dev_ex followed by tetris_observe_packed_dev(b, NULL, N, d_next_player, ...) of the
 * stepped (and, with TETRIS_STEP_AUTO_RESET, reset) state — produced from the registers the step leaves behind instead of a
 * second kernel that reads the state back.  d_next_player[N] = the player each game's next decision is for (NULL: player 0;
 * out-of-range entries are clamped); d_visual [P][N][H][10], d_vector [P][N][12], d_piece [P][N] as tetris_observe_packed.
 * Same outputs, bit for bit, as the two calls.  Colour batches, odd heights and outputs that are not 4-byte aligned run
 * the two kernels back to back; not available on split batches.                                                        */
int tetris_step_rt_observe_dev(tetris_batch *b, const uint8_t *d_rot, const uint8_t *d_trans,
                               const uint8_t *d_player, int ms, uint8_t *d_done, uint8_t *d_lines,
                               uint8_t *d_dead, int flags, const uint8_t *d_next_player,
                               uint8_t *d_visual, uint8_t *d_vector, uint8_t *d_piece);
/* replaces: reset() of the games selected by a DEVICE-side mask (d_mask[N], non-zero = reset; NULL = all games),
 * asynchronous.  d_seeds[N] int16 (device) or NULL = next seed of the built-in schedule (see above).              */
int tetris_reset_dev(tetris_batch *b, const uint8_t *d_mask, const int16_t *d_seeds);

/* replaces: reading PythonHandle.states[p].* / __getstate__() (PythonHandle.h:54-82,123-308).
 * records[n][P]; round_over[n]; last_winner[n] (PythonHandle.last_winner); NULLs allowed.       */
int tetris_observe_records(tetris_batch *b, const int32_t *idx, int n, tetris_record *records,
                           uint8_t *round_over, int8_t *last_winner);

/* replaces: state_processors.state_dict + agent_utils/state_unpack.unpacker in its default SVENton configuration
 * (state_processors.py:23-54; state_unpack.py:88-137: observation_mode 'separate', player_mode 'separate',
 * separate_piece): NN-ready observations written by one kernel.  Slot 0 = the board of player[i] ("me"), slot 1 = the
 * opponent (only when n_players == 2).  All outputs uint8:
 *   visual [S][n][H][W]  field > 0
 *   vector [S][n][12]    x, y, inc_lines, min(25000, combo_time + 50) / 100, combo_count, nextpiece one-hot (7)
 *   piece  [S][n]        index of the current piece (state_dict "piece_idx")
 * Host pointers (synchronous); the _dev variant takes device pointers and only enqueues.                          */
int tetris_observe_packed(tetris_batch *b, const int32_t *idx, int n, const uint8_t *player, uint8_t *visual,
                          uint8_t *vector, uint8_t *piece);
int tetris_observe_packed_dev(tetris_batch *b, const int32_t *d_idx, int n, const uint8_t *d_player,
                              uint8_t *d_visual, uint8_t *d_vector, uint8_t *d_piece);

/* replaces: PythonHandle.copy() / .set() (PythonHandle.cpp:36-42): exact state incl. RNG position.
 * blob[n][tetris_snapshot_words()] uint32 (host).  Blobs move between batches of equal geometry
 * and piece map.                                                                                 */
int tetris_snapshot(tetris_batch *b, const int32_t *idx, int n, uint32_t *blob);
int tetris_restore(tetris_batch *b, const int32_t *idx, int n, const uint32_t *blob);
/* replaces: Python writing State.dead on a live handle (data_types/state.py:11,16). dead[n][P]  */
int tetris_set_dead(tetris_batch *b, const int32_t *idx, int n, const uint8_t *dead);

/* replaces: the drop part of PythonHandle.get_actions(player) + simulate_actions(finalize=False)
 * (TestField.cpp:64-125 getMask/findNextMove; tetris_environment.py:87-100): for every listed game, the current
 * piece of player[i] (NULL = player 0) placed at (x, 0), x = xi - 1 for xi = 0..9, with absolute rotation r = 0..3
 * (only the rotations the reference enumerates: 1 for O, 2 for I/S/Z, 4 for L/J/T); where it fits it is hard-dropped
 * and stamped.  valid/land_y/cleared [n][4][10] (cleared = rows a finalize would remove);
 * after [n][4][10][10] = the stamped board's column bitboards before line clear (NULL to skip).  One lane per
 * (game, r, x): 40 lanes per board.                                                                              */
int tetris_enumerate_drops(tetris_batch *b, const int32_t *idx, int n, const uint8_t *player, uint8_t *valid,
                           int8_t *land_y, uint8_t *cleared, uint32_t *after);

/* same with device pointers (d_idx / d_player may be NULL), asynchronous on the batch's stream                        */
int tetris_enumerate_drops_dev(tetris_batch *b, const int32_t *d_idx, int n, const uint8_t *d_player,
                               uint8_t *d_valid, int8_t *d_land_y, uint8_t *d_cleared, uint32_t *d_after);
/* same with flags.  TETRIS_ENUM_PLANAR: rotation-minor planes — d_valid / d_land_y / d_cleared are [n][10][4] (game, column
 * index xi, rotation r) and d_after is [10][n][10][4]: column c of the placement (game i, rotation r, column index xi) at
 * d_after[((c * n + i) * 10 + xi) * 4 + r].  The four rotations of a (game, column) pair are adjacent, so the kernel writes
 * each result array with one 4-byte store and each afterstate column with one 16-byte store per lane, 1 KB contiguous per
 * wavefront; a consumer that feeds the afterstates to a network reads one column plane [n][10][4] at a time.
 * d_valid / d_land_y / d_cleared must be 4-byte aligned, d_after 16-byte aligned.                                      */
#define TETRIS_ENUM_PLANAR 1
int tetris_enumerate_drops_dev_ex(tetris_batch *b, const int32_t *d_idx, int n, const uint8_t *d_player,
                                  uint8_t *d_valid, int8_t *d_land_y, uint8_t *d_cleared, uint32_t *d_after, int flags);

/* replaces: PythonHandle.get_actions(player); masks[player].action (PythonHandle.cpp:190, TestField.cpp:64-415):
 * the reference's exact ordered key lists of the "place_block" action type — every (x, rotation) drop plus the
 * tuck / spin placements found by its backwards search — for the current piece of player[i] (NULL = player 0).
 * count[n] = number of lists; list k of game i: lens[i][k] keys at keys[i][k][0..]; masks[player].mask of the reference after
 * this call is count[i] ones (TestField.cpp:113-133 pushes a 1 beside every list).  Python applies
 * data_types.action_list (dedupe, null-move policy) on top.  TETRIS_E_ARG if a game has more than max_lists lists
 * or a list more than max_keys keys (64 / 48 always suffice for 10-wide boards up to 31 rows).                   */
int tetris_get_actions(tetris_batch *b, const int32_t *idx, int n, const uint8_t *player, uint8_t *keys,
                       uint8_t *lens, int32_t *count, int max_lists, int max_keys);

/* Built-in synthetic rollout = the worker loop of drl_tetris/worker.py:91-118 with a random policy
 * (SURVEY.md §8d): per env-step  Philox4x32-10(policy_seed; game, step) -> (r = w0 & 3,
 * t = w1 mod 10), acting player = step mod P, perform_action, auto-reset of finished games with
 * seed16 = (12345 + 7919 game + 104729 episode) mod 65536.  Runs `launches` kernel launches of
 * `steps_per_launch` env-steps each on all N games (state stays in registers inside a launch).
 * counters[4] += {env_steps, episodes, lines_cleared, garbage_lines_sent}.  elapsed_ms (optional)
 * = HIP-event time from before the first to after the last launch on the batch's stream.        */
int tetris_rollout_random(tetris_batch *b, int launches, int steps_per_launch, uint32_t policy_seed,
                          uint64_t first_step, int ms, uint64_t counters[4], float *elapsed_ms);
/* The launches of tetris_rollout_random alone: nothing but the `launches` step kernels lies between the two HIP events
 * and between call and return (plus one final stream synchronisation) — the region bench.py times.  Counters are read
 * with tetris_rollout_totals before and after, outside that region.                                                */
int tetris_rollout_launch(tetris_batch *b, int launches, int steps_per_launch, uint32_t policy_seed,
                          uint64_t first_step, int ms, float *elapsed_ms);

/* Chained launches of the built-in rollout (default on; single-player batches with one or more steps per launch, two-player
 * batches with one step per launch, on the batch's own stream): a game's step E depends only on the same game's step E - 1,
 * so consecutive launches rotate over three streams and are ordered per WAVE — the games of a wave wait, inside the kernel,
 * for an epoch word that the same wave of the previous launch publishes after its state stores have drained — instead of per
 * launch by the stream (where every launch waits for the slowest wave of the whole previous launch plus the kernel boundary).
 * Results are bit-identical.
 * CHAINING PRESUMES THAT THE DEVICE'S WAVE SLOTS ARE THIS BATCH'S TO USE.  A wave that waits keeps its slot; launches are
 * therefore chained only while three of them — for larger batches two, over two streams — fit on the device together, and that
 * fit is computed as if nothing else ran on the GPU: kernels of the host application (a policy network on another stream), of
 * another process or of a second copy of this library take slots the computation does not know of, and can keep a launch from
 * being resident beside its successor.  No wave waits unboundedly: after `polls` polls of its epoch word (default 2^22, about
 * 2 s; tetris_set_chain_spin_limit, or TETRIS_CHAIN_SPIN_LIMIT in the environment when the batch is created) it gives up and
 * leaves its games untouched, as do the same waves of the launches behind it.  The call then waits for its streams, steps the
 * games that were left behind to the end of the call with the un-chained kernel, and returns TETRIS_OK with the same results;
 * tetris_take_errors reports TETRIS_ERR_CHAIN_FELL_BACK once, and chained launches stay off for the batch until
 * tetris_set_chained(b, 1).  On a GPU that the batch shares with other work, switch chaining off up front:
 * tetris_set_chained(b, 0) — or TETRIS_NO_CHAIN=1 — costs 5.7 us per launch instead of 4.0 and cannot starve anything.
 * The three chain streams belong to the device and are shared by all its batches (their chained calls exclude each other).
 * Side effect a host application may notice: the three streams are created with three different stream priorities (that is how
 * the HIP runtime is made to keep them on three hardware queues, where alone they overlap; TETRIS_CHAIN_PRIO=0: equal
 * priorities); the batch's own stream has the default priority.
 * on = 0: every launch on the batch's one stream.                                                                       */
int tetris_set_chained(tetris_batch *b, int on);
/* DIRECT DISPATCH of long chained calls.  A chained launch takes the GPU 4 us; hipLaunchKernel costs the calling thread 2.4-4.2 us
 * of it, depending on the process.  Calls of at least `min_launches` launches (default 16) therefore do not go through
 * hipLaunchKernel: the library writes their AQL packets itself, into three HSA user-mode queues of its own per GPU (created on the
 * first such call; shared by the device's batches, whose chained calls exclude each other anyway): 0.2-0.5 us of host time per
 * launch, no helper threads, 1.5-2 % shorter calls.  Same kernels, same machine code (the gfx950 code object is taken from this
 * library's own fat binary and loaded through the HSA loader), same hand-over protocol, same results; a queue's launches are
 * ordered like a stream's (barrier bit), its first packet acquires at system scope.  Such a call first waits for what the batch's
 * stream still holds and returns with its launches retired, like every chained call.  A queue that has been idle takes 10 us to
 * start its first wave, as a stream does: in a 20-launch call the two paths are level (one player) or the queues 5-8 % ahead (two
 * players); calls of a handful of launches stay on the streams (profiles/r03/direct_dispatch.txt).  If the queues cannot be set up (no HSA agent for the HIP device, no host-visible device
 * memory for the kernel arguments, code object not found), the batch keeps launching through its streams.
 * min_launches = 0: never; n > 0: calls of at least n launches; < 0: the default (TETRIS_DIRECT_MIN in the environment, else 16;
 * TETRIS_DIRECT=0: batches are created with 0).  Pre-queued calls (TETRIS_PREQUEUE) go through the streams.              */
int tetris_set_direct_dispatch(tetris_batch *b, int min_launches);
/* 0 if the batch's last tetris_rollout_launch / tetris_rollout_random went through streams, 1 if through those queues, 2 if through
 * them with the XCD-affine kernel (below)                                                                                */
int tetris_rollout_was_direct(tetris_batch *b);
/* XCD-AFFINE chained launches (default on; one-player batches, direct dispatch only).  The MI355X has eight XCDs, each with an L2 of
 * its own that is coherent for its own CUs only; a queue deals a launch's workgroups round-robin over the XCDs, from a start of
 * its own.  With the plain hand-over a game's state therefore crosses the fabric twice per step (written through by one XCD, read
 * by another).  In the affine form the workgroup that finds itself on XCD x (hardware register XCC_ID) takes the game block
 * (b & ~7) | x: a game is stepped on the same XCD in every launch, its state and epoch word stay in that XCD's L2 (plain stores,
 * L1-bypassing loads), and the packets between a queue's first and last carry no cache maintenance; the last one releases at
 * system scope, so memory is current when the call returns.  3.5-3.6 us per launch instead of 4.0 at 64k boards, same results.
 * Relied on: one XCD's L2 is coherent for that XCD's CUs.  Checked, not relied on: that the eight workgroups of a group of eight
 * sit on eight different XCDs — the queues' start XCDs are measured when the queues are made, the form is used only if every
 * queue dealt 1024 workgroups round-robin twice, and in every launch every workgroup compares its XCC_ID with what that
 * measurement predicts; one that is elsewhere touches nothing, its games look abandoned to the next launch, and the call ends
 * like any chained call whose waves gave up (finished un-chained, exact, TETRIS_ERR_CHAIN_FELL_BACK) with this form off for
 * the batch from then on.  on = 0: the plain hand-over.  TETRIS_AFFINE=0 in the environment: batches are created with it off. */
int tetris_set_xcd_affine(tetris_batch *b, int on);
/* TEST AID for the check above: `skew` (0..7) is added to the start XCDs the kernels are told                              */
int tetris_debug_xcd_skew(tetris_batch *b, int skew);
/* TEST AID (needs no GPU): the gfx950 code objects direct dispatch would load — found in this library's own fat binary —: their
 * number and total size in bytes.  0 objects = direct dispatch cannot work with this build (e.g. a compressed offload bundle).  */
int tetris_debug_code_objects(int *count, uint64_t *bytes);
/* polls of its predecessor's epoch word after which a waiting wave of a chained launch gives up (0 = the default, 2^22);
 * one poll is an agent-scope load and a short sleep, about 0.5 us.                                                     */
int tetris_set_chain_spin_limit(tetris_batch *b, uint32_t polls);
/* MEASUREMENT AID: the GPU's shader clock of the moment in kHz (a 50 us probe kernel on the batch's stream; synchronous).      */
int tetris_debug_clock_khz(tetris_batch *b, int *khz);
/* TEST AID for the give-up path above (nothing in the product calls it): enqueues a kernel that idles for `microseconds` —
 * which = 0..2: on that one of the three chain streams (and of the device's own queues, see tetris_set_direct_dispatch), so the
 * launches the next rollout call puts there start late;
 * which = 3: on the batch's stream; which = -1: on a stream of its own, as workgroups that hold `percent` % of the device's
 * wave slots meanwhile (a co-tenant).  Asynchronous.                                                                   */
int tetris_debug_stall(tetris_batch *b, int which, int microseconds, int percent);
/* Environment variables read by the library (measurement aids; none changes a result):
 *   TETRIS_NO_CHAIN=1   batches are created with chained launches off (tetris_set_chained)
 *   TETRIS_NO_DUO=1     two-player single steps through k_game<2> (both players of a game in one lane) instead of k_duo
 *   TETRIS_GRAPH=1      un-chained rollout launches are replayed from HIP graphs of 128 kernel nodes (under rocprofv3 a plain
 *                       launch costs the host more than the kernel takes; from a graph the profiled kernels are back to back)
 *   TETRIS_PREQUEUE=1   (read per call) tetris_rollout_launch of <= 600 chained launches parks its streams behind a blocker kernel
 *                       that runs until the host has queued every launch of the call (it then sets a flag word in pinned memory):
 *                       the GPU-paced launch period, without the host's launch cost
 *   TETRIS_CHAIN_SPIN_LIMIT=<polls>  default of tetris_set_chain_spin_limit for batches created afterwards
 *   TETRIS_CHAIN_PRIO=0 the chain streams are created with equal priorities (they may then share a hardware queue)
 *   TETRIS_CHAIN_DEPTH=1..3  at most that many chained launches in flight; 1 = the chained kernel on one stream, i.e. dispatches
 *                       serialised by the stream (what per-dispatch PMC counters need: profiles/pmc_summary.py)
 *   TETRIS_EXT_EVENTS=0 chained calls record their timing / join events as packets of their own instead of attaching them to the
 *                       first and last kernels (hipExtLaunchKernel, the default)
 *   TETRIS_ENQUEUE_THREADS_MIN=<n>  (stream path) chained calls of at least n launches (default 256) are enqueued by one host thread per chain
 *                       stream (a launch costs the host 2.7-4.0 us, the GPU needs one every 4.0); 0 = always one thread
 *   TETRIS_GATE_GROUP=<n>  launches per run-ahead group (default 120: at most 241 in flight), 8..120
 *   TETRIS_DIRECT=0     batches are created with direct dispatch off; TETRIS_DIRECT_MIN=<n>: its default threshold
 *                       (tetris_set_direct_dispatch)
 *   TETRIS_AFFINE=0     no XCD-affine launches (tetris_set_xcd_affine)
 *   TETRIS_DIRECT_UNDER_TOOLS=1  direct dispatch also with a profiling tool library in the process (ROCP_TOOL_LIBRARIES, HSA_TOOLS_LIB or
 *                       LD_PRELOAD naming rocprof* / roctracer: by default the launches then stay on the streams)
 *   TETRIS_DIRECT_FENCE=none, TETRIS_DIRECT_EDGE=agent|system, TETRIS_DIRECT_PRIO=high  (experiments) direct dispatch: fence scope of
 *                       the packets between a queue's first and last / of its first and last packet; queue priority
 *   TETRIS_TIMING=1     tetris_rollout_launch prints its host-side costs (enqueue per launch, gate waits, until drained) to stderr */
/* 1 if tetris_rollout_launch / tetris_rollout_random would chain launches of `steps_per_launch` steps on this batch, 0 if not
 * (switched off — by the caller or by a fall-back —, caller-owned stream, split or colour batch, or not even two launches fit
 * on the device together: a waiting wave keeps its slot, so chaining is only used where it cannot keep the launch it waits
 * for from being dispatched).                                                                                            */
int tetris_rollout_is_chained(tetris_batch *b, int steps_per_launch);

/* Global id of this batch's game 0 (default 0): the built-in rollout keys its policy and its
 * reset-seed schedule by global game id, so that N batches on N GPUs simulate N*n_games distinct
 * games (the reference's equivalent: N worker containers, docker-compose.yaml:27).               */
int tetris_set_game_offset(tetris_batch *b, uint64_t first_game_id);

/* ---- split mode: the two players of a game on different GPUs (BASELINE config 5) ------------------------------
 * A split batch holds ONE side (player index `side`, 0 or 1) of n_games two-player games; the batch holding the
 * other side is normally on another GPU and created with the same seeds.  A step is three stages with one 32-bit
 * exchange word per board after each; the caller moves the words (RCCL all-gather in drl-tetris_amd/distributed.py).
 * replaces: PythonHandle::distributeLines + the winner logic across players (PythonHandle.cpp:124-136,151-187).
 *   stage 0: key interpreter for the acting side ([8]*r+[2]+[3]*t+[7]) + loop 1 (side 1 speculatively) -> words A
 *   stage 1: delayCheck of my player.  Side 0 runs it after the A exchange, side 1 after player 0's B words arrived
 *   stage 2: lines arriving after my tick, winner / round_over -> done[n], lines[n], dead[n] (any may be NULL)
 *   stage 3: stage 2 of this step and stage 0 of the NEXT step in one launch (one pass over the state instead of two): for
 *            loops that know the next action when a step ends — d_rot / d_trans / d_acting are the next step's, d_done /
 *            d_lines / d_dead describe the step that ends, d_out receives the next step's A words.  A step of such a loop
 *            is two launches (stage 1, stage 3) and three exchanges.
 * d_words: HOST array of four device pointers to uint32 [n] — my A words, the opponent's A words, player 0's B words,
 * player 1's B words (entries a stage does not read may be NULL): the words a stage wrote (d_out) and the rows an
 * all-gather delivered are read where they lie, nothing is copied in between.  d_out [n]: this stage's words (stages 0, 1, 3).
 * All data pointers are device pointers; everything is enqueued on the batch's stream.  tetris_reset() on a split
 * batch applies the two-player winner rule.
 * Why three exchanges and not two: within one step the reference's order makes player 0's tick depend on player 1's loop-1
 * words (A), player 1's tick on player 0's tick words (B0), and `done` on player 1's tick words (B1) — three messages that
 * depend on each other.  Folding B1 into the next step's A exchange would need a second speculative shadow state on side 0
 * (it would play the next key list before knowing whether the round had ended) and deliver `done` one step late.      */
int tetris_create_split(tetris_batch **out, int n_games, int side, int height, int width,
                        const uint8_t piece_map[7], int device, const int16_t *seeds);
int tetris_split_stage_dev(tetris_batch *b, int stage, const uint8_t *d_rot, const uint8_t *d_trans,
                           const uint8_t *d_acting, int ms, const uint32_t *const d_words[4], uint32_t *d_out,
                           uint8_t *d_done, uint8_t *d_lines, uint8_t *d_dead);
/* One stage of a split-mode step of the built-in synthetic rollout (same policy and reset-seed schedule as
 * tetris_rollout_random, keyed by global game id and `step`; acting player = step mod 2): stage 0 draws the action on
 * the device, stage 2 counts and auto-resets finished games — identically on both sides, so no host round trip is
 * needed inside a rollout; stage 3 = stage 2 of `step` + stage 0 of `step + 1`.  d_words / d_out as for
 * tetris_split_stage_dev.                                                                                           */
int tetris_split_rollout_stage_dev(tetris_batch *b, int stage, uint32_t policy_seed, uint64_t step, int ms,
                                   const uint32_t *const d_words[4], uint32_t *d_out);
/* cumulative counters of the built-in rollouts of this batch: totals[4] = {env_steps, episodes, lines_cleared,
 * garbage_sent}, each the sum over the games of a per-game word the step kernels keep (env_steps is COUNTED on the
 * device, one increment per game and step, not computed from the launch arguments); synchronous.                  */
int tetris_rollout_totals(tetris_batch *b, uint64_t totals[4]);

/* Run the batch on a caller-owned HIP stream (e.g. torch's current stream) so that its kernels are ordered with the
 * caller's copies and collectives without host synchronisation.  external != 0: use `hip_stream` as given — NULL is
 * then the legacy default stream, which is what torch.cuda.current_stream() usually is; external == 0: back to the
 * batch's own stream (hip_stream ignored).                                                                        */
int tetris_set_stream(tetris_batch *b, void *hip_stream, int external);

/* HIP-event stopwatch on the batch's stream, for timing sequences of _dev calls: start records an event, stop records
 * another, waits for it and returns the milliseconds in between.                                                    */
int tetris_timer_start(tetris_batch *b);
int tetris_timer_stop(tetris_batch *b, float *elapsed_ms);

/* plumbing for zero-copy callers (torch / another HIP library)                                   */
void *tetris_device_state(tetris_batch *b);            /* uint32 [NWORDS][P][N]                  */
void *tetris_stream(tetris_batch *b);                  /* hipStream_t                            */
int   tetris_layout_words(void);                       /* NWORDS                                 */
int   tetris_table_chunks(const tetris_batch *b);      /* RNG-table chunks currently resident    */

#ifdef __cplusplus
}
#endif
#endif
