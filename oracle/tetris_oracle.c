/* ==========================================================================
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See tetris_oracle.h.
 *
 * CPU restatement of the reference environment step, written from the prose
 * spec in SURVEY.md App. A/B and checked against the compiled reference
 * (oracle/_ref) through tests/golden/.  Every function cites the reference
 * file:line whose behaviour it restates (paths relative to
 * /root/reference/environment/game_backend/source/ unless noted).
 *
 * Floating point: compiled with -ffp-contract=off, no fast-math; float/double
 * mixing below mirrors the C++ promotion rules of the cited lines.
 * ========================================================================== */
#include "tetris_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ MT19937
 * libstdc++ std::mt19937 (third-party, not in /root/reference): the published
 * MT19937 of Matsumoto & Nishimura, init_genrand seeding.  KAT in
 * tests/test_oracle_kat.py: seed 1000 -> 2807145907, 882709079, 493951047
 * (SURVEY.md §8c).                                                          */
typedef struct { uint32_t s[624]; int pos; } mt_t;

static void mt_seed(mt_t *g, uint32_t seed) {
    g->s[0] = seed;
    for (int i = 1; i < 624; i++)
        g->s[i] = 1812433253u * (g->s[i - 1] ^ (g->s[i - 1] >> 30)) + (uint32_t)i;
    g->pos = 624;
}

static void mt_twist(mt_t *g) {
    for (int i = 0; i < 624; i++) {
        uint32_t y = (g->s[i] & 0x80000000u) | (g->s[(i + 1) % 624] & 0x7fffffffu);
        uint32_t v = g->s[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        g->s[i] = v;
    }
    g->pos = 0;
}

static uint32_t mt_next(mt_t *g) {
    if (g->pos >= 624) mt_twist(g);
    uint32_t y = g->s[g->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void or_mt19937_block(uint32_t seed, uint32_t *out, int n) {
    mt_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < n; i++) out[i] = mt_next(&g);
}

/* randomizer.h:21-26 — UniformRealDistribution<float>::operator():
 * dScale is (float)1 / ((float)(2^32-1) + (float)1) = 2^-32 held in a double;
 * the 64-bit draw times that double, plus (double)0.0f, rounded to float.    */
static float unit_float(uint32_t draw) {
    float span = (float)4294967295.0 + (float)1;
    double scale = (double)((float)1.0 / span);
    return (float)((double)draw * scale + (double)0.0f);
}

/* ------------------------------------------------------------------ Philox
 * Philox4x32-10 (Salmon et al., SC'11) — used only for the synthetic policy. */
void or_philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                      uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ pieces
 * SURVEY.md App. B table (probed against the reference for all 28 states):
 * rows y0..y3 as hex nibbles, bit x = column x, indexed [piece][current_rotation].
 * Restates gamePlay.cpp:116-158 (templates, spawn rotations) together with
 * pieces.cpp:5-53 (raw 3x3 / 4x4 rotation) as a lookup.                      */
static const uint16_t SHAPE_ROWS[7][4] = {
    /* L */ {0x2260, 0x0710, 0x3220, 0x4700},
    /* J */ {0x2230, 0x1700, 0x6220, 0x0740},
    /* S */ {0x2640, 0x0630, 0x1320, 0x6300},
    /* Z */ {0x2310, 0x3600, 0x4620, 0x0360},
    /* I */ {0x2222, 0x0f00, 0x4444, 0x00f0},
    /* T */ {0x0720, 0x2320, 0x2700, 0x2620},
    /* O */ {0x0660, 0x0660, 0x0660, 0x0660},
};
static const uint8_t SHAPE_VALUE[7] = {4, 3, 5, 7, 2, 1, 6}; /* gamePlay.cpp:125-139 */
static const uint8_t SPAWN_ROT[7]   = {3, 1, 3, 1, 1, 2, 0}; /* gamePlay.cpp:117     */

static int shape_cell(int kind, int rot, int gx, int gy) {
    if (kind > 6) return 0;
    int nib = (SHAPE_ROWS[kind][rot & 3] >> (4 * (3 - gy))) & 0xF;
    return (nib >> gx) & 1;
}

/* ------------------------------------------------------------------ state */
#define QCAP 256

typedef struct {
    uint8_t  cell[OR_MAX_H * OR_W];
    /* active piece (basePieces, pieces.h:7-28) */
    uint8_t  kind, rot;
    int8_t   px, py;
    uint8_t  next;
    uint8_t  dead, reward, inc_count;
    /* randomizer (randomizer.h:36-63) */
    mt_t     gen_piece, gen_hole;
    float    weight[7];
    uint32_t piece_draws, hole_draws;
    /* GarbageHandler (Garbage.h:15-37) */
    int      qlen;
    int16_t  qcount[QCAP];
    int32_t  qdelay[QCAP];
    int32_t  min_remaining;
    int      q_overflow;
    /* ComboCounter (Combo.h:8-33) */
    int32_t  combo_start, combo_time;
    uint8_t  line_count, combo_count;
    uint16_t combo_remaining;
    /* DropDelay (DropDelay.h:6-18) */
    int32_t  drop_delay, drop_time, speedup_time, lock_time;
    uint8_t  lock_armed;
    /* GamePlay scalars (gamePlay.h:58-70) and GameplayData (gamePlay.h:15-36) */
    float    incoming;
    int32_t  time_ms;
    uint16_t lines_cleared_seen;
    uint16_t lines_sent, lines_cleared, lines_blocked, garbage_cleared, max_combo;
} player_t;

typedef struct {
    uint8_t round_over;
    int8_t  last_winner;
} game_t;

struct or_batch {
    int n_games, n_players, H, W;
    uint8_t map[7];
    int only_sz;
    player_t *pl;   /* [n_games][n_players] */
    game_t *gm;
};

#define PL(b, g, p) (&(b)->pl[(size_t)(g) * (b)->n_players + (p)])

int or_record_size(void) { return (int)sizeof(or_record); }
int or_n_games(const or_batch *b) { return b->n_games; }
int or_n_players(const or_batch *b) { return b->n_players; }

/* ------------------------------------------------------------------ board */

/* gameField.cpp:10-20 BasicField::possible */
static int fits(const or_batch *b, const player_t *p, int kind, int rot, int px, int py) {
    for (int gx = 0; gx < 4; gx++)
        for (int gy = 0; gy < 4; gy++)
            if (shape_cell(kind, rot, gx, gy)) {
                int x = px + gx, y = py + gy;
                if (x < 0 || x > b->W - 1 || y < 0 || y > b->H - 1) return 0;
                if (p->cell[y * b->W + x]) return 0;
            }
    return 1;
}
static int fits_now(const or_batch *b, const player_t *p) {
    return fits(b, p, p->kind, p->rot, p->px, p->py);
}

/* gameField.cpp:22-47 mLeft / mRight / BasicField::mDown */
static int shift_piece(const or_batch *b, player_t *p, int dx, int dy) {
    if (fits(b, p, p->kind, p->rot, p->px + dx, p->py + dy)) {
        p->px = (int8_t)(p->px + dx);
        p->py = (int8_t)(p->py + dy);
        return 1;
    }
    return 0;
}

/* gameField.cpp:49-53 BasicField::hd */
static void drop_to_floor(const or_batch *b, player_t *p) {
    while (fits(b, p, p->kind, p->rot, p->px, p->py + 1)) p->py++;
}

/* gameField.cpp:93-103 kickTest: offsets tried in this fixed order relative to
 * the un-kicked position; on total failure the caller's `posX -= 2` undoes the
 * last offset (gameField.cpp:62,73,86).                                        */
static const int8_t KICK_DX[7] = {0, -1, +1, -1, +1, -2, +2};
static const int8_t KICK_DY[7] = {+1, 0, 0, +1, +1, 0, 0};

/* gameField.cpp:55-91 rcw / rccw / r180 (turn = +1, -1 (=+3), 2) */
static int rotate_piece(const or_batch *b, player_t *p, int turn) {
    int nr = (p->rot + turn) & 3;
    if (fits(b, p, p->kind, nr, p->px, p->py)) { p->rot = (uint8_t)nr; return 1; }
    for (int k = 0; k < 7; k++)
        if (fits(b, p, p->kind, nr, p->px + KICK_DX[k], p->py + KICK_DY[k])) {
            p->rot = (uint8_t)nr;
            p->px = (int8_t)(p->px + KICK_DX[k]);
            p->py = (int8_t)(p->py + KICK_DY[k]);
            return 1;
        }
    return 0;
}

/* gameField.cpp:105-110 addPiece: tile = piece index + 1 (gamePlay.cpp:146) */
static void stamp_piece(const or_batch *b, player_t *p) {
    if (p->kind > 6) return;
    for (int gx = 0; gx < 4; gx++)
        for (int gy = 0; gy < 4; gy++)
            if (shape_cell(p->kind, p->rot, gx, gy)) {
                int x = p->px + gx, y = p->py + gy;
                if (x >= 0 && x < b->W && y >= 0 && y < b->H) p->cell[y * b->W + x] = (uint8_t)(p->kind + 1);
            }
}

/* gameField.cpp:120-145 clearlines (+ removeline :112-118): rows posY..H-1 are
 * scanned bottom-up; a full row is removed, everything above shifts down and the
 * same index is scanned again.  out[0] = rows removed, out[1] = those holding an 8. */
static void clear_rows(const or_batch *b, player_t *p, int out[2]) {
    out[0] = out[1] = 0;
    for (int y = b->H - 1; y >= 0; y--) {
        int row = p->py + y;
        if (row > b->H - 1 || row < 0) continue;
        int full = 1, has_garbage = 0;
        for (int x = 0; x < b->W; x++) {
            uint8_t c = p->cell[row * b->W + x];
            if (c == 8) has_garbage = 1;
            if (c == 0) { full = 0; break; }
        }
        if (full) {
            for (int r = row; r > 0; r--) memcpy(&p->cell[r * b->W], &p->cell[(r - 1) * b->W], (size_t)b->W);
            memset(&p->cell[0], 0, (size_t)b->W);
            y++;
            out[0]++;
            if (has_garbage) out[1]++;
        }
    }
}

/* ------------------------------------------------------------------ rng */

/* randomizer.cpp:55-62 reset: integer 1000/7 = 142 */
static void weights_reset(player_t *p) {
    for (int i = 0; i < 7; i++) p->weight[i] = (float)(1000 / 7);
}

/* randomizer.cpp:10-32 getPiece */
static int draw_piece(player_t *p) {
    int chosen = 0;
    uint32_t u = mt_next(&p->gen_piece);
    p->piece_draws++;
    float ticket = unit_float(u) * (float)1000;
    for (int i = 0; i < 7; i++) {
        ticket = ticket - p->weight[i];
        if (ticket < 0) { chosen = i; break; }
    }
    float adjust = (p->weight[chosen] / (float)4) * (float)3;
    p->weight[chosen] = p->weight[chosen] - adjust;
    adjust = (float)((double)adjust / 6.0);
    for (int i = 0; i < 7; i++)
        if (i != chosen) p->weight[i] = p->weight[i] + adjust;
    return chosen;
}

/* randomizer.cpp:38-45 getHole (noStack is never passed true by the reference) */
static int draw_hole(const or_batch *b, player_t *p) {
    uint32_t u = mt_next(&p->gen_hole);
    p->hole_draws++;
    return (int)(short)(unit_float(u) * (float)b->W);
}

/* ------------------------------------------------------------------ garbage queue */

static void q_pop_front(player_t *p) {
    for (int i = 1; i < p->qlen; i++) { p->qcount[i - 1] = p->qcount[i]; p->qdelay[i - 1] = p->qdelay[i]; }
    p->qlen--;
}

/* Garbage.cpp:22-24 add: initialDelay 1000 */
static void q_add(player_t *p, int amount, int32_t t) {
    if (p->qlen >= QCAP) { p->q_overflow = 1; return; }
    p->qcount[p->qlen] = (int16_t)amount;
    p->qdelay[p->qlen] = t + 1000;
    p->qlen++;
}

/* Garbage.cpp:26-52 block: freezeDelay 450 */
static int q_block(player_t *p, int amount, int32_t t, int freeze) {
    if (p->qlen == 0) return amount;
    int32_t head_delay = p->qdelay[0];
    int blocked = 0;
    while (amount && p->qlen) {
        p->qcount[0]--;
        amount--;
        blocked++;
        if (p->qcount[0] == 0) q_pop_front(p);
    }
    p->lines_blocked = (uint16_t)(p->lines_blocked + blocked);
    if (p->qlen) {
        if (head_delay > p->qdelay[0]) p->qdelay[0] = head_delay;
        if (freeze) {
            int32_t a = p->qdelay[0] + 450, c = t + p->min_remaining + 450;
            p->qdelay[0] = a < c ? a : c;
        }
    } else
        p->min_remaining = 1000;
    return amount;
}

/* Garbage.cpp:54-72 check: addDelay 450 (Garbage.cpp:7) */
static int q_release(player_t *p, int32_t t) {
    if (p->qlen == 0) return 0;
    if (t > p->qdelay[0]) {
        int32_t next_delay = p->qdelay[0] + 450;
        if (--p->qcount[0] == 0) q_pop_front(p);
        if (p->qlen) {
            if (next_delay > p->qdelay[0]) p->qdelay[0] = next_delay;
            p->min_remaining = p->qdelay[0] - t;
        } else
            p->min_remaining = 1000;
        return 1;
    }
    int32_t left = p->qdelay[0] - t;
    if (left < p->min_remaining) p->min_remaining = left;
    return 0;
}

/* Garbage.cpp:9-14 count */
static int q_total(const player_t *p) {
    uint16_t total = 0;
    for (int i = 0; i < p->qlen; i++) total = (uint16_t)(total + p->qcount[i]);
    return total;
}

/* ------------------------------------------------------------------ combo */

double or_combo_pow(int c) { return pow((double)c, 1.4 + (double)c * 0.01); }

/* Combo.cpp:15-30 increase: integer divisions accumulate into a float, the sum
 * is added to the int32 comboTime in float and truncated back.                */
static void combo_gain(player_t *p, int32_t t, int amount) {
    if (p->combo_count == 0) { p->combo_start = t; p->combo_time = 0; }
    p->combo_count++;
    float line_time = 0;
    for (int i = 0; i < amount; i++) {
        p->line_count++;
        line_time = line_time + (float)(1000 / (int)p->line_count);
    }
    float add = (float)(800 / (int)p->combo_count) + line_time;
    p->combo_time = (int32_t)((float)p->combo_time + add);
    if (p->combo_count > p->max_combo) p->max_combo = p->combo_count;
}

/* Combo.cpp:32-48 check */
static int combo_expire(player_t *p, int32_t t) {
    int32_t left = p->combo_start + p->combo_time - t;
    p->combo_remaining = left < 0 ? 0 : (uint16_t)left;
    if (t > p->combo_start + p->combo_time && p->combo_count != 0) {
        float duration = 1.f + (float)t / 60000.f * 0.1f;
        double v = or_combo_pow(p->combo_count) * (double)duration;
        uint16_t lines = (uint16_t)(int64_t)v;
        p->combo_count = 0;
        p->line_count = 0;
        return lines;
    }
    return 0;
}

/* ------------------------------------------------------------------ player logic */

/* gamePlay.cpp:71-88 makeNewPiece / copyPiece; returns 1 when the spawn collides */
static int spawn_next(const or_batch *b, player_t *p) {
    p->kind = p->next;
    p->rot = p->kind <= 6 ? SPAWN_ROT[p->kind] : 0;
    p->px = (int8_t)((b->W - 4) / 2);
    p->py = 0;
    p->next = b->map[draw_piece(p)];
    if (!fits_now(b, p)) { stamp_piece(b, p); return 1; }
    return 0;
}

/* gamePlay.cpp:160-171 sendLines */
static int score_clears(player_t *p, const int cleared[2]) {
    p->garbage_cleared = (uint16_t)(p->garbage_cleared + cleared[1]);
    p->lines_cleared = (uint16_t)(p->lines_cleared + cleared[0]);
    if (cleared[0] == 0) { p->combo_time -= 200; return 0; }   /* Combo.cpp:50-52 noClear */
    int amount = q_block(p, cleared[0] - 1, p->time_ms, 1);
    p->lines_sent = (uint16_t)(p->lines_sent + amount);
    combo_gain(p, p->time_ms, cleared[0]);
    return amount;
}

/* gamePlay.cpp:48-52 hd_make; DropDelay.cpp:23-26 reset */
static void lock_piece(const or_batch *b, player_t *p) {
    drop_to_floor(b, p);
    stamp_piece(b, p);
    p->drop_time = p->time_ms;
    p->lock_armed = 0;
}

/* gamePlay.cpp:54-59 hd_finish: -1 = died */
static int settle(const or_batch *b, player_t *p) {
    int cleared[2];
    clear_rows(b, p, cleared);
    int sent = score_clears(p, cleared);
    if (spawn_next(b, p)) return -1;
    return sent;
}

/* gamePlay.cpp:61-69 GamePlay::mDown; DropDelay.cpp:23-26 reset, :37-41 set */
static int soft_drop(const or_batch *b, player_t *p) {
    if (shift_piece(b, p, 0, 1)) { p->drop_time = p->time_ms; p->lock_armed = 0; return 1; }
    if (!p->lock_armed) p->lock_time = p->time_ms + 400;
    p->lock_armed = 1;
    return 0;
}

/* DropDelay.cpp:3-21 check */
static int gravity_due(player_t *p, int32_t t) {
    if (t - p->speedup_time > 3000) {
        if (p->drop_delay > 200) p->drop_delay -= 10;
        else if (p->drop_delay > 100) p->drop_delay -= 5;
        else if (p->drop_delay > 50) p->drop_delay -= 2;
        else if (p->drop_delay > 10) p->drop_delay -= 1;
        p->speedup_time = t;
    }
    if (t - p->drop_time > p->drop_delay) { p->drop_time = t; return 1; }
    return 0;
}

/* gamePlay.cpp:179-204 pushGarbage / addGarbageLine: rows move up one, bottom row
 * is 8 except the hole; hole == W writes one byte past the row (harmless in the
 * reference) = no hole here.  Returns 1 when the piece dies.    */
static int push_garbage(const or_batch *b, player_t *p) {
    int hole = draw_hole(b, p);
    memmove(&p->cell[0], &p->cell[b->W], (size_t)(b->H - 1) * b->W);
    for (int x = 0; x < 10 && x < b->W; x++) p->cell[(b->H - 1) * b->W + x] = 8;
    if (hole >= 0 && hole < b->W) p->cell[(b->H - 1) * b->W + hole] = 0;
    if (p->py > 0) p->py--;
    if (!fits_now(b, p)) {
        if (p->py > 0) p->py--;
        else return 1;
    }
    return 0;
}

/* gamePlay.cpp:90-114 delayCheck */
static int tick(const or_batch *b, player_t *p, int ms) {
    p->time_ms += ms;
    if (gravity_due(p, p->time_ms)) soft_drop(b, p);
    if (p->lock_armed && p->time_ms > p->lock_time && !soft_drop(b, p)) {   /* DropDelay.cpp:43-48 */
        /* gamePlay.cpp:38-46 hd */
        lock_piece(b, p);
        return settle(b, p);
    }
    int whole = 0;
    while (p->incoming >= 1) { whole++; p->incoming = p->incoming - 1.f; }
    if (whole) q_add(p, whole, p->time_ms);
    int sent = combo_expire(p, p->time_ms);
    if (sent) {
        sent = q_block(p, sent, p->time_ms, 0);
        p->lines_sent = (uint16_t)(p->lines_sent + sent);
    }
    if (q_release(p, p->time_ms))
        if (push_garbage(b, p)) return -1;
    return sent;
}

/* gamePlay.cpp:206-216 restartRound (field.clear sets piece.piece = 7,
 * gameField.cpp:147-151); Garbage.cpp:16-20; Combo.cpp:7-13; DropDelay.cpp:28-35 */
static void restart_round(const or_batch *b, player_t *p) {
    (void)b;
    memset(p->cell, 0, sizeof p->cell);
    p->kind = 7;   /* grid and rotation fields keep their old contents in the reference;
                      they are overwritten by the spawn that always follows        */
    p->qlen = 0; p->lines_blocked = 0; p->min_remaining = 1000;
    p->combo_start = 0; p->combo_time = 0; p->max_combo = 0; p->combo_count = 0; p->line_count = 0;
    p->lines_sent = 0; p->garbage_cleared = 0; p->lines_cleared = 0;
    p->speedup_time = 0; p->drop_delay = 1000; p->drop_time = 0; p->lock_time = 0; p->lock_armed = 0;
    p->time_ms = 0;
    p->incoming = 0;
    p->lines_cleared_seen = 0;
    p->dead = 0;
}

/* gamePlay.cpp:218-230 seed; randomizer.cpp:34-36,47-49 (short truncation) */
static void seed_player(const or_batch *b, player_t *p, int16_t seed) {
    mt_seed(&p->gen_hole, (uint32_t)(int32_t)seed);
    mt_seed(&p->gen_piece, (uint32_t)(int32_t)seed);
    p->piece_draws = 0; p->hole_draws = 0;
    weights_reset(p);
    spawn_next(b, p);
    if (!b->only_sz)
        while (p->next == 2 || p->next == 3) { weights_reset(p); spawn_next(b, p); }
    spawn_next(b, p);
}

/* ------------------------------------------------------------------ batch API */

or_batch *or_create(int n_games, int n_players, int height, int width, const uint8_t piece_map[7],
                    const int16_t *seeds) {
    if (n_games < 1 || n_players < 1 || height < 4 || height > OR_MAX_H || width != OR_W) return NULL;
    or_batch *b = (or_batch *)calloc(1, sizeof *b);
    b->n_games = n_games; b->n_players = n_players; b->H = height; b->W = width;
    memcpy(b->map, piece_map, 7);
    /* PythonHandle.h:116-121 set_pieces: only_zs unless some entry is not 2/3 */
    b->only_sz = 1;
    for (int i = 0; i < 7; i++) if (piece_map[i] != 2 && piece_map[i] != 3) b->only_sz = 0;
    b->pl = (player_t *)calloc((size_t)n_games * n_players, sizeof(player_t));
    b->gm = (game_t *)calloc((size_t)n_games, sizeof(game_t));
    /* PythonHandle.cpp:5-25 init: GamePlay() leaves nextpiece = 0 (gamePlay.cpp:12-15),
     * restartRound, then seed()                                                       */
    for (int g = 0; g < n_games; g++) {
        b->gm[g].round_over = 0;
        b->gm[g].last_winner = -1;
        for (int q = 0; q < n_players; q++) {
            player_t *p = PL(b, g, q);
            p->next = 0;
            restart_round(b, p);
            seed_player(b, p, seeds ? seeds[g] : 0);
        }
    }
    return b;
}

void or_destroy(or_batch *b) {
    if (!b) return;
    free(b->pl); free(b->gm); free(b);
}

/* PythonHandle.cpp:49-71 reset + seed */
static void reset_game(or_batch *b, int g, int16_t seed) {
    game_t *G = &b->gm[g];
    G->round_over = 0;
    int winner = -1, alive = 0;
    for (int q = 0; q < b->n_players; q++) {
        player_t *p = PL(b, g, q);
        if (!p->dead) { alive++; winner = q; }
        restart_round(b, p);
    }
    G->last_winner = (int8_t)winner;
    if (b->n_players == 1) G->last_winner = 0;
    if (alive > 1) G->last_winner = -1;
    for (int q = 0; q < b->n_players; q++) seed_player(b, PL(b, g, q), seed);
}

void or_reset(or_batch *b, const int32_t *idx, int n, const int16_t *seeds) {
    for (int i = 0; i < n; i++) reset_game(b, idx ? idx[i] : i, seeds ? seeds[i] : 0);
}

/* PythonHandle.cpp:73-112 action_make */
static void press_key(const or_batch *b, player_t *p, int key) {
    switch (key) {
        case 1: shift_piece(b, p, -1, 0); break;
        case 2: while (shift_piece(b, p, -1, 0)) {} break;
        case 3: shift_piece(b, p, +1, 0); break;
        case 4: while (shift_piece(b, p, +1, 0)) {} break;
        case 5: soft_drop(b, p); break;
        case 6: while (soft_drop(b, p)) {} break;
        case 7: lock_piece(b, p); break;
        case 8: rotate_piece(b, p, 1); break;
        case 9: rotate_piece(b, p, 3); break;
        case 10: rotate_piece(b, p, 2); break;
        default: break;
    }
}

/* PythonHandle.cpp:124-136 distributeLines */
static void share_lines(or_batch *b, int g, int sender, int amount) {
    float others = (float)(b->n_players - 1);
    if (others < 1) return;
    float each = (float)amount / others;
    for (int q = 0; q < b->n_players; q++)
        if (q != sender) PL(b, g, q)->incoming = PL(b, g, q)->incoming + each;
}

/* PythonHandle.cpp:138-147 */
static void make_game(or_batch *b, int g, const uint8_t *keys, const uint8_t *lens, int max_keys) {
    if (b->gm[g].round_over) return;
    for (int q = 0; q < b->n_players; q++) {
        player_t *p = PL(b, g, q);
        if (p->dead) continue;
        for (int k = 0; k < lens[q]; k++) press_key(b, p, keys[q * max_keys + k]);
    }
}

/* PythonHandle.cpp:149-188 (+ action_finish :114-122) */
static int finish_game(or_batch *b, int g, int ms) {
    game_t *G = &b->gm[g];
    if (G->round_over) return 1;
    for (int q = 0; q < b->n_players; q++) {
        player_t *p = PL(b, g, q);
        if (p->dead) continue;
        int sent = settle(b, p);
        if (sent == -1) { p->dead = 1; break; }
        if (sent) share_lines(b, g, q, sent);
    }
    int alive = 0;
    for (int q = 0; q < b->n_players; q++) {
        player_t *p = PL(b, g, q);
        if (p->dead) continue;
        int sent = tick(b, p, ms);
        if (sent == -1) { p->dead = 1; continue; }
        if (sent) share_lines(b, g, q, sent);
        alive++;
        p->reward = (uint8_t)(p->lines_cleared - p->lines_cleared_seen);
        p->lines_cleared_seen = p->lines_cleared;
        p->inc_count = (uint8_t)q_total(p);
    }
    if ((b->n_players > 1 && alive < 2) || !alive) { G->round_over = 1; return 1; }
    return 0;
}

void or_make_actions(or_batch *b, const int32_t *idx, int n, const uint8_t *keys, const uint8_t *lens,
                     int max_keys) {
    int P = b->n_players;
    for (int i = 0; i < n; i++)
        make_game(b, idx ? idx[i] : i, keys + (size_t)i * P * max_keys, lens + (size_t)i * P, max_keys);
}

void or_finish_actions(or_batch *b, const int32_t *idx, int n, int ms, uint8_t *done) {
    for (int i = 0; i < n; i++) done[i] = (uint8_t)finish_game(b, idx ? idx[i] : i, ms);
}

/* sventon_utils.py:9-13 make_action + tetris_environment.py:102-116 perform_action */
static int step_rt_game(or_batch *b, int g, int r, int t, int player, int ms) {
    uint8_t keys[8 * 16];
    uint8_t lens[8];
    int P = b->n_players;
    for (int q = 0; q < P; q++) { keys[q * 16] = 0; lens[q] = 1; }
    int k = 0;
    uint8_t *mine = &keys[player * 16];
    for (int i = 0; i < r; i++) mine[k++] = 8;
    mine[k++] = 2;
    for (int i = 0; i < t; i++) mine[k++] = 3;
    mine[k++] = 7;
    lens[player] = (uint8_t)k;
    make_game(b, g, keys, lens, 16);
    return finish_game(b, g, ms);
}

void or_step_rt(or_batch *b, const uint8_t *rot, const uint8_t *trans, const uint8_t *player, int ms,
                uint8_t *done) {
    for (int g = 0; g < b->n_games; g++)
        done[g] = (uint8_t)step_rt_game(b, g, rot[g] & 3, trans[g] > 11 ? 11 : trans[g], player ? player[g] : 0, ms);
}

static void fill_record(const or_batch *b, const player_t *p, or_record *r) {
    memset(r, 0, sizeof *r);
    for (int y = 0; y < b->H; y++) memcpy(r->field[y], &p->cell[y * b->W], (size_t)b->W);
    for (int gy = 0; gy < 4; gy++)
        for (int gx = 0; gx < 4; gx++)
            r->grid[gy][gx] = shape_cell(p->kind, p->rot, gx, gy) ? SHAPE_VALUE[p->kind] : 0;
    r->x = p->px; r->y = p->py;
    r->piece = p->kind;
    r->tile = (uint8_t)(p->kind + 1);
    r->spawn_rot = p->kind <= 6 ? SPAWN_ROT[p->kind] : 0;
    r->cur_rot = p->rot;
    r->big = (p->kind == 4 || p->kind == 6);
    r->next = p->next; r->dead = p->dead; r->reward = p->reward; r->inc_count = p->inc_count;
    r->combo_count = p->combo_count; r->combo_remaining = p->combo_remaining;
    r->lock_armed = p->lock_armed;
    r->fifo_len = (uint8_t)(p->qlen > 255 ? 255 : p->qlen);
    r->line_count = p->line_count;
    r->fifo_overflow = (uint8_t)(p->q_overflow || p->qlen > OR_FIFO_CAP);
    r->time_ms = p->time_ms; r->incoming = p->incoming;
    r->drop_delay = p->drop_delay; r->drop_time = p->drop_time; r->speedup_time = p->speedup_time;
    r->lock_time = p->lock_time; r->min_remaining = p->min_remaining;
    r->combo_start = p->combo_start; r->combo_time = p->combo_time;
    for (int i = 0; i < p->qlen && i < OR_FIFO_CAP; i++) { r->fifo_delay[i] = p->qdelay[i]; r->fifo_count[i] = p->qcount[i]; }
    r->lines_sent = p->lines_sent; r->lines_cleared = p->lines_cleared; r->lines_blocked = p->lines_blocked;
    r->garbage_cleared = p->garbage_cleared; r->max_combo = p->max_combo;
    r->lines_cleared_seen = p->lines_cleared_seen;
    memcpy(r->weights, p->weight, sizeof r->weights);
    r->piece_draws = p->piece_draws; r->hole_draws = p->hole_draws;
}

void or_observe(const or_batch *b, const int32_t *idx, int n, or_record *records, uint8_t *round_over,
                int8_t *last_winner) {
    for (int i = 0; i < n; i++) {
        int g = idx ? idx[i] : i;
        for (int q = 0; q < b->n_players; q++) fill_record(b, PL(b, g, q), &records[(size_t)i * b->n_players + q]);
        if (round_over) round_over[i] = b->gm[g].round_over;
        if (last_winner) last_winner[i] = b->gm[g].last_winner;
    }
}

void or_copy_games(or_batch *dst, const int32_t *dst_idx, const or_batch *src, const int32_t *src_idx, int n) {
    for (int i = 0; i < n; i++) {
        int d = dst_idx ? dst_idx[i] : i, s = src_idx ? src_idx[i] : i;
        memcpy(PL(dst, d, 0), PL(src, s, 0), sizeof(player_t) * (size_t)src->n_players);
        dst->gm[d] = src->gm[s];
    }
}

void or_set_dead(or_batch *b, const int32_t *idx, int n, const uint8_t *dead) {
    for (int i = 0; i < n; i++)
        for (int q = 0; q < b->n_players; q++) PL(b, idx ? idx[i] : i, q)->dead = dead[(size_t)i * b->n_players + q];
}

/* TestField.cpp:64-125 getMask / findNextMove, drop placements only */
void or_enumerate_drops(const or_batch *b, const int32_t *idx, int n, const uint8_t *player, uint8_t *valid,
                        int8_t *land_y, uint8_t *cleared, uint8_t *after_cells) {
    const int cells = b->H * b->W;
    for (int i = 0; i < n; i++) {
        int g = idx ? idx[i] : i;
        const player_t *src = PL(b, g, player ? player[i] : 0);
        int kind = src->kind;
        int n_rot = kind == 6 ? 1 : (kind == 4 || kind == 2 || kind == 3) ? 2 : 4;   /* TestField.cpp:71-109 */
        for (int r = 0; r < 4; r++)
            for (int xi = 0; xi < 10; xi++) {
                size_t o = ((size_t)i * 4 + r) * 10 + xi;
                valid[o] = 0; land_y[o] = 0; cleared[o] = 0;
                if (after_cells) memcpy(after_cells + o * cells, src->cell, (size_t)cells);
                int x = xi - 1;
                int rot = kind == 6 ? src->rot : r;      /* O is used as it stands (TestField.cpp:71-80) */
                if (kind > 6 || r >= n_rot || x > b->W - 2) continue;
                if (!fits(b, src, kind, rot, x, 0)) continue;
                player_t tmp = *src;
                tmp.rot = (uint8_t)rot; tmp.px = (int8_t)x; tmp.py = 0;
                drop_to_floor(b, &tmp);
                stamp_piece(b, &tmp);
                valid[o] = 1; land_y[o] = tmp.py;
                if (after_cells) memcpy(after_cells + o * cells, tmp.cell, (size_t)cells);
                int out[2];
                clear_rows(b, &tmp, out);
                cleared[o] = (uint8_t)out[0];
            }
    }
}

/* ------------------------------------------------------------------ get_actions (TestField.cpp)
 * "place_block" enumeration: for every (x, rotation) that fits at the top row, the plain drop, then the
 * placements deeper in the same column range that can be reached by sliding under an overhang or by a
 * rotation with wall kick ("finesse"), each with the key list that produces it.  The search runs
 * BACKWARDS from the landing position up to row 0 and records its path, which is replayed reversed.  */
typedef struct {
    const or_batch *b;
    player_t pl;                 /* board copy + the probe piece */
    int spawn_rot;
    uint8_t path[512]; int path_len;          /* TestField::test_path                    */
    uint8_t best_path[512]; int best_len;     /* MoveInfo::path                          */
    int best_x, best_rot;                      /* MoveInfo::posX / rot                    */
    uint8_t *keys, *lens; int max_lists, max_keys, n_lists;
} probe_t;

typedef struct { uint8_t rot; int8_t x, y; } pose_t;
static pose_t pose_get(const probe_t *t) { pose_t p = {t->pl.rot, t->pl.px, t->pl.py}; return p; }
static void pose_set(probe_t *t, pose_t p) { t->pl.rot = p.rot; t->pl.px = p.x; t->pl.py = p.y; }
static int probe_fits(const probe_t *t) { return fits_now(t->b, &t->pl); }
static void path_push(probe_t *t, int v) { if (t->path_len < 512) t->path[t->path_len++] = (uint8_t)v; }

/* TestField.cpp:3-35 convert + emission of one key list */
static int convert_code(int m) {
    switch (m) { case 255: return 3; case 254: return 1; case 253: return 6; case 252: return 5;
                 case 241: return 8; case 242: return 10; case 243: return 9; default: return m; }
}
static void emit_begin(probe_t *t, int *n) { *n = 0; (void)t; }
static void emit_key(probe_t *t, int *n, int code) {
    if (t->n_lists < t->max_lists && *n < t->max_keys) t->keys[(size_t)t->n_lists * t->max_keys + *n] = (uint8_t)convert_code(code);
    (*n)++;
}
static void emit_end(probe_t *t, int n) {
    if (t->n_lists < t->max_lists) t->lens[t->n_lists] = (uint8_t)(n < t->max_keys ? n : t->max_keys);
    t->n_lists++;
}
/* TestField.cpp:37-42 addRotationValue, :44-49 makeStartSequence */
static void emit_rotation(probe_t *t, int *n, int value) { if (value < 0) value += 4; if (value) emit_key(t, n, 240 + value); }
static void emit_start_moves(probe_t *t, int *n, int x) {
    int mid = (t->b->W - 4) / 2;
    if (x > mid) for (int i = 0; i < x - mid; i++) emit_key(t, n, 255);
    else for (int i = 0; i < mid - x; i++) emit_key(t, n, 254);
}

/* TestField.cpp:392-410 moveUp */
static int probe_move_up(probe_t *t) {
    int landed = 0;
    t->pl.py++;
    if (!probe_fits(t)) landed = 1;
    t->pl.py--;
    int count = 0;
    do { t->pl.py--; count++; } while (probe_fits(t));
    count--;
    t->pl.py++;
    if (count && landed) return 253;
    return count & 255;
}

/* TestField.cpp:166-172 setFinesseMove */
static int probe_commit(probe_t *t) {
    memcpy(t->best_path, t->path, (size_t)t->path_len);
    t->best_len = t->path_len; t->best_x = t->pl.px; t->best_rot = t->pl.rot;
    return 1;
}

/* TestField.cpp:202-238 tryLeft / tryRight (tail recursion written as a loop) */
static int probe_slide(probe_t *t, int dir, int clear_first) {
    for (;;) {
        if (!shift_piece(t->b, &t->pl, dir, 0)) return 0;
        if (clear_first) t->path_len = 0;
        clear_first = 0;
        path_push(t, dir < 0 ? 255 : 254);
        int up = probe_move_up(t);
        if (up) path_push(t, up);
        if (t->pl.py == 0) return probe_commit(t);
    }
}

/* TestField.cpp:240-259 tryUp */
static int probe_up(probe_t *t, int turn) {
    t->path_len = 0;
    int up = probe_move_up(t);
    path_push(t, turn + 240);
    if (up) path_push(t, up);
    if (t->pl.py == 0) return probe_commit(t);
    if (probe_slide(t, -1, 0)) return 1;
    t->path_len = 0;
    path_push(t, turn + 240);
    if (up) path_push(t, up);
    return probe_slide(t, +1, 0);
}

/* TestField.cpp:280-356 doWallKick: un-rotate (with the kick offsets mirrored in y) to a pose from which
 * the forward rotation, kicks included, lands exactly on the target pose                              */
static int probe_wallkick(probe_t *t) {
    static const int8_t RDX[7] = {0, -1, +1, -1, +1, -2, +2};
    static const int8_t RDY[7] = {-1, 0, 0, -1, -1, 0, 0};
    const pose_t target = pose_get(t);
    int r, found = 0;
    for (r = 0; r < 4; r++) {
        if (r == target.rot) continue;
        t->pl.rot = (uint8_t)r;                 /* raw rotation keeps the position */
        if (probe_fits(t)) {
            int turn = (target.rot - r) & 3;
            int ok = probe_up(t, turn);
            pose_set(t, target);
            return ok;
        }
        int base_x = t->pl.px, base_y = t->pl.py, k;
        for (k = 0; k < 7; k++) {
            t->pl.px = (int8_t)(base_x + RDX[k]); t->pl.py = (int8_t)(base_y + RDY[k]);
            if (probe_fits(t)) break;
        }
        if (k < 7) { found = 1; break; }
        t->pl.px = (int8_t)base_x; t->pl.py = (int8_t)base_y;
    }
    if (!found) { pose_set(t, target); return 0; }
    int turn = (target.rot - r) & 3;
    const pose_t from = pose_get(t);
    if (turn == 0) { pose_set(t, target); return 0; }
    rotate_piece(t->b, &t->pl, turn == 1 ? 1 : turn == 2 ? 2 : 3);     /* gameField.cpp:55-91, with kicks */
    if (t->pl.px != target.x || t->pl.py != target.y) { pose_set(t, target); return 0; }
    pose_set(t, from);
    if (probe_up(t, turn)) { pose_set(t, target); return 1; }
    pose_set(t, from);
    t->path_len = 0;
    path_push(t, 240 + turn);
    if (probe_slide(t, -1, 0)) { pose_set(t, target); return 1; }
    pose_set(t, from);
    if (t->path_len > 1) { t->path_len = 0; path_push(t, 240 + turn); }
    int ok = probe_slide(t, +1, 0);
    pose_set(t, target);
    return ok;
}

/* TestField.cpp:358-390 r180KeepPos */
static void probe_flip_keep_pos(probe_t *t) {
    t->pl.rot = (uint8_t)((t->pl.rot + 2) & 3);
    int k = t->pl.kind, r = t->pl.rot;
    if (k == 4 || k == 3) { if (r == 0) t->pl.px++; else if (r == 1) t->pl.py++; else if (r == 2) t->pl.px--; else t->pl.py--; }
    if (k == 2) { if (r == 0) t->pl.px--; else if (r == 1) t->pl.py--; else if (r == 2) t->pl.px++; else t->pl.py++; }
}

/* TestField.cpp:261-278 reverseWallkick */
static int probe_reverse_kick(probe_t *t) {
    if (t->pl.kind == 6) return 0;
    if (shift_piece(t->b, &t->pl, +1, 0)) { shift_piece(t->b, &t->pl, -1, 0); return 0; }
    if (shift_piece(t->b, &t->pl, -1, 0)) { shift_piece(t->b, &t->pl, +1, 0); return 0; }
    const pose_t here = pose_get(t);
    if (probe_wallkick(t)) { pose_set(t, here); return 1; }
    if (t->pl.kind == 2 || t->pl.kind == 3 || t->pl.kind == 4) {
        probe_flip_keep_pos(t);
        int ok = probe_wallkick(t);
        pose_set(t, here);
        return ok;
    }
    pose_set(t, here);
    return 0;
}

/* TestField.cpp:189-200 finesseIsPossible */
static int probe_reachable(probe_t *t) {
    const pose_t here = pose_get(t);
    if (probe_reverse_kick(t)) { pose_set(t, here); return 1; }
    pose_set(t, here);
    if (probe_slide(t, -1, 1)) { pose_set(t, here); return 1; }
    int ok = probe_slide(t, +1, 1);
    pose_set(t, here);
    return ok;
}

/* TestField.cpp:127-164 useFinesseMove (use_mask == 2 branch) */
static void probe_emit_finesse(probe_t *t) {
    int n;
    emit_begin(t, &n);
    emit_rotation(t, &n, t->best_rot - t->spawn_rot);
    emit_start_moves(t, &n, t->best_x);
    for (int i = t->best_len - 1; i >= 0; i--) {
        int v = t->best_path[i];
        if (v < 240) for (int k = 0; k < v; k++) emit_key(t, &n, 252);
        else emit_key(t, &n, v);
    }
    emit_key(t, &n, 7);
    emit_end(t, n);
}

/* TestField.cpp:113-125 findNextMove + :174-187 tryAllFinesseMoves */
static void probe_column(probe_t *t) {
    int n;
    emit_begin(t, &n);
    emit_rotation(t, &n, (int)t->pl.rot - t->spawn_rot);
    emit_start_moves(t, &n, t->pl.px);
    emit_key(t, &n, 7);
    emit_end(t, n);
    drop_to_floor(t->b, &t->pl);
    const pose_t landed = pose_get(t);
    for (int y = landed.y + 2; y < t->b->H - 1; y++) {
        pose_set(t, landed);
        t->pl.py = (int8_t)y;
        if (probe_fits(t)) {
            drop_to_floor(t->b, &t->pl);
            y = t->pl.py;
            if (probe_reachable(t)) probe_emit_finesse(t);
        }
    }
    pose_set(t, landed);
}

/* PythonHandle.cpp:190 get_actions -> gamePlay.cpp:232-239 getMask(2) -> TestField.cpp:64-111 */
int or_get_actions(or_batch *b, int game, int player, uint8_t *keys, uint8_t *lens, int max_lists, int max_keys) {
    probe_t *t = (probe_t *)calloc(1, sizeof *t);
    t->b = b;
    t->pl = *PL(b, game, player);
    t->keys = keys; t->lens = lens; t->max_lists = max_lists; t->max_keys = max_keys;
    int kind = t->pl.kind;
    t->spawn_rot = kind <= 6 ? SPAWN_ROT[kind] : 0;
    int start_rot = t->pl.rot;
    int n_rot = kind == 6 ? 1 : (kind == 4 || kind == 2 || kind == 3) ? 2 : 4;
    if (kind <= 6)
        for (int x = -1; x < b->W - 1; x++)
            for (int r = 0; r < n_rot; r++) {
                t->pl.px = (int8_t)x; t->pl.py = 0;
                t->pl.rot = (uint8_t)(kind == 6 ? start_rot : r);
                if (!probe_fits(t)) continue;
                probe_column(t);
            }
    int n = t->n_lists;
    free(t);
    return n;
}

/* SURVEY.md §8(d) synthetic workload                                          */
static int16_t episode_seed(int g, uint32_t e) {
    return (int16_t)(uint16_t)((12345u + 7919u * (uint32_t)g + 104729u * e) & 0xFFFFu);
}

void or_rollout_random(or_batch *b, uint32_t policy_seed, uint64_t first_step, int steps, int ms,
                       uint32_t *episode, uint64_t counters[4], int threads) {
    or_rollout_random_shard(b, policy_seed, first_step, steps, ms, episode, counters, threads, 0u);
}

void or_rollout_random_shard(or_batch *b, uint32_t policy_seed, uint64_t first_step, int steps, int ms,
                             uint32_t *episode, uint64_t counters[4], int threads, uint32_t game_offset) {
    uint64_t n_steps = 0, n_episodes = 0, n_lines = 0, n_sent = 0;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(+ : n_steps, n_episodes, n_lines, n_sent)
#endif
    for (int g = 0; g < b->n_games; g++) {
        for (int s = 0; s < steps; s++) {
            uint64_t step = first_step + (uint64_t)s;
            uint32_t w[4];
            or_philox4x32_10(policy_seed, 0u, game_offset + (uint32_t)g, (uint32_t)step, (uint32_t)(step >> 32), 0u, w);
            int r = (int)(w[0] & 3u), t = (int)(w[1] % 10u);
            int player = b->n_players > 1 ? (int)(step % (uint64_t)b->n_players) : 0;
            uint16_t sent_before = 0;
            for (int q = 0; q < b->n_players; q++) sent_before = (uint16_t)(sent_before + PL(b, g, q)->lines_sent);
            int done = step_rt_game(b, g, r, t, player, ms);
            n_steps++;
            for (int q = 0; q < b->n_players; q++) {
                player_t *p = PL(b, g, q);
                if (!p->dead) n_lines += p->reward;
            }
            uint16_t sent_after = 0;
            for (int q = 0; q < b->n_players; q++) sent_after = (uint16_t)(sent_after + PL(b, g, q)->lines_sent);
            n_sent += (uint16_t)(sent_after - sent_before);
            if (done) {
                n_episodes++;
                episode[g]++;
                reset_game(b, g, episode_seed((int)(game_offset + (uint32_t)g), episode[g]));
            }
        }
    }
    (void)threads;
    counters[0] += n_steps; counters[1] += n_episodes; counters[2] += n_lines; counters[3] += n_sent;
}
