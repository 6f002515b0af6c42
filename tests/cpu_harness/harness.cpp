// TEST INFRASTRUCTURE — NOT PRODUCT CODE, NOT A FALLBACK.
//
// Compiles the product's kernel bodies (drl-tetris_amd/csrc/tetris_kernels.h, tetris_engine.h,
// tetris_tables.h — the exact sources hipcc builds for gfx950) with g++ and wraps them in plain
// loops over host memory, behind the same C symbols as include/tetris_hip.h.  Purpose: let the
// `-m "not gpu"` suite check the bitboard step logic, the RNG tables and the SoA load/store
// against the oracle in a container that has no GPU.  Only tests/ loads this library, always by
// explicit path; the product binding (drl-tetris_amd/capi.py) loads libtetris_hip.so and nothing else.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#define TE_PATH_COUNTERS 1
#include "../../drl-tetris_amd/csrc/tetris_kernels.h"

unsigned long long te::te_path_count[te::PC_NCOUNTERS];

using namespace te;

static thread_local std::string g_err;
static int fail(int code, const char* msg) { g_err = msg; return code; }

struct Tables {
    uint8_t map[8] = {0};
    int only_sz = 0;
    std::vector<uint32_t> mt;        // [624][65536]
    std::vector<float> w;            // [7][65536]
    std::vector<uint64_t> start;     // [65536] per-seed start entries
    double powtab[256];
    std::vector<uint8_t> table;      // [(chunk * 65536 + seed) * 624 + r]
    int n_chunks = 0;
    void extend() {
        int c = n_chunks;
        table.resize((size_t)65536 * CHUNK * (c + 1));
        uint8_t* out = table.data() + (size_t)65536 * CHUNK * c;
#pragma omp parallel for schedule(static)
        for (int seed = 0; seed < 65536; seed++) {
            uint64_t word = 0;
            gen_chunk_for_seed(mt.data() + seed, 65536, w.data() + seed, 65536, out + (size_t)seed * CHUNK, &word, c, map, only_sz != 0);
            if (c == 0) start[seed] = word;
        }
        n_chunks++;
    }
};
static std::map<std::string, Tables*> g_tables;

static Tables* tables_for(const uint8_t map[7]) {
    std::string key((const char*)map, 7);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return it->second;
    Tables* t = new Tables();
    memcpy(t->map, map, 7);
    t->only_sz = 1;
    for (int i = 0; i < 7; i++) if (map[i] != 2 && map[i] != 3) t->only_sz = 0;
    t->mt.resize((size_t)624 * 65536);
    t->w.resize((size_t)7 * 65536);
    t->start.resize(65536);
    for (int c = 0; c < 256; c++) t->powtab[c] = pow((double)c, 1.4 + (double)c * 0.01);
#pragma omp parallel for schedule(static)
    for (int seed = 0; seed < 65536; seed++) mt_seed(t->mt.data() + seed, 65536, (uint32_t)(int32_t)(int16_t)(uint16_t)seed);
    t->extend();
    t->extend();
    g_tables[key] = t;
    return t;
}

struct tetris_batch {
    int N, P, H;
    std::vector<uint32_t> state, gstate;
    uint32_t flags[NFLAGS] = {0};
    uint32_t margin = 64;
    uint32_t game_offset = 0;
    int split = 0, side = 0;
    int tint = 0, nw = NWORDS;
    std::vector<uint32_t> shadow;
    Tables* tab;
};

static KArgs base_args(tetris_batch* b, int n, const int32_t* idx) {
    KArgs a;
    memset(&a, 0, sizeof a);
    a.state = b->state.data(); a.gstate = b->gstate.empty() ? b->state.data() : b->gstate.data(); a.status = b->flags;
    a.table = b->tab->table.data(); a.start = b->tab->start.data(); a.combo_pow = b->tab->powtab;
    a.n_draws = (uint32_t)b->tab->n_chunks * CHUNK; a.margin = b->margin;
    a.H = b->H; a.n_games = b->N; a.n_stride = b->N; a.n_players = b->P; a.nw = b->nw; a.n = n; a.idx = idx; a.game_offset = b->game_offset;
    return a;
}
static Geo geo_of_batch(tetris_batch* b) {
    Geo g = {b->state.data(), b->gstate.empty() ? b->state.data() : b->gstate.data(), (size_t)b->N, b->P, b->nw, (size_t)b->N};
    return g;
}

template <int MODE>
static void run(tetris_batch* b, const KArgs& a, LaneCounters* total = nullptr) {
    LaneCounters sum = {0, 0, 0, 0};
    for (int i = 0; i < a.n; i++) {
        LaneCounters c = {0, 0, 0, 0};
        if (!lane_active(a, i)) continue;
        if (b->P == 1 && !b->tint) game_body<1, MODE, false>(a, i, SHAPES.s, c);
        else if (b->P == 1) game_body<1, MODE, true>(a, i, SHAPES.s, c);
        else if (b->P == 2 && !b->tint) game_body<2, MODE, false>(a, i, SHAPES.s, c);
        else if (b->P == 2) game_body<2, MODE, true>(a, i, SHAPES.s, c);
        else if (b->P == 3 && !b->tint) game_body<3, MODE, false>(a, i, SHAPES.s, c);
        else if (b->P == 3) game_body<3, MODE, true>(a, i, SHAPES.s, c);
        else if (!b->tint) game_body<4, MODE, false>(a, i, SHAPES.s, c);
        else game_body<4, MODE, true>(a, i, SHAPES.s, c);
        sum.steps += c.steps; sum.episodes += c.episodes; sum.lines += c.lines; sum.sent += c.sent;
    }
    if (total) *total = sum;
}

static int finish_call(tetris_batch* b) {
    if (b->flags[F_EXTEND]) {
        if (b->flags[F_EXTEND] >= (uint32_t)b->tab->n_chunks * CHUNK && b->tab->n_chunks < MAX_CHUNKS) b->tab->extend();
        b->flags[F_EXTEND] = 0;
    }
    if (b->flags[F_BADARG]) { b->flags[F_BADARG] = 0; return fail(TETRIS_E_ARG, "output capacity exceeded"); }
    return TETRIS_OK;
}

// host [n][P][K] -> [K][P][n], [n][P] -> [P][n]
struct KeyStage { std::vector<uint8_t> k, l; };
static int stage_keys(tetris_batch* b, int n, const uint8_t* keys, const uint8_t* lens, int max_keys, KArgs& a, KeyStage& s) {
    const int P = b->P;
    s.k.assign((size_t)n * P * max_keys, 0); s.l.assign((size_t)n * P, 0);
    for (int i = 0; i < n; i++)
        for (int p = 0; p < P; p++) {
            int len = lens[(size_t)i * P + p];
            if (len > max_keys) return fail(TETRIS_E_ARG, "lens > max_keys");
            s.l[(size_t)p * n + i] = (uint8_t)len;
            for (int k = 0; k < max_keys; k++) s.k[((size_t)k * P + p) * n + i] = keys[((size_t)i * P + p) * max_keys + k];
        }
    a.keys = s.k.data(); a.lens = s.l.data(); a.max_keys = max_keys;
    return TETRIS_OK;
}

struct OutStage { std::vector<uint8_t> done, lines, dead; };
static void stage_out(tetris_batch* b, int n, KArgs& a, OutStage& o) {
    o.done.assign(n, 0); o.lines.assign((size_t)n * b->P, 0); o.dead.assign((size_t)n * b->P, 0);
    a.done = o.done.data(); a.lines = o.lines.data(); a.dead = o.dead.data();
}
static void fetch_out(tetris_batch* b, int n, const OutStage& o, uint8_t* done, uint8_t* lines, uint8_t* dead) {
    if (done) memcpy(done, o.done.data(), n);
    for (int i = 0; i < n; i++)
        for (int p = 0; p < b->P; p++) {
            if (lines) lines[(size_t)i * b->P + p] = o.lines[(size_t)p * n + i];
            if (dead) dead[(size_t)i * b->P + p] = o.dead[(size_t)p * n + i];
        }
}

extern "C" {

const char* tetris_last_error(void) { return g_err.c_str(); }
int tetris_device_count(void) { return 0; }
int tetris_device_name(int, char* buf, int len) { if (buf && len > 0) snprintf(buf, (size_t)len, "cpu test harness (g++ build of the kernel bodies, no GPU)"); return TETRIS_OK; }
int tetris_record_size(void) { return (int)sizeof(tetris_record); }
int tetris_layout_words(void) { return NWORDS; }
int tetris_snapshot_words(const tetris_batch* b) { return b ? NGWORDS + b->P * b->nw : 0; }
int tetris_table_chunks(const tetris_batch* b) { return b ? b->tab->n_chunks : 0; }
void* tetris_device_state(tetris_batch* b) { return b ? b->state.data() : nullptr; }
void* tetris_stream(tetris_batch*) { return nullptr; }
int tetris_is_cpu_harness(void) { return 1; }

static int create_impl(tetris_batch** out, int n_games, int n_players, int height, int width, const uint8_t piece_map[7],
                       const int16_t* seeds, int split, int side, int flags = 0) {
    if (!out) return fail(TETRIS_E_ARG, "out is NULL");
    *out = nullptr;
    if (n_games < 1) return fail(TETRIS_E_ARG, "n_games must be >= 1");
    if (n_players < 1 || n_players > TETRIS_MAX_PLAYERS) return fail(TETRIS_E_ARG, "n_players must be 1..4");
    if (height < 4 || height > MAX_H) return fail(TETRIS_E_ARG, "height must be in [4, 31]");
    if (width != NCOL) return fail(TETRIS_E_ARG, "width must be 10");
    if (!piece_map) return fail(TETRIS_E_ARG, "piece_map is NULL");
    for (int i = 0; i < 7; i++) if (piece_map[i] > 6) return fail(TETRIS_E_ARG, "piece_map entries must be 0..6");
    tetris_batch* b = new tetris_batch();
    b->N = n_games; b->P = n_players; b->H = height;
    b->tint = (flags & TETRIS_FLAG_COLOURS) ? 1 : 0; b->nw = b->tint ? NWORDS_TINT : NWORDS;
    b->state.assign(state_words((size_t)n_games, n_players, b->nw), 0);
    b->gstate.assign(gstate_words((size_t)n_games), 0);
    b->tab = tables_for(piece_map);
    b->split = split; b->side = side;
    if (split && side == 1) b->shadow.assign((size_t)(UNDO_WORDS + b->nw) * (size_t)n_games, 0);
    KArgs a = base_args(b, n_games, nullptr);
    a.seeds = seeds; a.steps = side;
    if (split) run<M_SPLIT_INIT>(b, a); else run<M_INIT>(b, a);
    int rc = finish_call(b);
    if (rc) { delete b; return rc; }
    *out = b;
    return TETRIS_OK;
}

int tetris_create(tetris_batch** out, int n_games, int n_players, int height, int width, const uint8_t piece_map[7], int,
                  const int16_t* seeds) { return create_impl(out, n_games, n_players, height, width, piece_map, seeds, 0, 0); }
int tetris_create_ex(tetris_batch** out, int n_games, int n_players, int height, int width, const uint8_t piece_map[7], int,
                     const int16_t* seeds, int flags) {
    if (flags & ~TETRIS_FLAG_COLOURS) return fail(TETRIS_E_ARG, "unknown flag");
    return create_impl(out, n_games, n_players, height, width, piece_map, seeds, 0, 0, flags);
}
int tetris_create_split(tetris_batch** out, int n_games, int side, int height, int width, const uint8_t piece_map[7], int,
                        const int16_t* seeds) {
    if (side != 0 && side != 1) return fail(TETRIS_E_ARG, "side must be 0 or 1");
    return create_impl(out, n_games, 1, height, width, piece_map, seeds, 1, side);
}
int tetris_set_stream(tetris_batch*, void*, int) { return TETRIS_OK; }
static int split_stage_run(tetris_batch* b, int stage, KArgs& a, const uint32_t* const words[4], uint32_t* outw) {
    if (!b->split) return fail(TETRIS_E_ARG, "not a split batch");
    if (stage < 0 || stage > 3) return fail(TETRIS_E_ARG, "stage");
    for (int k = 0; k < 4; k++) a.xw[k] = words ? words[k] : nullptr;
    a.shadow = b->shadow.data(); a.xout = outw; a.split_side = b->side;
    for (int i = 0; i < b->N; i++) {
        if (stage == 0) split_body<0>(a, i, SHAPES.s);
        else if (stage == 1) split_body<1>(a, i, SHAPES.s);
        else if (stage == 2) split_body<2>(a, i, SHAPES.s);
        else split_body<3>(a, i, SHAPES.s);
    }
    return TETRIS_OK;
}
int tetris_split_stage_dev(tetris_batch* b, int stage, const uint8_t* rot, const uint8_t* trans, const uint8_t* acting, int ms,
                           const uint32_t* const words[4], uint32_t* outw, uint8_t* done, uint8_t* lines, uint8_t* dead) {
    KArgs a = base_args(b, b->N, nullptr);
    a.rot = rot; a.trans = trans; a.player = acting; a.ms = ms; a.done = done; a.lines = lines; a.dead = dead;
    return split_stage_run(b, stage, a, words, outw);
}
int tetris_split_rollout_stage_dev(tetris_batch* b, int stage, uint32_t policy_seed, uint64_t step, int ms, const uint32_t* const words[4],
                                   uint32_t* outw) {
    KArgs a = base_args(b, b->N, nullptr);
    a.ms = ms; a.policy_seed = policy_seed; a.first_step = step; a.steps = 1;
    return split_stage_run(b, stage, a, words, outw);
}
int tetris_rollout_totals(tetris_batch* b, uint64_t totals[4]) {
    totals[0] = totals[1] = totals[2] = totals[3] = 0;
    for (int i = 0; i < b->N; i++) {
        unsigned long long t[4];
        totals_of_game(geo_of_batch(b), i, t);
        for (int k = 0; k < 4; k++) totals[k] += t[k];
    }
    return TETRIS_OK;
}
int tetris_destroy(tetris_batch* b) { delete b; return TETRIS_OK; }
int tetris_sync(tetris_batch* b) { return finish_call(b); }
int tetris_take_errors(tetris_batch* b, uint32_t* bits) {
    if (!bits) return fail(TETRIS_E_ARG, "bits is NULL");
    *bits = (b->flags[F_FIFO] ? TETRIS_ERR_FIFO : 0u) | (b->flags[F_EXHAUSTED] ? TETRIS_ERR_STREAM : 0u);
    b->flags[F_FIFO] = 0; b->flags[F_EXHAUSTED] = 0;
    return TETRIS_OK;
}
int tetris_set_chained(tetris_batch*, int) { return TETRIS_OK; }
int tetris_set_chain_spin_limit(tetris_batch*, uint32_t) { return TETRIS_OK; }
int tetris_set_direct_dispatch(tetris_batch*, int) { return TETRIS_OK; }
int tetris_rollout_was_direct(tetris_batch*) { return 0; }
int tetris_set_xcd_affine(tetris_batch*, int) { return TETRIS_OK; }
int tetris_debug_xcd_skew(tetris_batch*, int) { return TETRIS_OK; }
int tetris_debug_code_objects(int* count, uint64_t* bytes) { if (count) *count = 0; if (bytes) *bytes = 0; return TETRIS_OK; }
int tetris_debug_stall(tetris_batch*, int, int, int) { return TETRIS_OK; }
int tetris_debug_clock_khz(tetris_batch*, int* khz) { if (khz) *khz = 0; return TETRIS_OK; }
int tetris_rollout_is_chained(tetris_batch*, int) { return 0; }
int tetris_set_game_offset(tetris_batch* b, uint64_t first) { b->game_offset = (uint32_t)first; return TETRIS_OK; }

static int check_idx(tetris_batch* b, const int32_t* idx, int n) {
    if (n < 0 || (!idx && n > b->N)) return fail(TETRIS_E_ARG, "n out of range");
    if (idx) for (int i = 0; i < n; i++) if (idx[i] < 0 || idx[i] >= b->N) return fail(TETRIS_E_ARG, "game index out of range");
    return TETRIS_OK;
}

int tetris_reset(tetris_batch* b, const int32_t* idx, int n, const int16_t* seeds) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    KArgs a = base_args(b, n, idx); a.seeds = seeds;
    if (b->split) run<M_SPLIT_RESET>(b, a); else run<M_RESET>(b, a);
    return finish_call(b);
}

int tetris_make_actions(tetris_batch* b, const int32_t* idx, int n, const uint8_t* keys, const uint8_t* lens, int max_keys) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    KArgs a = base_args(b, n, idx); KeyStage s;
    if ((rc = stage_keys(b, n, keys, lens, max_keys, a, s))) return rc;
    run<M_MAKE>(b, a);
    return finish_call(b);
}

int tetris_finish_actions(tetris_batch* b, const int32_t* idx, int n, int ms, uint8_t* done, uint8_t* lines, uint8_t* dead) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    KArgs a = base_args(b, n, idx); a.ms = ms; OutStage o; stage_out(b, n, a, o);
    run<M_FINISH>(b, a);
    fetch_out(b, n, o, done, lines, dead);
    return finish_call(b);
}

int tetris_step_keys(tetris_batch* b, const int32_t* idx, int n, const uint8_t* keys, const uint8_t* lens, int max_keys, int ms,
                     uint8_t* done, uint8_t* lines, uint8_t* dead) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    KArgs a = base_args(b, n, idx); a.ms = ms; KeyStage s; OutStage o;
    if ((rc = stage_keys(b, n, keys, lens, max_keys, a, s))) return rc;
    stage_out(b, n, a, o);
    run<M_STEP_KEYS>(b, a);
    fetch_out(b, n, o, done, lines, dead);
    return finish_call(b);
}

int tetris_step_rt(tetris_batch* b, const uint8_t* rot, const uint8_t* trans, const uint8_t* player, int ms, uint8_t* done,
                   uint8_t* lines, uint8_t* dead) {
    if (!rot || !trans) return fail(TETRIS_E_ARG, "rot/trans are NULL");
    if (player) for (int i = 0; i < b->N; i++) if (player[i] >= b->P) return fail(TETRIS_E_ARG, "player index out of range");
    KArgs a = base_args(b, b->N, nullptr); a.ms = ms; a.rot = rot; a.trans = trans; a.player = player;
    OutStage o; stage_out(b, b->N, a, o);
    run<M_STEP_RT>(b, a);
    fetch_out(b, b->N, o, done, lines, dead);
    return finish_call(b);
}

int tetris_step_rt_dev_ex(tetris_batch* b, const uint8_t* rot, const uint8_t* trans, const uint8_t* player, int ms, uint8_t* done,
                          uint8_t* lines, uint8_t* dead, int flags) {
    if (flags & ~TETRIS_STEP_AUTO_RESET) return fail(TETRIS_E_ARG, "unknown flag");
    int rc = finish_call(b); if (rc) return rc;          // (the product polls its flag words here instead)
    KArgs a = base_args(b, b->N, nullptr); a.ms = ms; a.rot = rot; a.trans = trans; a.player = player;
    a.done = done; a.lines = lines; a.dead = dead;
    if (flags & TETRIS_STEP_AUTO_RESET) run<M_STEP_RT_AUTO>(b, a); else run<M_STEP_RT>(b, a);
    return TETRIS_OK;
}
int tetris_step_rt_dev(tetris_batch* b, const uint8_t* rot, const uint8_t* trans, const uint8_t* player, int ms, uint8_t* done,
                       uint8_t* lines, uint8_t* dead) { return tetris_step_rt_dev_ex(b, rot, trans, player, ms, done, lines, dead, 0); }
int tetris_reset_dev(tetris_batch* b, const uint8_t* mask, const int16_t* seeds) {
    if (b->split) return fail(TETRIS_E_ARG, "tetris_reset_dev is not available on split batches");
    int rc = finish_call(b); if (rc) return rc;
    KArgs a = base_args(b, b->N, nullptr); a.mask = mask; a.seeds = seeds;
    if (seeds) run<M_RESET>(b, a); else run<M_RESET_SCHED>(b, a);
    return TETRIS_OK;
}

int tetris_observe_records(tetris_batch* b, const int32_t* idx, int n, tetris_record* records, uint8_t* round_over,
                           int8_t* last_winner) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    for (int i = 0; i < n; i++) {
        const Geo geo = geo_of_batch(b);
        if (b->P == 1 && !b->tint) observe_body<1, false>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
        else if (b->P == 1) observe_body<1, true>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
        else if (b->P == 2 && !b->tint) observe_body<2, false>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
        else if (b->P == 2) observe_body<2, true>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
        else if (b->P == 3 && !b->tint) observe_body<3, false>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
        else if (b->P == 3) observe_body<3, true>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
        else if (!b->tint) observe_body<4, false>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
        else observe_body<4, true>(geo, i, idx, b->H, SHAPES.s, records, round_over, last_winner);
    }
    return TETRIS_OK;
}

int tetris_snapshot(tetris_batch* b, const int32_t* idx, int n, uint32_t* blob) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    size_t total = (size_t)n * (NGWORDS + b->P * b->nw);
    for (size_t t = 0; t < total; t++) snapshot_body(geo_of_batch(b), t, idx, blob, 0);
    return TETRIS_OK;
}
int tetris_restore(tetris_batch* b, const int32_t* idx, int n, const uint32_t* blob) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    size_t total = (size_t)n * (NGWORDS + b->P * b->nw);
    for (size_t t = 0; t < total; t++) snapshot_body(geo_of_batch(b), t, idx, (uint32_t*)blob, 1);
    return TETRIS_OK;
}
int tetris_set_dead(tetris_batch* b, const int32_t* idx, int n, const uint8_t* dead) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    for (int t = 0; t < n * b->P; t++) set_dead_body(geo_of_batch(b), t, idx, dead);
    return TETRIS_OK;
}

int tetris_enumerate_drops_dev_ex(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* valid, int8_t* land_y,
                                  uint8_t* cleared, uint32_t* after, int flags) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    if (flags & ~TETRIS_ENUM_PLANAR) return fail(TETRIS_E_ARG, "unknown flag");
    for (int i = 0; i < n; i++) {
        if (b->P == 1) enumerate_body<1>(geo_of_batch(b), i, n, idx, player, b->H, SHAPES.s, valid, land_y, cleared, after, flags & TETRIS_ENUM_PLANAR);
        else if (b->P == 2) enumerate_body<2>(geo_of_batch(b), i, n, idx, player, b->H, SHAPES.s, valid, land_y, cleared, after, flags & TETRIS_ENUM_PLANAR);
        else if (b->P == 3) enumerate_body<3>(geo_of_batch(b), i, n, idx, player, b->H, SHAPES.s, valid, land_y, cleared, after, flags & TETRIS_ENUM_PLANAR);
        else enumerate_body<4>(geo_of_batch(b), i, n, idx, player, b->H, SHAPES.s, valid, land_y, cleared, after, flags & TETRIS_ENUM_PLANAR);
    }
    return TETRIS_OK;
}
int tetris_enumerate_drops(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* valid, int8_t* land_y,
                           uint8_t* cleared, uint32_t* after) { return tetris_enumerate_drops_dev_ex(b, idx, n, player, valid, land_y, cleared, after, 0); }
int tetris_enumerate_drops_dev(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* valid, int8_t* land_y,
                               uint8_t* cleared, uint32_t* after) { return tetris_enumerate_drops_dev_ex(b, idx, n, player, valid, land_y, cleared, after, 0); }
int tetris_timer_start(tetris_batch*) { return TETRIS_OK; }
int tetris_timer_stop(tetris_batch*, float* ms) { if (ms) *ms = 0.0f; return TETRIS_OK; }

int tetris_get_actions(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* keys, uint8_t* lens, int32_t* count,
                       int max_lists, int max_keys) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    const int LANE_LISTS = 16;
    const size_t lanes = (size_t)n * 40;
    std::vector<uint8_t> hc(lanes), hl(lanes * LANE_LISTS), hk(lanes * LANE_LISTS * max_keys);
    for (size_t t = 0; t < lanes; t++) {
        if (b->P == 1) actions_body<1>(geo_of_batch(b), t, idx, player, b->H, SHAPES.s, hc.data(), hl.data(), hk.data(), LANE_LISTS, max_keys, b->flags);
        else if (b->P == 2) actions_body<2>(geo_of_batch(b), t, idx, player, b->H, SHAPES.s, hc.data(), hl.data(), hk.data(), LANE_LISTS, max_keys, b->flags);
        else if (b->P == 3) actions_body<3>(geo_of_batch(b), t, idx, player, b->H, SHAPES.s, hc.data(), hl.data(), hk.data(), LANE_LISTS, max_keys, b->flags);
        else actions_body<4>(geo_of_batch(b), t, idx, player, b->H, SHAPES.s, hc.data(), hl.data(), hk.data(), LANE_LISTS, max_keys, b->flags);
    }
    if (b->flags[F_BADARG]) { b->flags[F_BADARG] = 0; return fail(TETRIS_E_ARG, "output capacity exceeded"); }
    for (int i = 0; i < n; i++) {
        int total = 0;
        for (int xi = 0; xi < 10; xi++)
            for (int r = 0; r < 4; r++) {
                const size_t lane = (size_t)i * 40 + r * 10 + xi;
                for (int k = 0; k < hc[lane]; k++) {
                    if (total >= max_lists) return fail(TETRIS_E_ARG, "more than max_lists key lists for one game");
                    const int len = hl[lane * LANE_LISTS + k];
                    lens[(size_t)i * max_lists + total] = (uint8_t)len;
                    memcpy(keys + ((size_t)i * max_lists + total) * max_keys, hk.data() + (lane * LANE_LISTS + k) * max_keys, (size_t)len);
                    total++;
                }
            }
        count[i] = total;
    }
    return TETRIS_OK;
}

int tetris_observe_packed(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* visual, uint8_t* vector,
                          uint8_t* piece) {
    int rc = check_idx(b, idx, n); if (rc) return rc;
    if (b->P > 2) return fail(TETRIS_E_ARG, "the packed observation is defined for one or two players (own / opponent's board: state_unpack.py:88-137)");
    const int cells = b->H * NCOL;
    for (int sl = 0; sl < b->P; sl++)
        for (int i = 0; i < n; i++) {
            size_t slot = idx ? (size_t)idx[i] : (size_t)i;
            int me = player ? player[i] : 0;
            int p = sl == 0 ? me : b->P - 1 - me;
            piece[(size_t)sl * n + i] = (uint8_t)observe_board(geo_of_batch(b), slot, p, b->H,
                                                               visual + ((size_t)sl * n + i) * cells, vector + ((size_t)sl * n + i) * 12);
        }
    return TETRIS_OK;
}
int tetris_observe_packed_dev(tetris_batch* b, const int32_t* idx, int n, const uint8_t* player, uint8_t* visual, uint8_t* vector,
                              uint8_t* piece) { return tetris_observe_packed(b, idx, n, player, visual, vector, piece); }

int tetris_step_rt_observe_dev(tetris_batch* b, const uint8_t* rot, const uint8_t* trans, const uint8_t* player, int ms, uint8_t* done,
                               uint8_t* lines, uint8_t* dead, int flags, const uint8_t* next_player, uint8_t* visual, uint8_t* vector,
                               uint8_t* piece) {
    if (!visual || !vector || !piece) return fail(TETRIS_E_ARG, "visual/vector/piece are NULL");
    int rc = tetris_step_rt_dev_ex(b, rot, trans, player, ms, done, lines, dead, flags);       // (the product fuses the two kernels)
    if (rc) return rc;
    return tetris_observe_packed_dev(b, nullptr, b->N, next_player, visual, vector, piece);
}

int tetris_rollout_launch(tetris_batch* b, int launches, int steps_per_launch, uint32_t policy_seed, uint64_t first_step, int ms,
                          float* elapsed_ms) {
    if (launches < 1 || steps_per_launch < 0 || steps_per_launch > 256) return fail(TETRIS_E_ARG, "launches/steps_per_launch");
    // the harness looks at the flag words after every launch, so the margin only has to cover one launch
    const uint32_t saved = b->margin;
    b->margin = (uint32_t)(2 * steps_per_launch + 16);
    if (b->margin < saved) b->margin = saved;
    int rc = TETRIS_OK;
    for (int l = 0; l < launches && !rc; l++) {
        KArgs a = base_args(b, b->N, nullptr);
        a.ms = ms; a.steps = steps_per_launch; a.policy_seed = policy_seed;
        a.first_step = first_step + (uint64_t)l * (uint64_t)steps_per_launch;
        run<M_ROLLOUT>(b, a);
        rc = finish_call(b);
    }
    b->margin = saved;
    if (elapsed_ms) *elapsed_ms = 0.0f;
    return rc;
}

int tetris_rollout_random(tetris_batch* b, int launches, int steps_per_launch, uint32_t policy_seed, uint64_t first_step, int ms,
                          uint64_t counters[4], float* elapsed_ms) {
    uint64_t before[4], after[4];
    tetris_rollout_totals(b, before);
    int rc = tetris_rollout_launch(b, launches, steps_per_launch, policy_seed, first_step, ms, elapsed_ms);
    if (rc) return rc;
    tetris_rollout_totals(b, after);
    if (counters) for (int k = 0; k < 4; k++) counters[k] += after[k] - before[k];
    return TETRIS_OK;
}

// test-only: reads (and clears) the key-interpreter path counters of tetris_engine.h
void harness_path_counts(unsigned long long* out) {
    for (int i = 0; i < PC_NCOUNTERS; i++) { out[i] = te_path_count[i]; te_path_count[i] = 0; }
}

}  // extern "C"
