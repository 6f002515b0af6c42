// RNG tables: the reference's two std::mt19937 engines per player, folded into one byte table.
//
// Facts used (all from /root/reference/environment/game_backend/source):
//  * PythonHandle::seed gives every player's hole_gen AND piece_gen the same value, truncated to
//    int16 (PythonHandle.cpp:68-71, randomizer.cpp:34-36,47-49) => one raw MT19937 stream per
//    16-bit seed, read at two positions (piece draws, hole draws).
//  * every consumer takes exactly one 32-bit output per event (randomizer.cpp:12,39).
//  * the piece picked from draw k depends on the 7 float weights, which are themselves a
//    deterministic recurrence over draws 0..k-1 restarted by GamePlay::seed (gamePlay.cpp:218-230,
//    randomizer.cpp:10-32,55-62) => the *sequence of dealt pieces* is a pure function of the seed
//    (and of the batch-wide piece map, PythonHandle.h:116-121).
//  => table[seed][k] = piece dealt by draw k (low nibble) | hole chosen by draw k (high nibble),
//     start[seed] (8 bytes) = j | table(j) << 8 | table(j+1) << 16 | pieces of the aligned group of 8 draws that
//     holds draw j+2, with j = index of the first draw accepted by the S/Z redraw loop.  After a reset a board holds
//     current = piece(j), next = piece(j+1), piece_draws = j+2, hole_draws = 0 — one 8-byte load.
//  A board also carries the dealt pieces of its current aligned group of 8 draws in one state word
//  (W_PIECE_GROUP), so it reads the table once per 8 pieces instead of pulling a 128-byte line per piece.
//
// std::mt19937 itself is libstdc++ (third party, not vendored in the reference): the published
// MT19937 (Matsumoto & Nishimura 1998), init_genrand seeding; KAT in tests (seed 1000 ->
// 2807145907, 882709079, 493951047).
//
// Everything here is scalar per seed and `__host__ __device__`: the GPU runs one lane per seed with
// the 624-word state strided [word][seed] in HBM (coalesced); the test-only CPU harness calls the
// same functions.
#pragma once
#include "tetris_layout.h"

#if defined(__HIPCC__)
#define TE_HD __host__ __device__ __forceinline__
#else
#define TE_HD static inline
#endif

namespace te {

// init_genrand; state word i lives at mt[i * stride]
TE_HD void mt_seed(uint32_t* mt, size_t stride, uint32_t seed) {
    uint32_t prev = seed;
    mt[0] = prev;
    for (int i = 1; i < 624; i++) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        mt[(size_t)i * stride] = prev;
    }
}

// in-place generation of the next 624 untempered words
TE_HD void mt_twist(uint32_t* mt, size_t stride) {
    uint32_t first = mt[0];
    uint32_t cur = first;
    for (int i = 0; i < 624; i++) {
        uint32_t nxt = (i == 623) ? mt[0] : mt[(size_t)(i + 1) * stride];
        uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
        int j = i + 397;
        if (j >= 624) j -= 624;
        uint32_t v = mt[(size_t)j * stride] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[(size_t)i * stride] = v;
        cur = nxt;
    }
    (void)first;
}

TE_HD uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// randomizer.h:21-26: the draw times a double 2^-32, plus (double)0.0f, rounded to float
TE_HD float unit_float(uint32_t draw) {
    const double scale = 1.0 / 4294967296.0;
    return (float)((double)draw * scale + 0.0);
}

// randomizer.cpp:10-32 getPiece on weights w[i*ws]
TE_HD int pick_piece(float* w, size_t ws, uint32_t draw) {
    int chosen = 0;
    float ticket = unit_float(draw) * 1000.0f;
    bool found = false;
    for (int i = 0; i < 7; i++) {
        ticket = ticket - w[(size_t)i * ws];
        if (!found && ticket < 0.0f) { chosen = i; found = true; }
        // (the reference breaks out of the loop; later subtractions do not matter)
    }
    float adjust = (w[(size_t)chosen * ws] / 4.0f) * 3.0f;
    w[(size_t)chosen * ws] = w[(size_t)chosen * ws] - adjust;
    adjust = (float)((double)adjust / 6.0);
    for (int i = 0; i < 7; i++)
        if (i != chosen) w[(size_t)i * ws] = w[(size_t)i * ws] + adjust;
    return chosen;
}

// randomizer.cpp:38-45 getHole, FIELD_WIDTH = 10
TE_HD int pick_hole(uint32_t draw) { return (int)(short)(unit_float(draw) * 10.0f); }

// Generates chunk `chunk` (draws chunk*624 .. chunk*624+623) for one seed.
//   mt      : 624 state words, strided; must hold the state left by the previous chunk (or mt_seed)
//   w       : 7 weights, strided; carried between chunks
//   out     : 624 bytes of this seed's row in the chunk table
//   first_ok: written when chunk == 0
// low nibbles of the 8 table bytes of one aligned group of draws -> one word, nibble k = piece of draw 8g + k
TE_HD uint32_t group_word(const uint8_t* bytes8) {
    uint32_t w = 0;
    for (int k = 0; k < 8; k++) w |= (uint32_t)(bytes8[k] & 0xFu) << (4 * k);
    return w;
}

TE_HD void gen_chunk_for_seed(uint32_t* mt, size_t stride, float* w, size_t ws, uint8_t* out,
                              uint64_t* start_word, int chunk, const uint8_t* map, bool only_sz) {
    uint8_t first_ok_store = 0;
    uint8_t* first_ok = &first_ok_store;
    mt_twist(mt, stride);
    bool redraw_open = (chunk == 0);       // gamePlay.cpp:218-230: weights reset before every draw
    if (chunk == 0) *first_ok = 0;         // up to and including the first one that is not S/Z
    uint32_t packed = 0;
    for (int k = 0; k < 624; k++) {
        uint32_t u = mt_temper(mt[(size_t)k * stride]);
        if (redraw_open)
            for (int i = 0; i < 7; i++) w[(size_t)i * ws] = (float)(1000 / 7);   // randomizer.cpp:55-62
        int piece = map[pick_piece(w, ws, u)];
        if (redraw_open) {
            bool again = !only_sz && (piece == 2 || piece == 3) && k < 254;
            if (!again) { redraw_open = false; *first_ok = (uint8_t)k; }
        }
        uint32_t byte = (uint32_t)piece | ((uint32_t)pick_hole(u) << 4);
        packed |= byte << (8 * (k & 3));
        if ((k & 3) == 3) { ((uint32_t*)out)[k >> 2] = packed; packed = 0; }
    }
    // start entry of this seed: everything a reset reads, in one 8-byte load:
    //   first_ok j [0:8) | table byte of draw j [8:16) | table byte of draw j+1 [16:24) | piece group of draw j+2 [32:64)
    if (chunk == 0) {
        uint32_t j = first_ok_store;
        uint32_t lo = j | ((uint32_t)out[j] << 8) | ((uint32_t)out[j + 1] << 16);
        uint32_t hi = group_word(out + ((j + 2) & ~7u));
        *start_word = (uint64_t)lo | ((uint64_t)hi << 32);
    }
}

}  // namespace te
