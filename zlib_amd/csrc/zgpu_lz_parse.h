// zgpu_lz_parse.h -- K3, parallel form (device code; the kernel around it is in zgpu_lz_parse.hip, walk_kernel of zgpu_lz_sorted.hip
// calls it as well): the deflate_slow control flow (/root/reference/qcsrc/deflate.c:1554-1674) over the
// match records of a chunk, one 1024-lane workgroup per chunk, no serial walk over the positions.
//
// parse_kernel (zgpu_lz_parallel.hip) runs the reference's loop as it stands, one lane per chunk: ~25 000 dependent
// record loads per chunk, 42 ms however many chunks are in flight.  The loop has more structure than that:
//
//   * After every emitted match the state is the start state (no literal pending, prev_length = MIN_MATCH-1) at the
//     position behind the match.  Call such a position a Q0 position.
//   * From a Q0 position q the loop emits literals up to the first position r >= q whose record holds a match
//     (has(r): length >= 3 after the TOO_FAR rule, deflate.c:1597-1606), then plays the lazy-evaluation game from r: while the
//     record of the next position is longer than the match in hand, the byte is emitted as a literal and the longer match
//     taken (deflate.c:1611-1648).  The game depends on r alone -- not on how r was reached -- so E(r), the Q0 position it
//     ends in, and the match (start m, length, distance) it emits are functions of r.
//   * The window slide (`off`, deflate.c:1293) fires at the first VISITED position at or above a threshold that depends
//     on n and the chunk base only, so "has the slide happened" is a function of the position as well.
//
// So the has-positions form a forest, r -> nextHas(E(r)), and the parse is the path from nextHas(0).  The kernel
//   1. marks has(p) for all p (bitmap),
//   2. per window of 16384 positions: computes the successor of every has-position in the window (lanes = positions),
//      threads the path through the window by speculative walks of 256-position blocks (see 2b below), and hands the
//      path's exit to the next window,
//      then, still per window: every node on the path contributes one match token at m and covers (m, m+len); every
//      position not covered is a token (a literal, or the match at m) whose index is the prefix count of such positions,
//   3. derives the 16383-token block cuts and the "may not be stored" flags (trees.c:921-1016 via deflate.c
//      FLUSH_BLOCK_ONLY) from token indices.
// Output is identical to parse_kernel's: tokens, ntok, nostore, in_bytes.
#pragma once
#include "zgpu_common.h"

namespace zgpu {

#ifndef ZGPU_P2WIN
#define ZGPU_P2WIN 8192 // positions per window of the path threading
#endif
constexpr uint32_t kP2Win = ZGPU_P2WIN, kP2Blk = kP2Win / 64;
constexpr uint32_t kP2Words = kChunkMax / 32, kP2Batch = 8, kP2Pair = 4, kP2Over = 8; // positions per lane whose loads are in flight together; overhang of a wave's games
constexpr uint32_t kNone = 0xffffffffu;

struct ParseCtx {
    const uint2 *rec;
    uint32_t n, base, good, lazy, strategy;
    int slide_at; // visited positions >= slide_at see the slid window (off != 0)
    __device__ bool slid(uint32_t p) const { return (int)p >= slide_at; }
    // the match the loop takes at p when the match in hand has length prev_len (deflate.c:1585-1606); 2 = none
    __device__ uint32_t take(uint32_t p, uint32_t prev_len, uint2 r, uint32_t &dist) const
    {
        // (one straight line of selects: written with early returns this becomes a ladder of exec-mask branches in every caller)
        const uint32_t pick = (prev_len >= good && strategy != kRle) ? r.y : r.x; // (longest_match_fast has no chain to shorten)
        const uint32_t d = (pick >> 9) & 32767u;
        const bool nil = ((r.y >> 24) & 1u) && slid(p);                            // first candidate became NIL in the slide
        const uint32_t len = nil ? 0u : (pick & 511u);
        const bool weak = len <= 5 && (strategy == kFiltered || (len == kMinMatch && d > kTooFar)); // deflate.c:1601-1611
        const bool ok = prev_len < lazy && len > prev_len && !weak;
        dist = ok ? d : dist;
        return ok ? len : kMinMatch - 1;
    }
    // the lazy-evaluation game from a has-position r: match start m, length, distance.  rr = rec[r], rn = rec[r+1] (callers load
    // them in batches: one load latency per position would otherwise be the whole cost of this kernel)
    __device__ void game(uint32_t r, uint2 rr, uint2 rn, uint32_t &m, uint32_t &len, uint32_t &dist) const
    {
        uint32_t L, D = 0;
        L = take(r, kMinMatch - 1, rr, D);
        uint32_t q = r + 1;
        for (;;) { // a match of L >= 3 bytes at q-1 ends inside the chunk, so q <= n-2 has a record
            uint32_t D2 = 0;
            const uint32_t L2 = take(q, L, rn, D2);
            if (L2 <= L) break; // (take returns 2 when it keeps the match in hand)
            L = L2; D = D2; q++;
            rn = rec[q];
        }
        m = q - 1; len = L; dist = D;
    }
};

// lane i <- lane (i + 1) mod 64
__device__ inline uint32_t wave_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x134, 0xf, 0xf, false); }

__device__ inline uint32_t next_bit(const uint32_t *bits, uint32_t x, uint32_t nwords) // smallest set bit index >= x, or kNone
{
    uint32_t w = x >> 5;
    if (w >= nwords) return kNone;
    uint32_t v = bits[w] & (~0u << (x & 31u));
    while (v == 0) { if (++w >= nwords) return kNone; v = bits[w]; }
    return (w << 5) + (uint32_t)__builtin_ctz(v);
}

#if defined(ZGPU_P2_TIME) && !defined(ZGPU_PARSE_HEADER_ONLY) // debug build only (scripts/p2_time.py): cycles per phase, summed over workgroups (lane 0's clock)
__device__ unsigned long long p2_time[8];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_p2_time(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(p2_time), sizeof z);
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(p2_time), z, sizeof z);
}
// (summed in registers and written once at the end: an atomic per phase would sit in the memory counter and be waited for by the next load)
#define P2_T(i) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); t_acc[i] += t_ - t_prev; t_prev = t_; } } while (0)
#define P2_T0() unsigned long long t_prev = wall_clock64(), t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define P2_TEND() do { if (tid == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&p2_time[i_], t_acc[i_]); } while (0)
#else
#define P2_T(i) do { } while (0)
#define P2_T0() do { } while (0)
#define P2_TEND() do { } while (0)
#endif

// LITE: the games have been played by walk_kernel (zgpu_lz_sorted.hip): gmv[r] holds the game of every position r a walker stood on
// with nothing in hand and found a match at (bit r of gsv), in the format of gm[] below -- a subset of the has-positions that
// contains the whole path, which is all that stages A2..D look at.
template <bool FUSED> __device__ inline uint32_t p2_ld(const uint32_t *p) { return FUSED ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
// LDS of the parse when it is carved out of a block (walk_kernel): offsets in bytes
constexpr uint32_t kP2OffJ = 0, kP2OffHAS = kP2Win * 2, kP2OffMARK = kP2OffHAS + kP2Words * 4, kP2OffCOV = kP2OffMARK + kP2Words * 4, kP2OffMAT = kP2OffCOV + kP2Words * 4,
                   kP2OffWbase = kP2OffMAT + kP2Words * 4, kP2OffVIS = kP2OffWbase + (kP2Words + 4) * 4, kP2OffEXITS = kP2OffVIS + kP2Win / 32 * 4, kP2OffWtot = kP2OffEXITS + 256,
                   kP2OffEntry = kP2OffWtot + 64, kP2LdsBytes = kP2OffEntry + 16;

} // namespace zgpu
