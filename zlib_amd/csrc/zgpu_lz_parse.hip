// zgpu_lz_parse.hip -- K3, parallel form: the deflate_slow control flow (/root/reference/qcsrc/deflate.c:1554-1674) over the
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
#include "zgpu_common.h"

namespace zgpu {

#ifndef ZGPU_P2WIN
#define ZGPU_P2WIN 8192 // positions per window of the path threading
#endif
constexpr uint32_t kP2Threads = 1024, kP2Win = ZGPU_P2WIN, kP2Own = kP2Win / kP2Threads, kP2Blk = kP2Win / 64;
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

#ifdef ZGPU_P2_TIME // debug build only (scripts/p2_time.py): cycles per phase, summed over workgroups (lane 0's clock)
__device__ unsigned long long p2_time[8];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_p2_time(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(p2_time), sizeof z);
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(p2_time), z, sizeof z);
}
#define P2_T(i) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&p2_time[i], t_ - t_prev); t_prev = t_; } } while (0)
#define P2_T0() unsigned long long t_prev = wall_clock64()
#else
#define P2_T(i) do { } while (0)
#define P2_T0() do { } while (0)
#endif

// LITE: the games have been played by walk_kernel (zgpu_lz_sorted.hip): gmv[r] holds the game of every position r a walker stood on
// with nothing in hand and found a match at (bit r of gsv), in the format of gm[] below -- a subset of the has-positions that
// contains the whole path, which is all that stages A2..D look at.
template <bool LITE>
__global__ void __launch_bounds__(kP2Threads, 8) parse2_kernel(ChunkGeom g, LevelCfg cfg, const uint2 *__restrict__ recs, const uint32_t *__restrict__ gmv_all,
                                                               const uint32_t *__restrict__ gsv_all, uint32_t *__restrict__ tokens, ChunkMeta *meta)
{
    __shared__ __attribute__((aligned(16))) uint16_t J[kP2Win]; // successor of a has-position, window-relative (0xffff: leaves the window)
    __shared__ uint32_t HAS[kP2Words], MARK[kP2Words], COV[kP2Words], MAT[kP2Words]; // per position: has a match / on the path / inside a match
                                                                                    // (later: is a token) / starts an emitted match
    __shared__ uint32_t wbase[kP2Words + 1];                   // tokens in front of each 32-position word
    __shared__ uint32_t VIS[kP2Win / 32], EXITS[64];
    __shared__ uint32_t wave_tot[kP2Threads / 64];
    __shared__ uint32_t sh_entry;
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    ParseCtx cx;
    cx.rec = LITE ? nullptr : recs + (size_t)c * kChunkMax;
    const uint32_t *gmv = LITE ? gmv_all + (size_t)c * kChunkMax : nullptr, *gsv = LITE ? gsv_all + (size_t)c * kP2Words : nullptr;
    cx.n = n; cx.base = chunk_base(g, c); cx.good = cfg.good; cx.lazy = cfg.lazy; cx.strategy = cfg.strategy;
    {
        // deflate.c:1278-1310 as a function of the position: the first check of "lookahead < MIN_LOOKAHEAD" happens at the first
        // visited p > buffered0 - 262, later ones at p > n - 262; the slide needs p + base >= WSIZE + MAX_DIST on top
        const int room = (int)(2 * kWSize - cx.base), b0 = (int)n < room ? (int)n : room;
        const int a = b0 - (int)kMinLookahead + 1, b = (int)(kWSize + kMaxDist) - (int)cx.base;
        cx.slide_at = a > b ? a : b;
    }
    uint32_t *tok = tokens + (size_t)c * kChunkMax;
    const uint8_t *src = g.in + lo;
    const uint32_t nwords = (n + 31) >> 5;

    P2_T0();
    // ---- 1. has(p) ----
    for (uint32_t i = tid; i < kP2Words; i += kP2Threads) { MARK[i] = 0; COV[i] = 0; MAT[i] = 0; if (LITE) HAS[i] = gsv[i]; }
    if (!LITE) for (uint32_t p0 = 0; p0 < kChunkMax; p0 += kP2Threads * kP2Batch) {
        uint2 rv[kP2Batch];
#pragma unroll
        for (uint32_t u = 0; u < kP2Batch; u++) { const uint32_t p = p0 + u * kP2Threads + tid; rv[u] = p < n ? cx.rec[p] : make_uint2(0, 0); }
#pragma unroll
        for (uint32_t u = 0; u < kP2Batch; u++) {
            const uint32_t pw = p0 + u * kP2Threads, p = pw + tid;
            uint32_t d;
            const bool h = p < n && cx.take(p, kMinMatch - 1, rv[u], d) >= kMinMatch;
            const unsigned long long b = __builtin_amdgcn_ballot_w64(h);
            if (lane == 0) { HAS[(pw >> 5) + 2 * wave] = (uint32_t)b; HAS[(pw >> 5) + 2 * wave + 1] = (uint32_t)(b >> 32); }
        }
    }
    __syncthreads();
    if (tid == 0) sh_entry = next_bit(HAS, 0, nwords);
    __syncthreads();
    P2_T(0);

    const uint32_t *TOK = COV; // a window's COV words turn into its token words in step C
    auto index_of = [&](uint32_t p) { return wbase[p >> 5] + (uint32_t)__builtin_popcount(TOK[p >> 5] & ~(~0u << (p & 31u))); };
    uint32_t tokbase = 0;                   // tokens in front of the window (uniform)
    uint32_t def_m = kNone, def_tok = 0;    // a match token whose position lies in the next window

    // ---- 2. window by window: path, matches, token indices, tokens ----
    for (uint32_t w0 = 0; w0 < n; w0 += kP2Win) {
        const uint32_t wend = w0 + kP2Win;
        const uint32_t entry = sh_entry;
        uint32_t gm[kP2Own]; // the game of this lane's has-positions: (m - p) << 24 | len << 15 | dist; 0: none
#pragma unroll
        for (uint32_t i = 0; i < kP2Own; i++) gm[i] = 0;
        if (entry != kNone && entry < wend) { // (uniform) the path has nodes in this window
            // A1. successors of all has-positions of the window.  A lane's game reads the records of the positions right behind
            // its own: those are its neighbours' records (lanes = consecutive positions), fetched with lane shuffles; the first
            // kP2Over positions behind the wave's 64 are loaded by lanes 0..kP2Over-1 as well.  (A dependent global load per step of
            // the game -- some lane of the wave always needs one -- was 45% of this kernel.)
            if (LITE) {
                uint32_t gv[kP2Own];
#pragma unroll
                for (uint32_t i = 0; i < kP2Own; i++) {
                    const uint32_t p = w0 + i * kP2Threads + tid;
                    gv[i] = p < n ? gmv[p] : 0; // (all positions, whatever the words of the others hold: loads under a per-lane condition are waited for one by one)
                }
#pragma unroll
                for (uint32_t i = 0; i < kP2Own; i++) {
                    const uint32_t p = w0 + i * kP2Threads + tid;
                    if (!(p < n && ((HAS[p >> 5] >> (p & 31u)) & 1u))) gv[i] = 0;
                }
#pragma unroll
                for (uint32_t i = 0; i < kP2Own; i++) {
                    const uint32_t xw = i * kP2Threads + tid, p = w0 + xw;
                    uint32_t succ = 0xffffu;
                    gm[i] = gv[i];
                    if (gv[i]) {
                        const uint32_t t = next_bit(HAS, p + (gv[i] >> 24) + ((gv[i] >> 15) & 511u), nwords);
                        if (t < wend) succ = t - w0;
                    }
                    J[xw] = (uint16_t)succ;
                }
            } else {
#pragma unroll
            for (uint32_t ib = 0; ib < kP2Own; ib += kP2Pair) {
                uint2 ra[kP2Pair], rx[kP2Pair];
#pragma unroll
                for (uint32_t u = 0; u < kP2Pair; u++) {
                    const uint32_t p = w0 + (ib + u) * kP2Threads + tid, px = p + 64; // the overhang: positions 64.. behind the wave's first, one per low lane
                    ra[u] = p < n ? cx.rec[p] : make_uint2(0, 0);
                    rx[u] = (lane < kP2Over && px < n) ? cx.rec[px] : make_uint2(0, 0);
                }
#pragma unroll
                for (uint32_t u = 0; u < kP2Pair; u++) {
                    const uint32_t x = (ib + u) * kP2Threads + tid, p = w0 + x;
                    const bool hs = p < n && ((HAS[p >> 5] >> (p & 31u)) & 1u);
                    uint32_t L = kMinMatch - 1, D = 0, j = 1; // the match in hand starts at p + j - 1
                    if (hs) L = cx.take(p, kMinMatch - 1, ra[u], D);
                    bool live = hs;
                    // the games of a wave advance in step (j is the same in every lane that is still playing), so the records
                    // they need next are the wave's records rotated by one more lane each time: a DPP rotate per register, the
                    // overhang entering at lane 63 (a lane shuffle through the LDS crossbar per step was the cost of this loop)
                    uint32_t qx = ra[u].x, qy = ra[u].y, ox = rx[u].x, oy = rx[u].y;
                    while (__builtin_amdgcn_ballot_w64(live)) {
                        const uint32_t sl = lane + j;
                        const uint32_t tx = wave_rol1(qx), ty = wave_rol1(qy);
                        ox = wave_rol1(ox); oy = wave_rol1(oy);
                        qx = lane == 63 ? ox : tx; qy = lane == 63 ? oy : ty;
                        uint2 rn;
                        rn.x = qx; rn.y = qy;
                        const bool far = live && sl >= 64 + kP2Over;
                        if (__builtin_amdgcn_ballot_w64(far)) { if (far) rn = cx.rec[p + j]; } // a game that long is rare
                        uint32_t D2 = D;
                        const uint32_t L2 = cx.take(p + j, L, rn, D2); // (pure arithmetic: every lane evaluates it, the ones in a game use it)
                        live = live && L2 > L;
                        L = live ? L2 : L; D = live ? D2 : D; j += live ? 1u : 0u;
                    }
                    uint32_t succ = 0xffffu;
                    if (hs) {
                        const uint32_t m = p + j - 1;
                        gm[ib + u] = ((m - p) << 24) | (L << 15) | D;
                        const uint32_t t = next_bit(HAS, m + L, nwords);
                        if (t < wend) succ = t - w0;
                    }
                    J[x] = (uint16_t)succ;
                }
            }
            }
            __syncthreads();
            P2_T(1);
            // A2. The path through the window, without walking it end to end: parses started at different positions fall into
            // step with each other after a match or two, so every block of 256 positions is walked speculatively from its first
            // has-position (64 lanes, ~30 dependent hops each), and the true path is then threaded through the blocks: it is
            // walked from where it enters a block until it meets the block's speculative walk; from there on that walk IS the path.
            if (wave == 0) {
                const uint32_t bs = w0 + lane * kP2Blk, be = bs + kP2Blk;
                uint32_t v8[kP2Blk / 32];
#pragma unroll
                for (uint32_t j = 0; j < kP2Blk / 32; j++) v8[j] = 0;
                uint32_t x = next_bit(HAS, bs, nwords); // (an empty block's exit: the first node behind it)
                if (x != kNone && x >= wend) x = kNone;
                while (x != kNone && x < be) {
                    const uint32_t wi = (x - bs) >> 5, bit = 1u << (x & 31u);
#pragma unroll
                    for (uint32_t j = 0; j < kP2Blk / 32; j++) v8[j] |= wi == j ? bit : 0u;
                    const uint32_t a2 = J[x - w0];
                    x = a2 == 0xffffu ? kNone : w0 + a2;
                }
#pragma unroll
                for (uint32_t j = 0; j < kP2Blk / 32; j++) VIS[lane * (kP2Blk / 32) + j] = v8[j];
                EXITS[lane] = x; // where the block's walk leaves the block (kNone: leaves the window)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // other lanes' VIS and EXITS (LDS, same wave: in order)
                __builtin_amdgcn_wave_barrier();
                // Threading, optimistic form: assume the path meets every block's speculative walk inside the block.  Then the blocks
                // it visits follow from the exits alone (a walk over at most 64 lanes' registers), every visited block finds its
                // meeting point on its own lane, and the assumption is checked block by block.
                uint32_t last = entry;
                bool ok;
                {
                    const uint32_t myexit = x;
                    uint32_t myentry = kNone;              // where the path enters this lane's block
                    uint32_t cb = (entry - w0) / kP2Blk, ce = entry, lastb = cb;
                    for (;;) {                             // wave-uniform, no memory
                        if (lane == cb) myentry = ce;
                        lastb = cb;
                        ce = (uint32_t)__builtin_amdgcn_readlane((int)myexit, (int)cb);
                        if (ce == kNone) break;
                        cb = (ce - w0) / kP2Blk;
                    }
                    // each visited block: from its entry to the meeting point
                    uint32_t m8[kP2Blk / 32];
#pragma unroll
                    for (uint32_t j = 0; j < kP2Blk / 32; j++) m8[j] = 0;
                    uint32_t x2 = myentry;
                    bool met = false;
                    while (x2 != kNone && x2 < be) {
                        const uint32_t wi = (x2 - bs) >> 5, bit = 1u << (x2 & 31u);
                        uint32_t vw = 0;
#pragma unroll
                        for (uint32_t j = 0; j < kP2Blk / 32; j++) vw = wi == j ? v8[j] : vw;
                        if (vw & bit) { // met: the rest of the speculative walk is the path
#pragma unroll
                            for (uint32_t j = 0; j < kP2Blk / 32; j++) m8[j] |= j > wi ? v8[j] : j == wi ? v8[j] & (~0u << (x2 & 31u)) : 0u;
                            met = true;
                            break;
                        }
#pragma unroll
                        for (uint32_t j = 0; j < kP2Blk / 32; j++) m8[j] |= wi == j ? bit : 0u;
                        const uint32_t a2 = J[x2 - w0];
                        x2 = a2 == 0xffffu ? kNone : w0 + a2;
                    }
                    // a block the path ran through without meeting the walk must at least leave it where the walk does
                    const bool bad = myentry != kNone && !met && x2 != myexit;
                    ok = __builtin_amdgcn_ballot_w64(bad) == 0;
                    if (ok) {
#pragma unroll
                        for (uint32_t j = 0; j < kP2Blk / 32; j++) if (m8[j]) atomicOr(&MARK[(w0 >> 5) + lane * (kP2Blk / 32) + j], m8[j]);
                        uint32_t hi = 0; // 1 + the last path node of this block (window-relative)
#pragma unroll
                        for (uint32_t j = 0; j < kP2Blk / 32; j++) if (m8[j]) hi = ((lane * (kP2Blk / 32) + j) << 5) + 32u - (uint32_t)__builtin_clz(m8[j]);
                        last = w0 + (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)lastb) - 1;
                    }
                }
                if (!ok) { // (rare) thread the path block by block, walking until it meets the block's speculative walk
                    uint32_t cur = entry, met_blk = kNone; // wave-uniform
                    last = entry;
                    while (cur != kNone) {
                        const uint32_t blk = (cur - w0) / kP2Blk, be2 = w0 + (blk + 1) * kP2Blk;
                        uint32_t x2 = cur;
                        met_blk = kNone;
                        while (x2 != kNone && x2 < be2 && !((VIS[(x2 - w0) >> 5] >> (x2 & 31u)) & 1u)) {
                            last = x2;
                            if (lane == 0) atomicOr(&MARK[x2 >> 5], 1u << (x2 & 31u));
                            const uint32_t a2 = J[x2 - w0];
                            x2 = a2 == 0xffffu ? kNone : w0 + a2;
                        }
                        if (x2 != kNone && x2 < be2) { // met at x2
                            const uint32_t cw = (x2 - w0) >> 5;
                            if (lane < kP2Blk / 32) {
                                const uint32_t wd = blk * (kP2Blk / 32) + lane;
                                uint32_t v = VIS[wd];
                                if (wd < cw) v = 0; else if (wd == cw) v &= ~0u << (x2 & 31u);
                                if (v) atomicOr(&MARK[(w0 >> 5) + wd], v);
                            }
                            met_blk = blk;
                            x2 = EXITS[blk];
                        }
                        cur = (uint32_t)__builtin_amdgcn_readfirstlane(x2);
                        last = (uint32_t)__builtin_amdgcn_readfirstlane(last);
                        met_blk = (uint32_t)__builtin_amdgcn_readfirstlane(met_blk);
                    }
                    if (met_blk != kNone) { // the path ended inside a speculative walk: its last node is that block's last visited one
                        uint32_t hi = 0;
                        for (uint32_t j = 0; j < kP2Blk / 32; j++) { const uint32_t v = VIS[met_blk * (kP2Blk / 32) + j]; if (v) hi = ((met_blk * (kP2Blk / 32) + j) << 5) + 32u - (uint32_t)__builtin_clz(v); }
                        last = w0 + hi - 1;
                    }
                }
                if (lane == 0) { // the last path node of the window leads to the entry of the next one
                    uint32_t m, L, D;
                    if (LITE) { const uint32_t gv = gmv[last]; m = last + (gv >> 24); L = (gv >> 15) & 511u; D = gv & 32767u; }
                    else cx.game(last, cx.rec[last], cx.rec[last + 1], m, L, D);
                    sh_entry = next_bit(HAS, m + L, nwords);
                }
            }
            __syncthreads();
            P2_T(2);
            // B. matches of the path: MAT bit at the match start, COV bits on the bytes behind it (they may reach into the next window)
#pragma unroll
            for (uint32_t i = 0; i < kP2Own; i++) {
                const uint32_t p = w0 + i * kP2Threads + tid;
                if (gm[i] && ((MARK[p >> 5] >> (p & 31u)) & 1u)) {
                    const uint32_t m = p + (gm[i] >> 24), L = (gm[i] >> 15) & 511u;
                    atomicOr(&MAT[m >> 5], 1u << (m & 31u));
                    const uint32_t a = m + 1, z = m + L; // [a, z)
                    for (uint32_t wd = a >> 5; wd <= (z - 1) >> 5; wd++) {
                        const uint32_t lo_b = wd == (a >> 5) ? (a & 31u) : 0u, hi_b = wd == ((z - 1) >> 5) ? ((z - 1) & 31u) : 31u;
                        atomicOr(&COV[wd], (~0u << lo_b) & (~0u >> (31u - hi_b)));
                    }
                } else gm[i] = 0;
            }
        }
        __syncthreads();
        P2_T(3);
        // C. token positions of the window = positions not inside a match; exclusive prefix counts per word (512 words: lanes 0..511)
        {
            const uint32_t wd = (w0 >> 5) + tid;
            uint32_t tw = 0;
            if (tid < kP2Win / 32 && wd < nwords) {
                tw = ~COV[wd];
                if (wd == nwords - 1 && (n & 31u)) tw &= ~0u >> (32u - (n & 31u));
            }
            const uint32_t cnt = (uint32_t)__builtin_popcount(tw);
            uint32_t x = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
            if (lane == 63) wave_tot[wave] = x;
            __syncthreads();
            uint32_t b = tokbase + x - cnt, tot = 0;
            for (uint32_t w = 0; w < kP2Win / 32 / 64; w++) { const uint32_t t = wave_tot[w]; tot += t; if (w < wave) b += t; }
            if (tid < kP2Win / 32) { wbase[wd] = b; COV[wd] = tw; } // (wd < kP2Words always; words past nwords hold no tokens)
            tokbase += tot;
            __syncthreads();
        }
        P2_T(4);
        // D. tokens of the window
        if (def_m != kNone) { tok[index_of(def_m)] = def_tok; def_m = kNone; }
#pragma unroll
        for (uint32_t ib = 0; ib < kP2Own; ib += kP2Batch) {
            uint32_t lit[kP2Batch];
            bool li[kP2Batch];
#pragma unroll
            for (uint32_t u = 0; u < kP2Batch; u++) {
                const uint32_t p = w0 + (ib + u) * kP2Threads + tid, bit = 1u << (p & 31u);
                li[u] = p < n && (TOK[p >> 5] & bit) && !(MAT[p >> 5] & bit);
                lit[u] = 0;
                if (li[u]) lit[u] = src[p];
            }
#pragma unroll
            for (uint32_t u = 0; u < kP2Batch; u++) {
                const uint32_t p = w0 + (ib + u) * kP2Threads + tid;
                if (li[u]) tok[index_of(p)] = tok_lit(lit[u]);
                const uint32_t gv = gm[ib + u];
                if (gv) {
                    const uint32_t m = p + (gv >> 24), t = tok_match(gv & 32767u, ((gv >> 15) & 511u) - kMinMatch);
                    if (m < wend) tok[index_of(m)] = t; else { def_m = m; def_tok = t; } // (at most one per lane: m - p < 1024)
                }
            }
        }
    }
    P2_T(5);
    const uint32_t ntok = tokbase, nw_done = ((n + kP2Win - 1) / kP2Win) * (kP2Win / 32); // words that have a wbase entry
    __syncthreads();

    // ---- 3. block cuts (deflate.c:1620-1626, 1651-1656: after the 16383rd token of a block) and the stored-block veto ----
    if (tid == 0) {
        uint32_t nostore = 0, nblk = 0, block_start = 0;
        for (uint32_t b = 0;; b++) {
            const uint32_t e = (b + 1) * kBlockTokens - 1; // index of the token that fills block b
            if (e >= ntok) break;
            uint32_t lo_w = 0, hi_w = nw_done; // the word holding token e: wbase[lo_w] <= e < wbase[lo_w + 1]
            while (hi_w - lo_w > 1) { const uint32_t mid = (lo_w + hi_w) >> 1; if (wbase[mid] <= e) lo_w = mid; else hi_w = mid; }
            uint32_t v = TOK[lo_w];
            for (uint32_t s2 = e - wbase[lo_w]; s2; s2--) v &= v - 1;
            const uint32_t pe = (lo_w << 5) + (uint32_t)__builtin_ctz(v);
            const bool is_mat = (MAT[pe >> 5] >> (pe & 31u)) & 1u;
            if (!is_mat && pe == n - 1) break; // the last byte's literal is emitted behind the loop (deflate.c:1660-1665): no cut
            // the token is emitted while the loop stands at pe+1; a match leaves it at the next token position
            uint32_t ns = pe + 1;
            if (is_mat) { ns = next_bit(TOK, pe + 1, nwords); if (ns == kNone) ns = n; }
            if (cx.slid(pe + 1) && block_start + cx.base < kWSize) nostore |= 1u << nblk;
            nblk++; block_start = ns;
        }
        if ((int)n >= (int)(kWSize + kMaxDist) - (int)cx.base && block_start + cx.base < kWSize) nostore |= 1u << nblk; // the check at p == n slides too
        meta[c].ntok = ntok; meta[c].nostore = nostore; meta[c].in_bytes = n;
    }
    P2_T(6);
}

void launch_parse2(const ChunkGeom &g, LevelCfg cfg, const uint2 *recs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st)
{
    hipLaunchKernelGGL(parse2_kernel<false>, dim3(g.nchunks), dim3(kP2Threads), 0, st, g, cfg, recs, nullptr, nullptr, tokens, meta);
}

void launch_parse_lite(const ChunkGeom &g, LevelCfg cfg, const uint32_t *gm, const uint32_t *gs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st)
{
    hipLaunchKernelGGL(parse2_kernel<true>, dim3(g.nchunks), dim3(kP2Threads), 0, st, g, cfg, nullptr, gm, gs, tokens, meta);
}

} // namespace zgpu
