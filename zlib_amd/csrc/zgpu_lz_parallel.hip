// zgpu_lz_parallel.hip -- LZ77 stage, parallel form for levels 4-9 (deflate_slow, /root/reference/qcsrc/deflate.c:1554-1674).
//
// At levels 4-9 the reference inserts *every* position p <= n-3 into its hash chains, whatever the parse decides
// (deflate.c:1579-1581 and 1633-1637), so the chain of a position is a pure function of the data:
//     link(p) = the largest q < p with the same 3-byte hash, or none.
// That splits longest_match + deflate_slow into three data-parallel kernels:
//
//   K1 chain_kernel   one wave per chunk: link(p) for all p, via a head table in LDS (64 KiB) updated in position
//                     order, 64 positions per step; duplicate hashes inside a step are resolved with ballots.
//   K2 match_kernel   one 1024-lane workgroup per chunk: the whole chunk (64 KiB) plus a ring of the most recent
//                     40960 links (80 KiB) live in LDS; every lane walks the chain of its own position exactly as
//                     longest_match does (deflate.c:1027-1168) with best_len seeded at MIN_MATCH-1 and records two
//                     results: after max_chain candidates and after max_chain>>2 candidates (the budget used when
//                     prev_length >= good_match, deflate.c:1064-1066).  A seeded call returns that result when it is
//                     longer than the seed and "no improvement" otherwise, so these two records answer every call
//                     the parser can make.
//   K3 parse_kernel   one lane per chunk: the deflate_slow control flow over the records (no chain walking, no byte
//                     compares), producing the token stream and the per-block "may not be stored" flags.
//
// Record of position p (2 x u32):  w0 = len_full | dist_full << 9 | byte(p) << 24,  w1 = len_quarter | dist_quarter << 9
// | flags << 24.  len 0 = no match.  Flag bit 0: the first chain candidate sits at window index 32768, which the
// reference turns into NIL when it slides the window (deflate.c:1309-1312); K3 applies it only once the slide has
// happened.
#include "zgpu_common.h"
#include "../../include/zamd_gpu.h"

namespace zgpu {

void prof_span_begin(void *eng, hipStream_t st, hipEvent_t *a);
void prof_span_end(void *eng, hipStream_t st, int stage, hipEvent_t a);

constexpr uint32_t kNoLink = 0xFFFFu;
constexpr uint32_t kTile = 8192, kRing = 40960; // ring >= MAX_DIST + tile: 40960 links = 80 KiB
constexpr uint32_t kMatchThreads = 1024;

struct ParWorkspace { uint16_t *links; uint2 *recs; };

size_t lz_parallel_workspace_bytes(uint32_t batch) { return (size_t)batch * kChunkMax * (sizeof(uint16_t) + sizeof(uint2)) + 256; }
bool lz_parallel_available() { return true; }

// ------------------------------------------------------------------------------------------------- K1
__global__ void __launch_bounds__(64) chain_kernel(ChunkGeom g, uint16_t *__restrict__ links)
{
    __shared__ uint16_t head[kHashSize];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    uint16_t *lk = links + (size_t)c * kChunkMax;
    for (uint32_t i = lane; i < kHashSize / 2; i += 64) reinterpret_cast<uint32_t *>(head)[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t npos = n >= 3 ? n - 2 : 0; // positions 0 .. n-3 carry a hash
    volatile uint16_t *vhead = head; // the claim/read-back below must really go through LDS
    for (uint32_t p0 = 0; p0 < npos; p0 += 64) {
        const uint32_t p = p0 + lane;
        const bool live = p < npos;
        uint32_t h = 0, old = kNoLink;
        if (live) { h = hash3(src[p], src[p + 1], src[p + 2]); old = vhead[h]; }
        uint32_t link = old;
        bool last = live; // the highest lane of a hash group leaves its position in head[]
        // claim the slot (LDS executes one wave's operations in order; when several lanes hit the same slot one of them
        // lands), then read it back: a lane that does not see itself shares its hash with another lane of this step
        if (live) vhead[h] = (uint16_t)p;
        const uint32_t seen = live ? (uint32_t)vhead[h] : p;
        unsigned long long clash = __ballot(live && seen != p);
        while (clash) {
            const int f = __ffsll((long long)clash) - 1;
            const uint32_t h0 = __shfl(h, f);
            const unsigned long long grp = __ballot(live && h == h0);
            if (live && h == h0) {
                const unsigned long long below = grp & ((1ull << lane) - 1);
                if (below) link = p0 + (63 - __clzll((long long)below)); // nearest lower lane with the same hash
                last = (grp >> lane) == 1ull;
            }
            clash &= ~grp;
        }
        if (live && last) vhead[h] = (uint16_t)p;
        if (live) lk[p] = (uint16_t)link;
    }
    for (uint32_t p = npos + lane; p < n; p += 64) lk[p] = (uint16_t)kNoLink;
}

// ------------------------------------------------------------------------------------------------- K2
// The ring keeps the links of the most recent kRing positions: slot(q) = q mod kRing (q < 65536 < 2*kRing).
__device__ inline uint32_t ring_slot(uint32_t q) { return q >= kRing ? q - kRing : q; }

// Lane states of the walk: every wave iteration advances each lane by one unit of work, either one chain
// candidate (quick reject + link hop) or one 8-byte piece of a full comparison, so the lanes of a wave never wait
// for one lane's inner loop.  Idle lanes are refilled with new positions in groups.
enum : uint32_t { kIdle = 0, kWalk = 1, kCmp = 2 };

__global__ void __launch_bounds__(kMatchThreads) match_kernel(ChunkGeom g, LevelCfg cfg, const uint16_t *__restrict__ links, uint2 *__restrict__ recs)
{
    extern __shared__ uint32_t lds[];
    uint32_t *d32 = lds;                                                       // 65536 + 64 bytes of chunk data
    uint16_t *ring = reinterpret_cast<uint16_t *>(lds + (kChunkMax + 64) / 4); // kRing links
    uint32_t *tile_next = lds + (kChunkMax + 64) / 4 + kRing / 2;              // work counter of the current tile
    const uint8_t *d8 = reinterpret_cast<const uint8_t *>(d32);

    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    const uint16_t *lk = links + (size_t)c * kChunkMax;
    uint2 *rec = recs + (size_t)c * kChunkMax;
    const uint32_t base = chunk_base(g, c);

    // stage the chunk (zero padded) in LDS
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *s128 = reinterpret_cast<const uint4 *>(src);
        uint4 *d128 = reinterpret_cast<uint4 *>(d32);
        const uint32_t nv = n >> 4;
        for (uint32_t i = tid; i < (kChunkMax + 64) / 16; i += kMatchThreads) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i < nv) v = s128[i];
            else if (i == nv) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < (n & 15); k++) w[k >> 2] |= (uint32_t)src[(nv << 4) + k] << (8 * (k & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            d128[i] = v;
        }
    } else {
        for (uint32_t i = tid; i < (kChunkMax + 64) / 4; i += kMatchThreads) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < 4; k++) { uint32_t a = (i << 2) + k; if (a < n) v |= (uint32_t)src[a] << (8 * k); }
            d32[i] = v;
        }
    }
    const uint32_t ntiles = (n + kTile - 1) / kTile;
    const uint32_t chainF = cfg.chain, chainQ = cfg.chain >> 2;

    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t tile_lo = t * kTile, tile_hi = tile_lo + kTile < n ? tile_lo + kTile : n;
        __syncthreads(); // the previous tile's walks are done: the slots this tile overwrites are dead
        {   // kTile links -> ring, 8 per lane (tile_lo is a multiple of 8 and so is its slot)
            const uint32_t q0 = tile_lo + tid * 8;
            uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (q0 + 8 <= n) v = *reinterpret_cast<const uint4 *>(lk + q0);
            else if (q0 < n) {
                uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                for (uint32_t k = 0; q0 + k < n; k++) w[k >> 1] = (w[k >> 1] & ~(0xFFFFu << (16 * (k & 1)))) | ((uint32_t)lk[q0 + k] << (16 * (k & 1)));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4 *>(ring + ring_slot(q0)) = v;
            if (tid == 0) *tile_next = tile_lo;
        }
        __syncthreads();

        // per-lane walk state
        uint32_t mode = kIdle, p = 0, q = 0, best = 0, bestq = 0, steps = 0, snap = 0, l = 0, cap = 0, nice = 0, flags = 0;
        uint32_t chk_off = 0, chk_mask = 0, chk_word = 0;
        int limit = 0;
        uint64_t scan0 = 0;
        uint32_t sup_next = 0, sup_end = 0; // this wave's private supply of positions (wave-uniform)
        bool tile_dry = false;

        for (;;) {
            const unsigned long long idle = __ballot(mode == kIdle);
            if (idle) {
                const uint32_t nidle = (uint32_t)__popcll(idle);
                if (sup_next == sup_end && !tile_dry) { // fetch another 128 positions for this wave
                    uint32_t got = 0;
                    if (lane == 0) got = atomicAdd(tile_next, 128u);
                    got = __shfl(got, 0);
                    if (got >= tile_hi) tile_dry = true;
                    else { sup_next = got; sup_end = got + 128 < tile_hi ? got + 128 : tile_hi; }
                }
                const bool have = sup_next != sup_end;
                if (!have && nidle == 64) break;
                if (have && (nidle >= 16 || nidle == 64)) {
                    if (mode == kIdle) {
                        const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1));
                        const uint32_t np = sup_next + rank;
                        if (np < sup_end) {
                            p = np;
                            const uint32_t look = n - p;
                            cap = look < kMaxMatch ? look : kMaxMatch;
                            nice = cfg.nice < look ? cfg.nice : look;
                            const int w = (int)(p + base);
                            limit = w > (int)kMaxDist ? w - (int)kMaxDist : 0;
                            q = ring[ring_slot(p)];
                            scan0 = *reinterpret_cast<const uint64_t *>(d8 + p);
                            const bool valid = look >= kMinMatch && q != kNoLink && (int)(q + base) > 0 && (uint32_t)(w - (int)(q + base)) <= kMaxDist;
                            flags = (valid && q + base == kWSize) ? 1u : 0u;
                            best = kMinMatch - 1; bestq = p; steps = 0; snap = 0;
                            chk_off = 0; chk_mask = 0x00FFFFFFu; chk_word = (uint32_t)scan0 & 0x00FFFFFFu;
                            if (valid) mode = kWalk;
                            else rec[p] = make_uint2((uint32_t)(scan0 & 0xFF) << 24, 0); // no candidate at all
                        }
                    }
                    const uint32_t take = nidle < sup_end - sup_next ? nidle : sup_end - sup_next;
                    sup_next += take;
                }
            }
            bool hop = false;
            if (mode == kCmp) {
                const uint64_t a = *reinterpret_cast<const uint64_t *>(d8 + q + l);
                const uint64_t b = l == 0 ? scan0 : *reinterpret_cast<const uint64_t *>(d8 + p + l);
                const uint64_t x = a ^ b;
                if (x == 0 && l + 8 < cap) l += 8;
                else {
                    uint32_t len = x ? l + ((uint32_t)__builtin_ctzll(x) >> 3) : l + 8;
                    len = len < cap ? len : cap;
                    if (len > best) {
                        best = len; bestq = q;
                        chk_off = best - 3; chk_mask = 0xFFFFFFFFu; // bytes best-3 .. best must match for a longer match
                        chk_word = *reinterpret_cast<const uint32_t *>(d8 + p + chk_off);
                    }
                    mode = kWalk; hop = true;
                }
            } else if (mode == kWalk) {
                const uint32_t cw = *reinterpret_cast<const uint32_t *>(d8 + q + chk_off) & chk_mask;
                if (cw == chk_word) { mode = kCmp; l = 0; } // candidate may be longer: compare it in full
                else hop = true;
            }
            if (hop) { // candidate q is dealt with: account for it and move to the next one (deflate.c:1152-1164)
                steps++;
                if (steps == chainQ) snap = best | ((p - bestq) << 9);
                bool stop = best >= nice || steps == chainF;
                const uint32_t nq = ring[ring_slot(q)];
                stop = stop || nq == kNoLink || (int)(nq + base) <= limit;
                q = nq;
                if (stop) {
                    uint32_t full = best | ((p - bestq) << 9);
                    if (steps < chainQ) snap = full;
                    if ((full & 511) < kMinMatch) full = 0;
                    if ((snap & 511) < kMinMatch) snap = 0;
                    rec[p] = make_uint2(full | ((uint32_t)(scan0 & 0xFF) << 24), snap | (flags << 24));
                    mode = kIdle;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- K3
__global__ void __launch_bounds__(64) parse_kernel(ChunkGeom g, LevelCfg cfg, const uint2 *__restrict__ recs, uint32_t *__restrict__ tokens,
                                                   ChunkMeta *meta)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.nchunks) return;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint2 *rec = recs + (size_t)c * kChunkMax;
    uint32_t *tok = tokens + (size_t)c * kChunkMax;
    const uint32_t base = chunk_base(g, c);
    const uint32_t room = 2 * kWSize - base;
    uint32_t buffered = n < room ? n : room, off = 0;
    uint32_t p = 0, match_len = kMinMatch - 1, cur_dist = 0, prev_len, prev_dist, prev_byte = 0;
    uint32_t ntok = 0, blk_tok0 = 0, nblk = 0, nostore = 0, block_start = 0;
    bool pending = false;
    for (;;) {
        if (buffered - p < kMinLookahead) {
            if ((int)(p + base) - (int)off >= (int)(kWSize + kMaxDist)) off += kWSize; // the slide, deflate.c:1293
            buffered = n;
            if (p == n) break;
        }
        const uint2 r = rec[p];
        prev_len = match_len; prev_dist = cur_dist; match_len = kMinMatch - 1;
        if (prev_len < cfg.lazy) {
            const uint32_t pick = prev_len >= cfg.good ? r.y : r.x;
            uint32_t len = pick & 511, dist = (pick >> 9) & 32767;
            if (off != 0 && ((r.y >> 24) & 1)) len = 0; // first candidate became NIL in the slide
            if (len > prev_len) {
                match_len = len; cur_dist = dist;
                if (match_len == kMinMatch && cur_dist > kTooFar) match_len = kMinMatch - 1;
            }
        }
        bool cut = false;
        if (prev_len >= kMinMatch && match_len <= prev_len) {
            tok[ntok++] = tok_match(prev_dist, prev_len - kMinMatch);
            cut = ntok - blk_tok0 == kBlockTokens;
            p += prev_len - 1;
            pending = false; match_len = kMinMatch - 1;
        } else if (pending) {
            tok[ntok++] = tok_lit(prev_byte);
            cut = ntok - blk_tok0 == kBlockTokens;
            if (cut) { // FLUSH_BLOCK_ONLY happens before strstart++ (deflate.c:1651-1654)
                if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk;
                nblk++; blk_tok0 = ntok; block_start = p; cut = false;
            }
            p++;
        } else { pending = true; p++; }
        prev_byte = r.x >> 24;
        if (cut) {
            if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk;
            nblk++; blk_tok0 = ntok; block_start = p;
        }
    }
    if (pending) tok[ntok++] = tok_lit(prev_byte);
    if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk;
    meta[c].ntok = ntok; meta[c].nostore = nostore; meta[c].in_bytes = n;
}

void launch_lz_parallel(const ChunkGeom &g, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, void *prof)
{
    uint16_t *links = static_cast<uint16_t *>(workspace);
    // records follow the links; links take batch*65536*2 bytes, the caller sized the workspace for its batch capacity
    // (the record array starts at a 256-byte aligned offset computed from nchunks of *this* launch's capacity owner)
    uint2 *recs = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(workspace) + (((size_t)g.nchunks * kChunkMax * sizeof(uint16_t) + 255) & ~(size_t)255));
    hipEvent_t ev{};
    prof_span_begin(prof, st, &ev);
    hipLaunchKernelGGL(chain_kernel, dim3(g.nchunks), dim3(64), 0, st, g, links);
    prof_span_end(prof, st, ZGPU_STAGE_CHAIN, ev);
    prof_span_begin(prof, st, &ev);
    const size_t lds_bytes = (kChunkMax + 64) + kRing * sizeof(uint16_t) + 64;
    static bool lds_opt_in = false; // > 64 KiB of dynamic LDS needs an explicit opt-in
    if (!lds_opt_in) { hipFuncSetAttribute(reinterpret_cast<const void *>(match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); lds_opt_in = true; }
    hipLaunchKernelGGL(match_kernel, dim3(g.nchunks), dim3(kMatchThreads), lds_bytes, st, g, cfg, links, recs);
    prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
    prof_span_begin(prof, st, &ev);
    hipLaunchKernelGGL(parse_kernel, dim3((g.nchunks + 63) / 64), dim3(64), 0, st, g, cfg, recs, tokens, meta);
    prof_span_end(prof, st, ZGPU_STAGE_PARSE, ev);
}

} // namespace zgpu
