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
#include <cstdlib>
#include "../../include/zamd_gpu.h"

namespace zgpu {

void prof_span_begin(void *eng, hipStream_t st, hipEvent_t *a);
void prof_span_end(void *eng, hipStream_t st, int stage, hipEvent_t a);

constexpr uint32_t kNoLink = 0; // links are stored as q+1 (q = position of the previous same-hash string), 0 = none
constexpr uint32_t kTile = 8192, kRing = 40960; // ring >= MAX_DIST + tile: 40960 links = 80 KiB
constexpr uint32_t kMatchThreads = 1024;
constexpr int kSlots = 1; // positions a lane walks concurrently (1 or 2)
#ifndef ZGPU_REFILL_MIN
#define ZGPU_REFILL_MIN 16
#endif
#ifndef ZGPU_CMP_MIN
#define ZGPU_CMP_MIN 16
#endif
#ifndef ZGPU_WALK_UNROLL
#define ZGPU_WALK_UNROLL 8
#endif

struct ParWorkspace { uint16_t *links; uint2 *recs; };

size_t lz_parallel_workspace_bytes(uint32_t batch) { return (size_t)batch * kChunkMax * (sizeof(uint16_t) + sizeof(uint2)) + 256; }
bool lz_parallel_available() { return true; }

// ------------------------------------------------------------------------------------------------- K1
// One wave per chunk.  head[] (LDS, 64 KiB) holds p+1 of the latest position of each hash, 0 = empty, and is updated in
// position order, 64 positions per step.  Global memory is touched once per 1024 positions only: the next 1 KiB of input
// is fetched into registers while the current one is processed, and the links of the previous 1 KiB are written out from
// an LDS staging buffer -- so the single vmcnt wait per kilobyte finds its loads and stores long finished (per-step
// stores made every step wait for the previous step's stores: loads and stores share one counter on gfx9).
constexpr uint32_t kSuper = 1024; // positions per superblock

__global__ void __launch_bounds__(64) chain_kernel(ChunkGeom g, uint16_t *__restrict__ links)
{
    __shared__ uint16_t head[kHashSize];
    __shared__ __attribute__((aligned(16))) uint32_t in_stage[kSuper / 4 + 4]; // 1 KiB of input + 8 bytes of the next KiB
    __shared__ __attribute__((aligned(16))) uint16_t link_stage[kSuper];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    uint16_t *lk = links + (size_t)c * kChunkMax;
    for (uint32_t i = lane; i < kHashSize / 2; i += 64) reinterpret_cast<uint32_t *>(head)[i] = 0;
    const uint32_t npos = n >= 3 ? n - 2 : 0; // positions 0 .. n-3 carry a hash
    volatile uint16_t *vhead = head;          // the claim/read-back below must really go through LDS
    const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    auto fetch = [&](uint32_t sb) -> uint4 { // 16 bytes: bytes [sb*1024 + lane*16, +16) of the chunk, zero padded past n
        const uint32_t a = sb * kSuper + lane * 16;
        if (a + 16 <= n && aligned) return *reinterpret_cast<const uint4 *>(src + a);
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < 16; k++) if (a + k < n) w[k >> 2] |= (uint32_t)src[a + k] << (8 * (k & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    const uint32_t nsuper = (n + kSuper - 1) / kSuper;
    uint4 cur = fetch(0), nxt = fetch(1);
    __syncthreads();
    const uint8_t *s8 = reinterpret_cast<const uint8_t *>(in_stage);
    for (uint32_t sb = 0; sb < nsuper; sb++) {
        // input of this superblock -> LDS (the registers were loaded one superblock ago)
        reinterpret_cast<uint4 *>(in_stage)[lane] = cur;
        if (lane == 0) { in_stage[kSuper / 4] = nxt.x; in_stage[kSuper / 4 + 1] = nxt.y; }
        // links of the previous superblock -> global, 32 bytes per lane
        if (sb > 0) {
            const uint32_t p0 = (sb - 1) * kSuper + lane * 16;
            const uint4 *ls = reinterpret_cast<const uint4 *>(link_stage);
            if (p0 + 16 <= n) { uint4 *dst = reinterpret_cast<uint4 *>(lk + p0); dst[0] = ls[lane * 2]; dst[1] = ls[lane * 2 + 1]; }
            else for (uint32_t k = 0; k < 16; k++) if (p0 + k < n) lk[p0 + k] = link_stage[lane * 16 + k];
        }
        cur = nxt;
        nxt = fetch(sb + 2);
        __syncthreads();
        const uint32_t base_p = sb * kSuper;
#pragma unroll 1
        for (uint32_t st = 0; st < kSuper / 64; st++) {
            const uint32_t o = st * 64 + lane, p = base_p + o, p0 = base_p + st * 64;
            const bool live = p < npos;
            uint32_t h = 0, old = kNoLink;
            if (live) { h = hash3(s8[o], s8[o + 1], s8[o + 2]); old = vhead[h]; }
            uint32_t link = old;
            bool last = live; // the highest lane of a hash group leaves its position in head[]
            // claim the slot (LDS executes one wave's operations in order; when several lanes hit the same slot one of
            // them lands), then read it back: a lane that does not see itself shares its hash with another lane
            if (live) vhead[h] = (uint16_t)(p + 1);
            const uint32_t seen = live ? (uint32_t)vhead[h] : p + 1;
            unsigned long long clash = __ballot(live && seen != p + 1);
            while (clash) {
                const int f = __ffsll((long long)clash) - 1;
                const uint32_t h0 = __builtin_amdgcn_readlane(h, f); // f is wave-uniform: v_readlane, no LDS round trip
                const unsigned long long grp = __ballot(live && h == h0);
                if (live && h == h0) {
                    const unsigned long long below = grp & ((1ull << lane) - 1);
                    if (below) link = p0 + (63 - __clzll((long long)below)) + 1; // nearest lower lane with the same hash
                    last = (grp >> lane) == 1ull;
                }
                clash &= ~grp;
            }
            if (live && last) vhead[h] = (uint16_t)(p + 1);
            link_stage[o] = (uint16_t)(live ? link : kNoLink);
        }
        __syncthreads();
    }
    if (nsuper) { // links of the last superblock
        const uint32_t p0 = (nsuper - 1) * kSuper + lane * 16;
        for (uint32_t k = 0; k < 16; k++) if (p0 + k < n) lk[p0 + k] = link_stage[lane * 16 + k];
    }
}

// ------------------------------------------------------------------------------------------------- K2
// The ring keeps the links of the most recent kRing positions: slot(q) = q mod kRing (q < 65536 < 2*kRing).
__device__ inline uint32_t ring_slot(uint32_t q) { return q >= kRing ? q - kRing : q; }

// Lane states of the walk.  A wave alternates between three homogeneous phases so that its lanes execute the same
// short code most of the time: WALK (one chain candidate per iteration: 1-byte quick reject + link hop, both LDS
// reads issued together), COMPARE (4 bytes of a full comparison per iteration, entered when enough lanes wait for
// one) and REFILL (store finished records, hand new positions to idle lanes, in groups of >= 16).
// LDS access rule (measured, scripts/micro/lds_cost.hip): naturally aligned ds_read_u8/u16/b32 cost ~4.3 clk per
// wave-instruction per CU with random addresses, any unaligned b32/b64 ~36 clk -- so only aligned reads are used
// (the dword pairs below are volatile so that the compiler cannot fuse them into one unaligned ds_read_b64).
enum : uint32_t { kIdle = 0, kWalk = 1, kCmp = 2, kDone = 3 };

struct WalkLane {
    uint32_t mode, p, q, best, bestq, steps, snap_best, snap_q, l, cap, nice, flags, scan2; // scan2: bytes p+best-1, p+best
    int limit_q; // chain continues while (int)q > limit_q; "none" decodes to q = -1 and always stops
};

// byte offset of an LDS object inside the workgroup's LDS allocation (what the ds_* instructions address)
__device__ inline uint32_t lds_offset(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// 4 bytes at any byte offset: one ds_read2_b32 of the two aligned dwords around it + v_alignbyte (the compiler would
// fuse two plain dword loads into a single unaligned ds_read_b64, which is ~8x slower)
__device__ inline uint32_t lds_load32u(uint32_t lds_byte_addr)
{
    uint64_t v;
    const uint32_t a = lds_byte_addr & ~3u;
    asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return __builtin_amdgcn_alignbyte((uint32_t)(v >> 32), (uint32_t)v, lds_byte_addr & 3);
}

// the three loads of one walk step, issued back to back with a single wait: the two quick-reject bytes of the
// candidate and its chain link
__device__ inline void lds_walk_loads(uint32_t addr_b0, uint32_t addr_ring, uint32_t &b0, uint32_t &b1, uint32_t &link)
{
    asm volatile("ds_read_u8 %0, %3\n\tds_read_u8 %1, %3 offset:1\n\tds_read_u16 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(b0), "=&v"(b1), "=&v"(link) : "v"(addr_b0), "v"(addr_ring) : "memory");
}
// the same for the two positions a lane works on at once: six loads in flight behind one wait
__device__ inline void lds_walk_loads2(uint32_t a0, uint32_t r0, uint32_t a1, uint32_t r1, uint32_t &x0, uint32_t &y0, uint32_t &l0,
                                       uint32_t &x1, uint32_t &y1, uint32_t &l1)
{
    asm volatile("ds_read_u8 %0, %6\n\tds_read_u8 %1, %6 offset:1\n\tds_read_u16 %2, %7\n\t"
                 "ds_read_u8 %3, %8\n\tds_read_u8 %4, %8 offset:1\n\tds_read_u16 %5, %9\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x0), "=&v"(y0), "=&v"(l0), "=&v"(x1), "=&v"(y1), "=&v"(l1) : "v"(a0), "v"(r0), "v"(a1), "v"(r1) : "memory");
}

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
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q0 + 8 <= n) v = *reinterpret_cast<const uint4 *>(lk + q0);
            else if (q0 < n) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; q0 + k < n; k++) w[k >> 1] |= (uint32_t)lk[q0 + k] << (16 * (k & 1));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4 *>(ring + ring_slot(q0)) = v;
            if (tid == 0) *tile_next = tile_lo;
        }
        __syncthreads();

        // A lane can work on kSlots positions at once ("slot" = one of the 64*kSlots walks of a wave).  Measured on MI355X:
        // kSlots = 2 (six LDS loads in flight per lane) is 25 % slower than kSlots = 1, the walk is issue-bound, not
        // latency-bound, so 1 it is.
        WalkLane s[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; j++) s[j] = WalkLane{};
        uint32_t sup_next = 0, sup_end = 0; // this wave's private supply of positions (wave-uniform)
        bool tile_dry = false;
        const uint32_t dbase = lds_offset(d32), rbase = lds_offset(ring);

        for (;;) {
            uint32_t nw = 0, nc = 0;
#pragma unroll
            for (int j = 0; j < kSlots; j++) { nw += (uint32_t)__popcll(__ballot(s[j].mode == kWalk)); nc += (uint32_t)__popcll(__ballot(s[j].mode == kCmp)); }
            const uint32_t nfree = 64 * kSlots - nw - nc;

            // ---- REFILL: store finished records, hand out new positions ----
            if (nfree >= ZGPU_REFILL_MIN * kSlots || nw + nc == 0) {
#pragma unroll
                for (int j = 0; j < kSlots; j++) {
                    WalkLane &z = s[j];
                    if (z.mode == kDone) {
                        uint32_t full = z.best | ((z.p - z.bestq) << 9), snap = z.snap_best | ((z.p - z.snap_q) << 9);
                        if (z.steps < chainQ) snap = full;
                        if (z.best < kMinMatch) full = 0;
                        if ((snap & 511) < kMinMatch) snap = 0;
                        rec[z.p] = make_uint2(full | ((uint32_t)d8[z.p] << 24), snap | (z.flags << 24));
                        z.mode = kIdle;
                    }
                }
                for (int round = 0; round < 4 * kSlots; round++) {
                    const int j = round % kSlots;
                    WalkLane &z = s[j];
                    const unsigned long long idle = __ballot(z.mode == kIdle);
                    const uint32_t nidle = (uint32_t)__popcll(idle);
                    if (nidle == 0) continue;
                    if (sup_next == sup_end && !tile_dry) { // fetch another batch of positions for this wave
                        uint32_t got = 0;
                        if (lane == 0) got = atomicAdd(tile_next, 256u * kSlots);
                        got = __shfl(got, 0);
                        if (got >= tile_hi) tile_dry = true;
                        else { sup_next = got; sup_end = got + 256 * kSlots < tile_hi ? got + 256 * kSlots : tile_hi; }
                    }
                    if (sup_next == sup_end) break;
                    if (z.mode == kIdle) {
                        const uint32_t np = sup_next + (uint32_t)__popcll(idle & ((1ull << lane) - 1));
                        if (np < sup_end) {
                            z.p = np;
                            const uint32_t look = n - np;
                            z.cap = look < kMaxMatch ? look : kMaxMatch;
                            z.nice = cfg.nice < look ? cfg.nice : look;
                            const int w = (int)(np + base), limit = w > (int)kMaxDist ? w - (int)kMaxDist : 0;
                            const int lq = limit - (int)base; // q + base > limit  <=>  q > lq
                            z.limit_q = lq < -1 ? -1 : lq;
                            z.q = (uint32_t)ring[ring_slot(np)] - 1;
                            const bool valid = look >= kMinMatch && z.q != 0xFFFFFFFFu && (int)(z.q + base) > 0 && (uint32_t)(w - (int)(z.q + base)) <= kMaxDist;
                            z.flags = (valid && z.q + base == kWSize) ? 1u : 0u;
                            z.best = kMinMatch - 1; z.bestq = np; z.steps = 0; z.snap_best = 0; z.snap_q = np;
                            z.scan2 = (uint32_t)d8[np + 1] | ((uint32_t)d8[np + 2] << 8);
                            if (valid) z.mode = kWalk;
                            else rec[np] = make_uint2((uint32_t)d8[np] << 24, 0); // no candidate at all
                        }
                    }
                    const uint32_t take = nidle < sup_end - sup_next ? nidle : sup_end - sup_next;
                    sup_next += take;
                }
                unsigned long long busy = 0;
#pragma unroll
                for (int j = 0; j < kSlots; j++) busy |= __ballot(s[j].mode == kWalk || s[j].mode == kCmp);
                if (busy == 0) {
                    if (tile_dry && sup_next == sup_end) break; // nothing left in this tile for this wave
                    continue;                                   // only candidate-less positions so far: refill again
                }
            }

            // ---- COMPARE: when many slots wait for a full comparison, or few can walk ----
            if (nc >= ZGPU_CMP_MIN * kSlots || (nc > 0 && nw < 16 * kSlots)) {
#pragma unroll
                for (int j = 0; j < kSlots; j++) {
                    WalkLane &z = s[j];
                    while (__ballot(z.mode == kCmp)) {
                        if (z.mode == kCmp) {
                            const uint32_t x = lds_load32u(dbase + z.q + z.l) ^ lds_load32u(dbase + z.p + z.l);
                            if (x == 0 && z.l + 4 < z.cap) z.l += 4;
                            else {
                                uint32_t len = x ? z.l + ((uint32_t)__builtin_ctz(x) >> 3) : z.l + 4;
                                len = len < z.cap ? len : z.cap;
                                if (len > z.best) { z.best = len; z.bestq = z.q; z.scan2 = (uint32_t)d8[z.p + len - 1] | ((uint32_t)d8[z.p + len] << 8); }
                                // candidate dealt with: count it and hop (deflate.c:1152-1164)
                                z.steps++;
                                if (z.steps == chainQ) { z.snap_best = z.best; z.snap_q = z.bestq; }
                                const uint32_t nq = (uint32_t)ring[ring_slot(z.q)] - 1;
                                const bool stop = z.best >= z.nice || z.steps == chainF || (int)nq <= z.limit_q;
                                z.q = nq;
                                z.mode = stop ? kDone : kWalk;
                            }
                        }
                    }
                }
            }

            // ---- WALK: a few candidates per slot; the quick-reject bytes and the next link are fetched together ----
#pragma unroll
            for (int k = 0; k < ZGPU_WALK_UNROLL; k++) {
                bool w[kSlots], any = false;
#pragma unroll
                for (int j = 0; j < kSlots; j++) { w[j] = s[j].mode == kWalk; any = any || w[j]; }
                if (any) {
                    uint32_t x[2], y[2], e[2];
                    if constexpr (kSlots == 1) {
                        lds_walk_loads(dbase + s[0].q + s[0].best - 1, rbase + 2 * ring_slot(s[0].q), x[0], y[0], e[0]);
                    } else {
                        // a slot that is not walking still issues its loads (one asm block for both): point them at a valid address
                        const uint32_t q0 = w[0] ? s[0].q : 0u, q1 = w[kSlots - 1] ? s[kSlots - 1].q : 0u;
                        lds_walk_loads2(dbase + q0 + s[0].best - 1, rbase + 2 * ring_slot(q0), dbase + q1 + s[kSlots - 1].best - 1,
                                        rbase + 2 * ring_slot(q1), x[0], y[0], e[0], x[1], y[1], e[1]);
                    }
#pragma unroll
                    for (int j = 0; j < kSlots; j++) {
                        WalkLane &z = s[j];
                        if (w[j]) {
                            if ((x[j] | (y[j] << 8)) == z.scan2) { z.mode = kCmp; z.l = 0; } // may be longer than best: compare in full
                            else {
                                const uint32_t nq = e[j] - 1;
                                z.steps++;
                                if (z.steps == chainQ) { z.snap_best = z.best; z.snap_q = z.bestq; }
                                const bool stop = z.steps == chainF || (int)nq <= z.limit_q; // best < nice while walking
                                z.q = nq;
                                z.mode = stop ? kDone : kWalk;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- K3
// One lane per chunk.  Measured alternatives (MI355X, 1 GiB = 16384 chunks): a wave-per-chunk scalar automaton over a
// register window of records is bound by the single scalar unit per CU (59 ms); lane-per-chunk with per-lane LDS
// windows refilled by divergent 16-byte loads is bound by those refills (187 ms); this plain form costs 85 ms at
// 1 GiB (one wave per CU, latency-bound) and scales with the number of resident waves at larger inputs.
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
            const uint32_t pick = (prev_len >= cfg.good && cfg.strategy != kRle) ? r.y : r.x;
            uint32_t len = pick & 511, dist = (pick >> 9) & 32767;
            if (off != 0 && ((r.y >> 24) & 1)) len = 0; // first candidate became NIL in the slide
            if (len > prev_len) {
                match_len = len; cur_dist = dist;
                if (match_len <= 5 && (cfg.strategy == kFiltered || (match_len == kMinMatch && cur_dist > kTooFar))) match_len = kMinMatch - 1;
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
    if (pending) { tok[ntok++] = tok_lit(prev_byte); if (ntok - blk_tok0 == kBlockTokens) nostore |= kFullFinalBlock; } // (deflate.c:1660-1665: no cut behind this literal)
    if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk;
    meta[c].ntok = ntok; meta[c].nostore = nostore; meta[c].in_bytes = n;
}

void launch_parse2(const ChunkGeom &g, LevelCfg cfg, const uint2 *recs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st); // zgpu_lz_parse.hip

void launch_parse(const ChunkGeom &g, LevelCfg cfg, const uint2 *recs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st)
{
    static int serial = -1; // ZGPU_PARSE=1: the lane-per-chunk restatement of the reference loop (kept as the cross-check)
    if (serial < 0) { const char *e = getenv("ZGPU_PARSE"); serial = e && atoi(e) == 1 ? 1 : 0; }
    if (serial) hipLaunchKernelGGL(parse_kernel, dim3((g.nchunks + 63) / 64), dim3(64), 0, st, g, cfg, recs, tokens, meta);
    else launch_parse2(g, cfg, recs, tokens, meta, st);
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
