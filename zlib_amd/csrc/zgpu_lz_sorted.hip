// zgpu_lz_sorted.hip -- LZ77 stage for levels 4-9, second parallel form: hash buckets as sorted arrays.
//
// Same mathematics as zgpu_lz_parallel.hip (static chains, all-position search with two budgets, record-driven lazy
// parse -- SURVEY.md 8a A4/A5), different data structure.  Instead of walking `link(p)` pointers through an LDS ring, the
// positions of a chunk are counting-sorted by their 3-byte hash:
//
//     S[start(h) .. start(h)+count(h))  = the positions with hash h, ascending
//     idx(p)   = index of p in S,   rank(p) = number of earlier positions with the same hash
//
// so the hash chain of p, nearest candidate first, is simply S[idx-1], S[idx-2], ... S[idx-rank]: no dependent pointer
// hop per candidate (the next candidates are known in advance and are fetched four at a time with one 8-byte load), no
// link ring in LDS (the match kernel needs only the 64 KiB chunk there, so two 1024-lane workgroups fit a CU instead
// of one), no tiles and no per-tile barriers (any position can be searched at any time).
//
//   K1' sort_kernel    one wave per chunk: rank(p) by a sequential pass over count[h] in LDS (64 positions per step,
//                      duplicates inside a step resolved with ballots), exclusive scan of the counts, scatter to S.
//   K2' match2_kernel  one 1024-lane workgroup per chunk, lanes = positions, the state machine of zgpu_lz_parallel.hip
//                      with rounds of four candidates.
//   K3                 parse_kernel of zgpu_lz_parallel.hip, unchanged (same record format).
#include "zgpu_common.h"
#include <cstdlib>
#include "../../include/zamd_gpu.h"

namespace zgpu {

void prof_span_begin(void *eng, hipStream_t st, hipEvent_t *a);
void prof_span_end(void *eng, hipStream_t st, int stage, hipEvent_t a);
void launch_parse(const ChunkGeom &g, LevelCfg cfg, const uint2 *recs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st);

constexpr uint32_t kSuperS = 1024;   // positions per superblock of the sort passes
constexpr uint32_t kSPad = 8;        // u16 entries in front of every chunk's S (group loads may reach below index 0)
constexpr uint32_t kSStride = kChunkMax + kSPad;

// per-chunk workspace layout (bytes): S | rank | idx | records
size_t lz_sorted_workspace_bytes(uint32_t batch) { return (size_t)batch * (kSStride * 2 + kChunkMax * 2 + kChunkMax * 2 + kChunkMax * 8) + 1024; }

// ------------------------------------------------------------------------------------------------- K1'
// One 256-lane workgroup (4 waves) per chunk.  Pass A is sequential in position order only among positions that share a
// hash, so the hash space is split four ways: every wave sweeps all positions but counts only the hashes with
// (h & 3) == its wave number -- four sequential sweeps run side by side on the four SIMDs with a quarter of the
// duplicate-resolution work each.  Passes B and C are plainly parallel over the 256 lanes.
#ifndef ZGPU_SORT_WAVES
#define ZGPU_SORT_WAVES 8
#endif
constexpr uint32_t kSortWaves = ZGPU_SORT_WAVES, kSortThreads = 64 * kSortWaves; // waves per chunk: 4 or 8 (or 16)

__global__ void __launch_bounds__(kSortThreads) sort_kernel(ChunkGeom g, uint16_t *__restrict__ S_all, uint16_t *__restrict__ rank_all, uint16_t *__restrict__ idx_all)
{
    __shared__ uint16_t cnt[kHashSize];                                         // counts, then bucket starts
    __shared__ __attribute__((aligned(16))) uint32_t in_stage[kSuperS / 4 + 4]; // 1 KiB of input + 8 bytes of the next
    __shared__ __attribute__((aligned(16))) uint16_t out_stage[kSuperS];
    __shared__ uint8_t tag[4096];
    __shared__ uint32_t wave_tot[kSortWaves];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    uint16_t *S = S_all + (size_t)c * kSStride + kSPad, *rk = rank_all + (size_t)c * kChunkMax, *ix = idx_all + (size_t)c * kChunkMax;
    for (uint32_t i = tid; i < kHashSize / 2; i += kSortThreads) reinterpret_cast<uint32_t *>(cnt)[i] = 0;
    const uint32_t npos = n >= 3 ? n - 2 : 0;
    volatile uint16_t *vcnt = cnt;
    volatile uint8_t *vtag = tag;
    const unsigned long long lt_mask = (1ull << lane) - 1;
    const bool aligned = (reinterpret_cast<uintptr_t>(src) & 3) == 0;
    auto fetch = [&](uint32_t sb) -> uint32_t { // dword `tid` of superblock sb (threads 0..255), zero padded past n
        if (tid >= kSuperS / 4) return 0;
        const uint32_t a = sb * kSuperS + tid * 4;
        if (a + 4 <= n && aligned) return *reinterpret_cast<const uint32_t *>(src + a);
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4; k++) if (a + k < n) v |= (uint32_t)src[a + k] << (8 * k);
        return v;
    };
    auto flush = [&](uint16_t *dst_base, uint32_t sb) { // out_stage -> global, 8 bytes per lane (threads 0..255)
        if (tid >= kSuperS / 4) return;
        const uint32_t p0 = sb * kSuperS + tid * 4;
        if (p0 + 4 <= n) *reinterpret_cast<uint2 *>(dst_base + p0) = reinterpret_cast<const uint2 *>(out_stage)[tid];
        else for (uint32_t k = 0; k < 4; k++) if (p0 + k < n) dst_base[p0 + k] = out_stage[tid * 4 + k];
    };
    const uint32_t nsuper = (n + kSuperS - 1) / kSuperS;
    const uint8_t *s8 = reinterpret_cast<const uint8_t *>(in_stage);

    // ---- pass A: rank(p) = number of earlier positions with the same hash ----
    uint32_t cur = fetch(0), nxt = fetch(1);
    __syncthreads();
    for (uint32_t sb = 0; sb < nsuper; sb++) {
        if (tid < kSuperS / 4) in_stage[tid] = cur;
        if (tid < 2) in_stage[kSuperS / 4 + tid] = __shfl(nxt, tid);
        if (sb > 0) flush(rk, sb - 1);
        cur = nxt; nxt = fetch(sb + 2);
        __syncthreads();
        const uint32_t base_p = sb * kSuperS;
#pragma unroll 1
        for (uint32_t st = 0; st < kSuperS / 64; st++) {
            const uint32_t o = st * 64 + lane, p = base_p + o;
            const uint32_t h = hash3(s8[o], s8[o + 1], s8[o + 2]);
            const bool live = p < npos && (h & (kSortWaves - 1)) == wave; // this wave owns its share of the hash space
            uint32_t old = 0;
            if (live) old = vcnt[h];
            // lanes of this step that share a hash: detected through a small tag table (a false alarm is harmless; the
            // index keeps the two ownership bits, so waves never touch each other's tags)
            if (live) vtag[h & 4095] = (uint8_t)lane;
            const uint32_t seen = live ? (uint32_t)vtag[h & 4095] : lane;
            unsigned long long clash = __ballot(live && seen != lane);
            uint32_t rank = old, group = 1;
            bool last = live;
            while (clash) {
                const int f = __ffsll((long long)clash) - 1;
                const uint32_t h0 = __builtin_amdgcn_readlane(h, f);
                const unsigned long long grp = __ballot(live && h == h0);
                if (live && h == h0) {
                    rank = old + (uint32_t)__popcll(grp & lt_mask);
                    group = (uint32_t)__popcll(grp);
                    last = (grp >> lane) == 1ull;
                }
                clash &= ~grp;
            }
            if (live && last) vcnt[h] = (uint16_t)(old + group);
            if (live) out_stage[o] = (uint16_t)rank;
            else if (p >= npos && wave == 0) out_stage[o] = 0;
        }
        __syncthreads();
    }
    if (nsuper && tid < kSuperS / 4) { const uint32_t p0 = (nsuper - 1) * kSuperS + tid * 4; for (uint32_t k = 0; k < 4; k++) if (p0 + k < n) rk[p0 + k] = out_stage[tid * 4 + k]; }
    __syncthreads();

    // ---- pass B: exclusive scan of the 32768 counts -> bucket starts (in place), 128 consecutive counts per lane ----
    {
        const uint32_t per = kHashSize / kSortThreads;
        uint32_t sum = 0;
        for (uint32_t i = 0; i < per; i++) sum += cnt[tid * per + i];
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        if (lane == 63) wave_tot[wave] = x;
        __syncthreads();
        uint32_t basev = x - sum;
        for (uint32_t w = 0; w < wave; w++) basev += wave_tot[w];
        for (uint32_t i = 0; i < per; i++) { const uint32_t v = cnt[tid * per + i]; cnt[tid * per + i] = (uint16_t)basev; basev += v; }
    }
    __syncthreads();

    // ---- pass C: idx(p) = start(hash) + rank(p);  S[idx] = p ----
    cur = fetch(0); nxt = fetch(1);
    for (uint32_t sb = 0; sb < nsuper; sb++) {
        if (tid < kSuperS / 4) in_stage[tid] = cur;
        if (tid < 2) in_stage[kSuperS / 4 + tid] = __shfl(nxt, tid);
        cur = nxt; nxt = fetch(sb + 2);
        __syncthreads();
        const uint32_t base_p = sb * kSuperS;
#pragma unroll
        for (uint32_t st = 0; st < kSuperS / kSortThreads; st++) {
            const uint32_t o = st * kSortThreads + tid, p = base_p + o;
            if (p < npos) {
                const uint32_t h = hash3(s8[o], s8[o + 1], s8[o + 2]), r = rk[p], id = (uint32_t)cnt[h] + r;
                S[id] = (uint16_t)p;
                ix[id] = (uint16_t)r; // rank in S order: the match kernel walks S, not the positions
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------- K2'
__device__ inline uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }
__device__ inline uint32_t lds_ld32u(uint32_t a) // 4 bytes at any LDS byte offset: aligned ds_read2_b32 + v_alignbyte
{
    uint64_t v; const uint32_t al = a & ~3u;
    asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(al) : "memory");
    return __builtin_amdgcn_alignbyte((uint32_t)(v >> 32), (uint32_t)v, a & 3);
}
__device__ inline void lds_ld2bytes(uint32_t a, uint32_t &b0, uint32_t &b1)
{
    asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %2 offset:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(a) : "memory");
}
__device__ inline uint64_t gload64u(const uint16_t *p) // 8 bytes at 2-byte alignment (gfx950 serves it, scripts/micro/global_unaligned.hip)
{
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct __attribute__((packed, aligned(2))) U64u { uint64_t v; }; // 8 bytes at 2-byte alignment: one global_load_dwordx2 on gfx950
__device__ inline void lds_ld8bytes(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t (&b)[8])
{
    asm volatile("ds_read_u8 %0, %8\n\tds_read_u8 %1, %8 offset:1\n\tds_read_u8 %2, %9\n\tds_read_u8 %3, %9 offset:1\n\t"
                 "ds_read_u8 %4, %10\n\tds_read_u8 %5, %10 offset:1\n\tds_read_u8 %6, %11\n\tds_read_u8 %7, %11 offset:1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}

enum : uint32_t { sIdle = 0, sWalk = 1, sCmp = 2, sDone = 3 };
constexpr uint32_t kM2Threads = 1024;
#ifndef ZGPU_M2_REFILL
#define ZGPU_M2_REFILL 16
#endif
#ifndef ZGPU_M2_CMP
#define ZGPU_M2_CMP 16
#endif
#ifndef ZGPU_M2_ROUNDS
#define ZGPU_M2_ROUNDS 1
#endif

// chunk bytes -> LDS (zero padded to kChunkMax + 64), by all kM2Threads lanes of the workgroup
__device__ inline void stage_chunk(const uint8_t *src, uint32_t n, uint32_t *d32, uint32_t tid)
{
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *s128 = reinterpret_cast<const uint4 *>(src);
        uint4 *d128 = reinterpret_cast<uint4 *>(d32);
        const uint32_t nv = n >> 4;
        for (uint32_t i = tid; i < (kChunkMax + 64) / 16; i += kM2Threads) {
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
        for (uint32_t i = tid; i < (kChunkMax + 64) / 4; i += kM2Threads) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < 4; k++) { uint32_t a = (i << 2) + k; if (a < n) v |= (uint32_t)src[a] << (8 * (k & 3)); }
            d32[i] = v;
        }
    }
}

__global__ void __launch_bounds__(kM2Threads, 8) match2_kernel(ChunkGeom g, LevelCfg cfg, const uint16_t *__restrict__ S_all, const uint16_t *__restrict__ rank_all,
                                                               const uint16_t *__restrict__ idx_all, uint2 *__restrict__ recs)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *d32 = lds;                                  // 65536 + 64 bytes of chunk data
    uint32_t *work_next = lds + (kChunkMax + 64) / 4;     // next unassigned position of the chunk
    const uint8_t *d8 = reinterpret_cast<const uint8_t *>(d32);
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    const uint16_t *S = S_all + (size_t)c * kSStride + kSPad, *rkS = idx_all + (size_t)c * kChunkMax; // rank of S[i], in S order
    (void)rank_all;
    uint2 *rec = recs + (size_t)c * kChunkMax;
    const uint32_t npos = n >= 3 ? n - 2 : 0;
    const uint32_t base = chunk_base(g, c);

    stage_chunk(src, n, d32, tid);
    if (tid == 0) *work_next = 0;
    for (uint32_t q2 = npos + tid; q2 < n; q2 += kM2Threads) rec[q2] = make_uint2((uint32_t)src[q2] << 24, 0); // the last two positions carry no hash
    __syncthreads();

    const uint32_t chainF = cfg.chain, chainQ = cfg.chain >> 2, dbase = lds_off(d32);
    // per-lane walk state
    uint32_t mode = sIdle, p = 0, k = 0, avail = 0, best = 0, bestq = 0, snap_best = 0, snap_q = 0, l = 0, cap = 0, nice = 0, flags = 0, scan2 = 0, q = 0;
    int thr = 0;          // a candidate q is usable iff (int)q >= thr; thr rises by one after the first candidate (deflate.c:1588-1589 vs 1163)
    int thr_next = 0;
    uint64_t cq = 0;      // up to four candidates, next one in the top 16 bits
    const uint16_t *sp = S; // &S[idx]: candidate k is sp[-1-k]
    uint32_t sup_next = 0, sup_end = 0;
    bool dry = false;

    for (;;) {
        const uint32_t nw = (uint32_t)__popcll(__ballot(mode == sWalk)), nc = (uint32_t)__popcll(__ballot(mode == sCmp)), nfree = 64 - nw - nc;

        // ---- REFILL ----
        if (nfree >= ZGPU_M2_REFILL || nw + nc == 0) {
            if (mode == sDone) {
                uint32_t full = best | ((p - bestq) << 9), snap = snap_best | ((p - snap_q) << 9);
                if (best < kMinMatch) full = 0;
                // the one position whose first candidate can sit at window index 32768 (NIL after the slide, deflate.c:1309-1312)
                if (p + base == kWSize + kMaxDist && avail != 0 && (uint32_t)sp[-1] + base == kWSize) flags = 1;
                if ((snap & 511) < kMinMatch) snap = 0;
                rec[p] = make_uint2(full | ((uint32_t)d8[p] << 24), snap | (flags << 24));
                mode = sIdle;
            }
            for (int round = 0; round < 4; round++) {
                const unsigned long long idle = __ballot(mode == sIdle);
                const uint32_t nidle = (uint32_t)__popcll(idle);
                if (nidle == 0) break;
                if (sup_next == sup_end && !dry) {
                    uint32_t got = 0;
                    if (lane == 0) got = atomicAdd(work_next, 256u);
                    got = __builtin_amdgcn_readfirstlane(got);
                    if (got >= npos) dry = true;
                    else { sup_next = got; sup_end = got + 256 < npos ? got + 256 : npos; }
                }
                if (sup_next == sup_end) break;
                if (mode == sIdle) {
                    // work items are indices into S: neighbouring lanes get neighbouring entries of one hash bucket, i.e. chains
                    // of almost equal length (lane balance) that overlap in memory (the 8-byte candidate loads hit in cache)
                    const uint32_t wi = sup_next + (uint32_t)__popcll(idle & ((1ull << lane) - 1));
                    if (wi < sup_end) {
                        const uint32_t np = S[wi];
                        p = np;
                        const uint32_t look = n - np;
                        cap = look < kMaxMatch ? look : kMaxMatch;
                        nice = cfg.nice < look ? cfg.nice : look;
                        const int w = (int)(np + base);
                        const int t_first = (w - (int)kMaxDist > 1 ? w - (int)kMaxDist : 1) - (int)base;  // first candidate: dist <= MAX_DIST, not NIL
                        const int t_next = (w - (int)kMaxDist + 1 > 1 ? w - (int)kMaxDist + 1 : 1) - (int)base; // later ones: strictly inside
                        thr = t_first; thr_next = t_next;
                        const uint32_t rank = rkS[wi];
                        avail = rank < chainF ? rank : chainF;
                        k = 0; best = kMinMatch - 1; bestq = np; snap_best = 0; snap_q = np; flags = 0;
                        scan2 = (uint32_t)d8[np + 1] | ((uint32_t)d8[np + 2] << 8);
                        if (avail) { sp = S + wi; mode = sWalk; }
                        else rec[np] = make_uint2((uint32_t)d8[np] << 24, 0); // no candidate at all
                    }
                }
                const uint32_t take = nidle < sup_end - sup_next ? nidle : sup_end - sup_next;
                sup_next += take;
            }
            if (__ballot(mode == sWalk || mode == sCmp) == 0) {
                if (dry && sup_next == sup_end) break;
                continue;
            }
        }

        // ---- COMPARE ----
        if (nc >= ZGPU_M2_CMP || (nc > 0 && nw < 16)) {
            while (__ballot(mode == sCmp)) {
                if (mode == sCmp) {
                    const uint32_t x = lds_ld32u(dbase + q + l) ^ lds_ld32u(dbase + p + l);
                    if (x == 0 && l + 4 < cap) l += 4;
                    else {
                        uint32_t len = x ? l + ((uint32_t)__builtin_ctz(x) >> 3) : l + 4;
                        len = len < cap ? len : cap;
                        if (len > best) {
                            best = len; bestq = q; scan2 = (uint32_t)d8[p + len - 1] | ((uint32_t)d8[p + len] << 8);
                            if (k < chainQ) { snap_best = len; snap_q = q; } // candidate number k+1 is within the quarter budget
                        }
                        k++;
                        mode = (best >= nice || k >= avail) ? sDone : sWalk;
                    }
                }
            }
        }

        // ---- WALK: rounds of up to four candidates (one 8-byte load of S per group of four) ----
#pragma unroll 1
        for (int rnd = 0; rnd < ZGPU_M2_ROUNDS; rnd++) {
        if (mode == sWalk && (k & 3) == 0) cq = gload64u(sp - 4 - k);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (mode == sWalk) {
                q = (uint32_t)(cq >> 48);
                if ((int)q < thr) mode = sDone; // beyond MAX_DIST (or the NIL position): the chain ends here (deflate.c:1163)
                else {
                    thr = thr_next;
                    uint32_t b0, b1;
                    lds_ld2bytes(dbase + q + best - 1, b0, b1);
                    cq <<= 16;
                    if ((b0 | (b1 << 8)) == scan2) { mode = sCmp; l = 0; } // candidate may be longer than best: compare in full
                    else {
                        k++;
                        if (k >= avail) mode = sDone;
                        else if ((k & 3) == 0) break; // the next group of candidates is fetched at the top of the next round
                    }
                }
            }
        }
        }
    }
}

// ------------------------------------------------------------------------------------------------- K2''
// Lockstep form of the same search.  A wave takes 64 CONSECUTIVE entries of S (one work item = one block of 64 S
// indices); lane L searches position S[wi], wi = 64*blk + L, and all lanes examine their k-th candidate S[wi-1-k] in the
// same step.  Neighbouring entries of S belong to the same hash bucket, so their chains have (almost) the same length and
// the candidate loads of a step are one contiguous, coalesced run of S.
//
// The full string comparison is taken out of the walk altogether.  longest_match's result has an order-free statement:
// with len(k) = common prefix of the scan string and candidate k (capped by the lookahead), the walk ends at the FIRST k
// with len(k) >= nice_match (deflate.c:1224) and returns that candidate; otherwise it returns the largest len(k), the
// earliest k among equals (deflate.c:1198, strict >).  Both are the maximum of one integer key per candidate:
//       len >= nice :  1<<31 | (4095-k)<<16 | len            len < nice :  len<<16 | (4095-k)
// So a candidate that passes the two-byte quick check (deflate.c:1187-1190) is merely appended to a per-wave ring in LDS,
// and whenever 64 of them have gathered, the 64 lanes compute 64 prefix lengths at full occupancy -- any lane serves any
// owner -- and fold the keys into the owner's slot with an LDS atomic max.  The owners then read their slot back and walk
// on with the new best length.  Walking with a stale (shorter) best length is exact: the quick check against it rejects
// only candidates that match at most that many bytes, which can be neither an improvement nor a nice_match stop; and
// whatever was appended after the stopping candidate loses against its key.  The quarter-budget result (deflate.c:1146)
// is the slot as it stands once every candidate k < chain/4 has been folded.
#ifndef ZGPU_M3_PERIOD
#define ZGPU_M3_PERIOD 32 // steps between forced folds (keeps the walkers' best length fresh); a power of two <= 256
#endif
#ifdef ZGPU_M3_STATS // debug build only (scripts/m3_stats.py): 0 wave-steps, 1 active lane-steps, 2 ring entries, 3 folds, 4 fold iterations, 5 positions
__device__ unsigned long long m3_stats[8];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_m3_stats(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(m3_stats), sizeof z);
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(m3_stats), z, sizeof z);
}
#define M3_STAT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&m3_stats[i], v_); } while (0)
#else
#define M3_STAT(i, v) do { } while (0)
#endif
constexpr uint32_t kRing = 128;                                  // parked candidates per wave: q | owner<<16 | (k&1023)<<22
constexpr uint32_t kM3WaveLds = kRing * 4 + 64 * 4 + 64 * 2;     // ring + slots + owners' positions
constexpr uint32_t kM3DataLds = kChunkMax + 64 + 320;            // chunk bytes + zero pad + slack for reads past a garbage candidate
constexpr uint32_t kM3Lds = kM3DataLds + 16 + (kM2Threads / 64) * kM3WaveLds;

// wave-private LDS words by byte offset (plain C++ volatile accesses through a generic pointer compile to flat_* memory instructions)
__device__ inline void lds_st32(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline void lds_st16(uint32_t a, uint32_t v) { asm volatile("ds_write_b16 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline void lds_max32(uint32_t a, uint32_t v) { asm volatile("ds_max_u32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline uint32_t lds_ld32(uint32_t a) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ inline uint32_t lds_ld16(uint32_t a) { uint32_t v; asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v; }

__global__ void __launch_bounds__(kM2Threads, 8) match3_kernel(ChunkGeom g, LevelCfg cfg, const uint16_t *__restrict__ S_all, const uint16_t *__restrict__ idx_all,
                                                               uint2 *__restrict__ recs)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *d32 = lds;
    uint32_t *work_next = lds + kM3DataLds / 4; // next unassigned block of S
    const uint8_t *d8 = reinterpret_cast<const uint8_t *>(d32);
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ring = (uint32_t)__builtin_amdgcn_readfirstlane(lds_off(lds) + kM3DataLds + 16 + wave * kM3WaveLds), slot = ring + kRing * 4, pw = slot + 64 * 4; // LDS byte offsets
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    const uint16_t *S = S_all + (size_t)c * kSStride + kSPad, *rkS = idx_all + (size_t)c * kChunkMax; // rank of S[i], in S order
    uint2 *rec = recs + (size_t)c * kChunkMax;
    const uint32_t npos = n >= 3 ? n - 2 : 0;
    const uint32_t base = chunk_base(g, c);
    stage_chunk(src, n, d32, tid);
    if (tid == 0) *work_next = 0;
    for (uint32_t q2 = npos + tid; q2 < n; q2 += kM2Threads) rec[q2] = make_uint2((uint32_t)src[q2] << 24, 0); // the last two positions carry no hash
    __syncthreads();

    const uint32_t chainF = cfg.chain, chainQ = cfg.chain >> 2, dbase = lds_off(d32), nblk = (npos + 63) >> 6;
    for (;;) {
        uint32_t blk = 0;
        if (lane == 0) blk = atomicAdd(work_next, 1u);
        blk = __builtin_amdgcn_readfirstlane(blk);
        if (blk >= nblk) break;
        const uint32_t wi = (blk << 6) + lane;
        const bool valid = wi < npos;
        uint32_t p = 0, avail = 0;
        if (valid) { p = S[wi]; const uint32_t rank = rkS[wi]; avail = rank < chainF ? rank : chainF; }
        const int w = (int)(p + base);
        int thr = (w - (int)kMaxDist > 1 ? w - (int)kMaxDist : 1) - (int)base;                 // first candidate: dist <= MAX_DIST, not NIL
        const int thr_next = (w - (int)kMaxDist + 1 > 1 ? w - (int)kMaxDist + 1 : 1) - (int)base; // later ones: strictly inside
        uint32_t best = kMinMatch - 1, key_seen = 0, snapkey = 0;
        uint32_t scan2 = (uint32_t)d8[p + 1] | ((uint32_t)d8[p + 2] << 8);
        uint32_t boff = dbase + best - 1;
        bool active = avail != 0, snap_taken = false;
        const uint16_t *sp = S + wi; // candidate k is sp[-1-k]
        lds_st32(slot + lane * 4, 0);
        lds_st16(pw + lane * 2, p);
        uint32_t tail = 0; // entries on the ring (wave-uniform)
        const uint32_t lanebits = lane << 16;
        M3_STAT(5, __popcll(__builtin_amdgcn_ballot_w64(valid)));

        // fold up to 64 ring entries into their owners' slots, then let every owner pick up its new best length
        auto fold = [&](uint32_t kcur) {
            const uint32_t cnt = tail < 64 ? tail : 64;
            M3_STAT(3, 1);
            tail = (uint32_t)__builtin_amdgcn_readfirstlane(tail - cnt); // the ring is a stack: any order of folding gives the same maxima
            if (lane < cnt) {
                const uint32_t e = lds_ld32(ring + ((tail + lane) << 2));
                const uint32_t q = e & 0xffffu, o = (e >> 16) & 63u;
                const uint32_t k = kcur - ((kcur - (e >> 22)) & 1023u);
                const uint32_t po = lds_ld16(pw + o * 2), look = n - po, cap = look < kMaxMatch ? look : kMaxMatch, nice = cfg.nice < look ? cfg.nice : look;
                uint32_t l = 0, x;
                for (;;) {
                    x = lds_ld32u(dbase + q + l) ^ lds_ld32u(dbase + po + l);
                    M3_STAT(4, 1); // (counts only the iterations lane 0 takes part in)
                    if (x != 0 || l + 4 >= cap) break;
                    l += 4;
                }
                uint32_t len = x ? l + ((uint32_t)__builtin_ctz(x) >> 3) : l + 4;
                len = len < cap ? len : cap;
                if (len >= kMinMatch) {
                    const uint32_t key = len >= nice ? (0x80000000u | ((4095u - k) << 16) | len) : ((len << 16) | (4095u - k));
                    lds_max32(slot + o * 4, key);
                }
            }
            const uint32_t key = lds_ld32(slot + lane * 4);
            if (key != key_seen) {
                key_seen = key;
                if (key >> 31) { best = key & 0x1ffu; active = false; }
                else best = key >> 16;
                boff = dbase + best - 1;
                scan2 = (uint32_t)d8[p + best - 1] | ((uint32_t)d8[p + best] << 8);
            }
        };

        uint64_t cq = 0;
        if (active) cq = reinterpret_cast<const U64u *>(sp - 4)->v;
        uint32_t k = 0;
        for (;; k += 4) {
            if (__builtin_amdgcn_ballot_w64(active) == 0) break;
            if (k == chainQ) {
                while (tail) fold(k);
                snapkey = key_seen; snap_taken = true;
            } else if ((k & (ZGPU_M3_PERIOD - 1)) == 0) { while (tail) fold(k); }
            uint64_t cqn = 0;
            if (active && k + 4 < avail) cqn = reinterpret_cast<const U64u *>(sp - 8 - k)->v;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                // no divergent control flow in the step: idle lanes read a harmless in-range address and are masked out of the ballot
                const uint32_t q = (uint32_t)(cq >> (48 - 16 * j)) & 0xffffu;
                uint32_t b0, b1;
                lds_ld2bytes(boff + q, b0, b1);
                active = active && (int)q >= thr; // beyond MAX_DIST (or the NIL position): the chain ends here (deflate.c:1163)
                if (k == 0 && j == 0) thr = thr_next;
                const bool pass = active && (b0 | (b1 << 8)) == scan2;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                M3_STAT(0, 1); M3_STAT(1, __popcll(__builtin_amdgcn_ballot_w64(active))); M3_STAT(2, __popcll(m));
                if (m) {
                    const uint32_t rt = (uint32_t)__builtin_amdgcn_readfirstlane(ring + (tail << 2));
                    if (pass) lds_st32(rt + (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)) << 2), q | lanebits | ((k + j) << 22));
                    tail = (uint32_t)__builtin_amdgcn_readfirstlane(tail + (uint32_t)__popcll(m));
                    if (tail >= 64) fold(k + j);
                }
                active = active && k + j + 1 < avail;
            }
            cq = cqn;
        }
        while (tail) fold(k);
        if (!snap_taken) snapkey = key_seen;
        if (valid) {
            uint32_t full = 0, snap = 0, flags = 0;
            if (key_seen) {
                const uint32_t kb = 4095u - ((key_seen >> 31) ? (key_seen >> 16) & 0xfffu : key_seen & 0xfffu);
                full = best | ((p - (uint32_t)sp[-1 - (int)kb]) << 9);
            }
            if (snapkey) {
                const uint32_t ks = 4095u - ((snapkey >> 31) ? (snapkey >> 16) & 0xfffu : snapkey & 0xfffu);
                const uint32_t sl = (snapkey >> 31) ? snapkey & 0x1ffu : snapkey >> 16;
                snap = sl | ((p - (uint32_t)sp[-1 - (int)ks]) << 9);
            }
            // the one position whose first candidate can sit at window index 32768 (NIL after the slide, deflate.c:1309-1312)
            if (p + base == kWSize + kMaxDist && avail != 0 && (uint32_t)sp[-1] + base == kWSize) flags = 1;
            rec[p] = make_uint2(full | ((uint32_t)d8[p] << 24), snap | (flags << 24));
        }
    }
}

void launch_lz_sorted(const ChunkGeom &g, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, void *prof)
{
    uint8_t *w = static_cast<uint8_t *>(workspace);
    const size_t nch = g.nchunks;
    uint16_t *S = reinterpret_cast<uint16_t *>(w);
    uint16_t *rk = reinterpret_cast<uint16_t *>(w + ((nch * kSStride * 2 + 255) & ~(size_t)255));
    uint16_t *ix = rk + nch * kChunkMax;
    uint2 *recs = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(ix + nch * kChunkMax));
    hipEvent_t ev{};
    prof_span_begin(prof, st, &ev);
    hipLaunchKernelGGL(sort_kernel, dim3(g.nchunks), dim3(kSortThreads), 0, st, g, S, rk, ix);
    prof_span_end(prof, st, ZGPU_STAGE_CHAIN, ev);
    prof_span_begin(prof, st, &ev);
    const size_t lds_bytes = (kChunkMax + 64) + 64 + 320; // slack: speculative quick-reject reads may reach 258 bytes past a garbage candidate
    static bool opt_in = false, opt_in3 = false;
    if (!opt_in) { hipFuncSetAttribute(reinterpret_cast<const void *>(match2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); opt_in = true; }
    static int impl = -1;
    if (impl < 0) { const char *e = getenv("ZGPU_MATCH"); impl = e ? atoi(e) : 3; }
    if (!opt_in3) { hipFuncSetAttribute(reinterpret_cast<const void *>(match3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kM3Lds); opt_in3 = true; }
    if (impl == 2) hipLaunchKernelGGL(match2_kernel, dim3(g.nchunks), dim3(kM2Threads), lds_bytes, st, g, cfg, S, rk, ix, recs);
    else hipLaunchKernelGGL(match3_kernel, dim3(g.nchunks), dim3(kM2Threads), kM3Lds, st, g, cfg, S, ix, recs);
    prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
    prof_span_begin(prof, st, &ev);
    launch_parse(g, cfg, recs, tokens, meta, st);
    prof_span_end(prof, st, ZGPU_STAGE_PARSE, ev);
}

} // namespace zgpu
