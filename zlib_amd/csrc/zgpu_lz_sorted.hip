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
//   K1'' sort3_kernel   the sort: ranks by ordered LDS atomics under a wave token, scatter staged through LDS, self-check;
//                       output S (u16 positions), one bit per S index marking bucket heads, heads_below per 64 indices
//   K1'  sort_kernel    the same result with ballots only (6x slower): fallback when sort3's self-check fails, ZGPU_SORT=1
//   K2'' match3_kernel  one 1024-lane workgroup per chunk, a wave per 64 consecutive S entries, all lanes on their k-th
//                       candidate in the same step, full compares deferred to a per-wave stack and folded with atomic max
//   K3                  parse2_kernel (zgpu_lz_parse.hip), or parse_kernel of zgpu_lz_parallel.hip with ZGPU_PARSE=1
#include "zgpu_common.h"
#define ZGPU_PARSE_HEADER_ONLY
#include "zgpu_lz_parse.h"
#include <cstdlib>
#include "../../include/zamd_gpu.h"

#include <atomic>
static std::atomic<int> g_inject_sort_fault{0};
extern "C" __attribute__((visibility("default"))) void zgpu_debug_inject_sort_fault(void) { g_inject_sort_fault.store(1); }

namespace zgpu {

void prof_span_begin(void *eng, hipStream_t st, hipEvent_t *a);
void prof_span_end(void *eng, hipStream_t st, int stage, hipEvent_t a);
void launch_parse(const ChunkGeom &g, LevelCfg cfg, const uint2 *recs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st);
void launch_parse_lite(const ChunkGeom &g, LevelCfg cfg, const uint32_t *gm, const uint32_t *gs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st);
void launch_lz_fastwin(const ChunkGeom &g, LevelCfg cfg, const uint16_t *S, const uint32_t *ir, uint32_t *tokens, ChunkMeta *meta, hipStream_t st);

__device__ inline uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }
// wave-private LDS words by byte offset (plain C++ volatile accesses through a generic pointer compile to flat_* memory instructions)
__device__ inline void lds_st32(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline void lds_st16(uint32_t a, uint32_t v) { asm volatile("ds_write_b16 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline void lds_max32(uint32_t a, uint32_t v) { asm volatile("ds_max_u32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline uint32_t lds_ld32(uint32_t a) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ inline uint32_t lds_ld16(uint32_t a) { uint32_t v; asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v; }

__device__ inline void lds_st8(uint32_t a, uint32_t v) { asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline uint32_t lds_ld8(uint32_t a) { uint32_t v; asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v; }

constexpr uint32_t kSuperS = 1024;   // positions per superblock of the sort passes
constexpr uint32_t kSPad = 8;        // entries in front of every chunk's S (group loads may reach below index 0)
constexpr uint32_t kSStride = kChunkMax + kSPad;

// workspace layout: 256-byte header (fault word) | S (u16) | rank, then idx, by position (u16) | bucket-head bits | records |
// ir: idx | rank << 16 by position (u32; what walk_kernel looks a position up with)

constexpr uint32_t kHeadWords = kChunkMax / 32;                  // dwords of bucket-head bits per chunk ...
constexpr uint32_t kHeadStride = kHeadWords + kChunkMax / 64 / 2; // ... followed by one u16 per 64 S indices (dwords per chunk)

size_t lz_sorted_workspace_bytes(uint32_t batch) { return 256 + (size_t)batch * (kSStride * 2 + kChunkMax * 2 + kHeadStride * 4 + kChunkMax * 8 + kChunkMax * 4) + 1024; }

// For block w of 64 S indices (1024 lanes, lane w holds the 64 head bits of its block): the distance from index 64*w down
// to the last bucket head below the block, 65535 if there is none (w == 0) or it is farther.  The rank of an S entry in
// its bucket is its distance from the last head at or below it, and this is the part of it that lies outside its own block.
__device__ inline uint32_t heads_below(unsigned long long x, uint32_t tid, uint32_t *wave_scratch)
{
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t v = x ? 64u * tid + 64u - (uint32_t)__builtin_clzll(x) : 0; // 1 + index of the highest head of the block, 0: none
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if ((int)lane >= d && y > inc) inc = y; }
    if (lane == 63) wave_scratch[wave] = inc;
    __syncthreads();
    uint32_t ex = __shfl_up(inc, 1);
    if (lane == 0) ex = 0;
    for (uint32_t w = 0; w < wave; w++) { const uint32_t y = wave_scratch[w]; if (y > ex) ex = y; }
    const uint32_t dist = 64u * tid - (ex - 1);
    return ex && dist < 65535u ? dist : 65535u;
}

__global__ void __launch_bounds__(1024) heads_below_kernel(uint32_t *__restrict__ heads_all) // after sort_kernel (the fallback path)
{
    __shared__ uint32_t scratch[16];
    uint32_t *hd = heads_all + (size_t)blockIdx.x * kHeadStride;
    const uint32_t b = heads_below(reinterpret_cast<const unsigned long long *>(hd)[threadIdx.x], threadIdx.x, scratch);
    reinterpret_cast<uint16_t *>(hd + kHeadWords)[threadIdx.x] = (uint16_t)b;
}

// ------------------------------------------------------------------------------------------------- K1'
// One 256-lane workgroup (4 waves) per chunk.  Pass A is sequential in position order only among positions that share a
// hash, so the hash space is split four ways: every wave sweeps all positions but counts only the hashes with
// (h & 3) == its wave number -- four sequential sweeps run side by side on the four SIMDs with a quarter of the
// duplicate-resolution work each.  Passes B and C are plainly parallel over the 256 lanes.
#ifndef ZGPU_SORT_WAVES
#define ZGPU_SORT_WAVES 8
#endif
constexpr uint32_t kSortWaves = ZGPU_SORT_WAVES, kSortThreads = 64 * kSortWaves; // waves per chunk: 4 or 8 (or 16)

__global__ void __launch_bounds__(kSortThreads) sort_kernel(ChunkGeom g, uint16_t *__restrict__ S_all, uint16_t *__restrict__ rank_all, uint32_t *__restrict__ heads_all,
                                                            uint32_t *__restrict__ ir_all)
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
    uint16_t *S = S_all + (size_t)c * kSStride + kSPad, *rk = rank_all + (size_t)c * kChunkMax;
    uint32_t *hd = heads_all + (size_t)c * kHeadStride, *ir = ir_all + (size_t)c * kChunkMax;
    if (tid < kSPad) S[-(int)kSPad + (int)tid] = 0;
    for (uint32_t i = tid; i < kChunkMax / 32; i += kSortThreads) hd[i] = 0; // (pass C sets bits; the barriers of pass A lie in between)
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
            const bool out = g.nexcl != 0 && p < npos && excl_has(g, lo + p); // (continuous stream: a position in front of an earlier flush point is in no chain)
            const bool live = p < npos && !out && (h & (kSortWaves - 1)) == wave; // this wave owns its share of the hash space
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
            else if ((p >= npos || out) && wave == 0) out_stage[o] = out ? 0xffffu : 0;
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
                if (r == 0xffffu) continue; // not in the chains
                S[id] = (uint16_t)p;
                ir[p] = id | (r << 16);
                if (r == 0) atomicOr(&hd[id >> 5], 1u << (id & 31u)); // bucket head
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------- K1''
// The default sort.  What makes a counting sort that keeps equal keys in position order sequential is the ranking,
// rank(p) = count[h(p)]++ taken in position order.  On this hardware one LDS atomic instruction serves the lanes that
// hit the same address in ascending lane order (scripts/micro/lds_atomic_order.hip: no exception in 5e10 lane-operations),
// and the LDS serves the instructions of a CU in arrival order.  So 64 positions are ranked by ONE ds_add_rtn_u32 (two
// 16-bit counts per word; the add is 1 or 1<<16), and all that has to be sequenced is the order in which the 16 waves of
// the workgroup issue their instructions: a token in LDS walks round the waves, a turn is 8 instructions (512 positions)
// followed by the token store -- same wave, same queue, so the next holder's atomics arrive behind them.  Hashes are
// computed, and ranks stored, outside the turn.
// The lane order inside an atomic is an observed property, not an architected one: pass V checks the result (a bucket
// must ascend), and a violation makes the engine redo the call with sort_kernel above, which relies on ballots only.
// Output: S (positions, u16) and one bit per S index that marks the first entry of a bucket -- the rank of an entry in its
// bucket, which bounds its chain, is its distance from the last marked index.
#ifdef ZGPU_S3_STOP // timing builds only (scripts/sweep_variants.sh): leave after phase N with a trivially valid S (no candidates anywhere)
#define S3_STOP(N) do { if (ZGPU_S3_STOP == N) { for (uint32_t i_ = tid; i_ < kChunkMax; i_ += kS3Threads) { S[i_] = (uint16_t)i_; if (i_ < kHeadStride) hd[i_] = ~0u; } return; } } while (0)
#else
#define S3_STOP(N) do { } while (0)
#endif
#ifndef ZGPU_S3_BATCH
#define ZGPU_S3_BATCH 8 // positions a lane has in flight in the index and scatter passes (A/B builds: scripts/build_variant.sh NAME -DZGPU_S3_BATCH=n)
#endif
constexpr uint32_t kS3Threads = 1024, kS3Waves = kS3Threads / 64, kS3TurnSteps = 8, kS3TurnPos = 64 * kS3TurnSteps, kS3Batch = ZGPU_S3_BATCH;
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
__device__ inline uint32_t lds_add_rtn32_nowait(uint32_t a, uint32_t v) { uint32_t o; asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(v) : "memory"); return o; }

__global__ void __launch_bounds__(kS3Threads) sort3_kernel(ChunkGeom g, uint16_t *__restrict__ S_all, uint16_t *__restrict__ rank_all, uint32_t *__restrict__ heads_all,
                                                           uint32_t *__restrict__ fault, uint32_t *__restrict__ ir_all, ChunkMeta *__restrict__ meta)
{
    // (the chunk's Adler-32 rides along: pass A has every byte of the chunk in a register once, and the kernel is waiting for the LDS, not for
    // the vector unit -- a kernel of its own re-read the input for 0.7 ms per 4 GiB)
    __shared__ uint32_t ad1[kS3Waves];
    __shared__ uint64_t ad2[kS3Waves];
    __shared__ __attribute__((aligned(16))) uint32_t cnt[kHashSize / 2]; // count of hash h in half (h & 1) of word h >> 1; later the bucket starts
    __shared__ uint32_t wave_tot[kS3Waves];
    __shared__ uint32_t token;
    __shared__ __attribute__((aligned(8))) uint32_t heads[kChunkMax / 32]; // bit i: S[i] is the first entry of its bucket
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    uint16_t *S = S_all + (size_t)c * kSStride + kSPad, *rk = rank_all + (size_t)c * kChunkMax;
    uint32_t *hd = heads_all + (size_t)c * kHeadStride, *ir = ir_all + (size_t)c * kChunkMax;
    if (threadIdx.x < kSPad) S[-(int)kSPad + (int)threadIdx.x] = 0; // the pad reads as "position 0" (match3's finished lanes)
    uint32_t last_of_half0 = 0;
    const uint32_t npos = n >= 3 ? n - 2 : 0, nturns = (npos + kS3TurnPos - 1) / kS3TurnPos;
    // continuous stream: positions in front of earlier flush points have three bytes but are in no chain: no rank, no place in S (marker 0xffff in rk)
    const uint32_t nout = g.nexcl ? excl_lower(g, lo + npos) - excl_lower(g, lo) : 0u, nin = npos - nout;
    const uint32_t cnt_a = lds_off(cnt), tok_a = lds_off(&token);
    {
        uint4 *z = reinterpret_cast<uint4 *>(cnt);
        for (uint32_t i = tid; i < kHashSize * 2 / 16; i += kS3Threads) z[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0) token = 0;
        for (uint32_t i = tid; i < kChunkMax / 32; i += kS3Threads) heads[i] = 0;
    }
    __syncthreads();
    S3_STOP(0);
    auto bytes3 = [&](uint32_t p) -> uint32_t { // b0 | b1<<8 | b2<<16 of position p < npos
        if (p + 4 <= n) return reinterpret_cast<const U32u *>(src + p)->v & 0xffffffu;
        return (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
    };
    auto hash_of = [](uint32_t v) { return hash3(v & 255u, (v >> 8) & 255u, v >> 16); };

    // ---- pass A: rank(p) ----
    {
        uint32_t hv[kS3TurnSteps];
        uint32_t s1 = 0, s2 = 0; // Adler: sum of this lane's bytes, and of (n - p) * byte (64 positions a lane: below 2^32)
        auto preload = [&](uint32_t T) {
#pragma unroll
            for (uint32_t u = 0; u < kS3TurnSteps; u++) {
                const uint32_t p = T * kS3TurnPos + 64 * u + lane, v = p < npos ? bytes3(p) : 0u;
                hv[u] = p < npos ? hash_of(v) : ~0u;
                if (nout && p < npos && excl_has(g, lo + p)) { hv[u] = ~0u; rk[p] = 0xffffu; }
                s1 += v & 255u; s2 += (n - p) * (v & 255u);
            }
        };
        if (wave < nturns) preload(wave);
#pragma unroll 1
        for (uint32_t T = wave; T < nturns; T += kS3Waves) {
            uint32_t aa[kS3TurnSteps], vv[kS3TurnSteps], old[kS3TurnSteps]; // nothing but the atomics happens while the token is held
#pragma unroll
            for (uint32_t u = 0; u < kS3TurnSteps; u++) {
                const bool ok = hv[u] != ~0u;
                aa[u] = cnt_a + (ok ? (hv[u] >> 1) << 2 : 0);
                vv[u] = ok ? 1u << ((hv[u] & 1u) << 4) : 0; // adding 0 is harmless
            }
            while (lds_ld32(tok_a) != T) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (uint32_t u = 0; u < kS3TurnSteps; u++) old[u] = lds_add_rtn32_nowait(aa[u], vv[u]);
            lds_st32(tok_a, T + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])::"memory");
            static_assert(kS3TurnSteps == 8, "the wait names eight results");
#pragma unroll
            for (uint32_t u = 0; u < kS3TurnSteps; u++) {
                const uint32_t p = T * kS3TurnPos + 64 * u + lane;
                if (hv[u] != ~0u) rk[p] = (uint16_t)(old[u] >> ((hv[u] & 1u) << 4));
            }
            if (T + kS3Waves < nturns) preload(T + kS3Waves);
        }
        uint64_t t2 = s2;
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); t2 += __shfl_down(t2, o); }
        if (lane == 0) { ad1[wave] = s1; ad2[wave] = t2; }
    }
    __syncthreads();
    if (tid == 0) { // A = 1 + sum b_i, B = n + sum (n - i) b_i (mod 65521); the last two bytes start no three-byte string and are added here
        uint64_t a = 1, b = n;
        for (uint32_t w = 0; w < kS3Waves; w++) { a += ad1[w]; b += ad2[w] % 65521u; }
        for (uint32_t p = npos; p < n; p++) { a += src[p]; b += (uint64_t)(n - p) * src[p]; }
        ChunkMeta &mc = meta[chunk_of(g, c)];
        mc.adler_a = (uint32_t)(a % 65521u); mc.adler_b = (uint32_t)(b % 65521u); mc.in_bytes = n;
    }
    S3_STOP(1);

    // ---- pass B: exclusive scan of the 32768 counts -> bucket starts (in place); 32 consecutive counts per lane ----
    {
        constexpr uint32_t per = kHashSize / kS3Threads, nv = per * 2 / 16; // counts and 16-byte vectors per lane
        static_assert(per % 8 == 0, "a lane scans whole 16-byte vectors");
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt) + tid * nv;
        uint32_t v[nv * 4], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < nv; i++) { const uint4 q = c4[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
#pragma unroll
        for (uint32_t i = 0; i < nv * 4; i++) sum += (v[i] & 0xffffu) + (v[i] >> 16);
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        if (lane == 63) wave_tot[wave] = x;
        __syncthreads();
        uint32_t basev = x - sum;
        for (uint32_t w = 0; w < wave; w++) basev += wave_tot[w];
#pragma unroll
        for (uint32_t i = 0; i < nv * 4; i++) {
            const uint32_t c0 = v[i] & 0xffffu, c1 = v[i] >> 16;
            v[i] = (basev & 0xffffu) | ((basev + c0) << 16);
            basev += c0 + c1;
        }
#pragma unroll
        for (uint32_t i = 0; i < nv; i++) c4[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    __syncthreads();
    S3_STOP(2);

    // ---- pass C0: idx(p) = start(hash) + rank, kept in place of the rank; a bit per bucket head ----
    const uint16_t *start = reinterpret_cast<const uint16_t *>(cnt);
    for (uint32_t i0 = 0; i0 < npos; i0 += kS3Threads * kS3Batch) {
        uint32_t bv[kS3Batch], rv[kS3Batch];
#pragma unroll
        for (uint32_t u = 0; u < kS3Batch; u++) { const uint32_t p = i0 + u * kS3Threads + tid; bv[u] = p < npos ? bytes3(p) : 0; rv[u] = p < npos ? rk[p] : 0; }
#pragma unroll
        for (uint32_t u = 0; u < kS3Batch; u++) {
            const uint32_t p = i0 + u * kS3Threads + tid;
            if (p < npos && rv[u] != 0xffffu) {
                const uint32_t id = (uint32_t)start[hash_of(bv[u])] + rv[u];
                rk[p] = (uint16_t)id;
                ir[p] = id | (rv[u] << 16);
                if (rv[u] == 0) atomicOr(&heads[id >> 5], 1u << (id & 31u));
            }
        }
    }
    __syncthreads();
    S3_STOP(3);
    for (uint32_t i = tid; i < kChunkMax / 32; i += kS3Threads) hd[i] = heads[i];
    reinterpret_cast<uint16_t *>(hd + kHeadWords)[tid] = (uint16_t)heads_below(reinterpret_cast<const unsigned long long *>(heads)[tid], tid, wave_tot);

    // ---- pass C1: the scatter itself, through LDS (a scattered 2-byte store to HBM costs a whole partial line): the half of S
    //      with idx >> 15 == half is assembled in the memory of the dead count table and written out in order ----
    uint16_t *stage = reinterpret_cast<uint16_t *>(cnt);
    bool bad = false;
    for (uint32_t half = 0; half < 2 && half * 32768u < nin; half++) {
        __syncthreads(); // the table (pass C0), or the previous half's write-out, is done with this memory
        for (uint32_t i0 = 0; i0 < npos; i0 += kS3Threads * kS3Batch) {
            uint32_t iv[kS3Batch];
#pragma unroll
            for (uint32_t u = 0; u < kS3Batch; u++) { const uint32_t p = i0 + u * kS3Threads + tid; iv[u] = p < npos ? rk[p] : ~0u; }
#pragma unroll
            for (uint32_t u = 0; u < kS3Batch; u++) if ((iv[u] >> 15) == half && iv[u] != 0xffffu) stage[iv[u] & 32767u] = (uint16_t)(i0 + u * kS3Threads + tid);
        }
        __syncthreads();
        const uint32_t cntH = nin - half * 32768u < 32768u ? nin - half * 32768u : 32768u; // entries of this half
        for (uint32_t v = tid; v * 8 < cntH; v += kS3Threads) { // 8 entries = 16 bytes per lane and step
            const uint4 q = reinterpret_cast<const uint4 *>(stage)[v];
            *reinterpret_cast<uint4 *>(S + half * 32768u + v * 8) = q; // S is 16-byte aligned (kSPad); the tail past npos is don't-care
            // pass V: inside a bucket the positions must ascend (the lane order of the LDS atomic, see above)
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            const uint32_t hb = (heads[(half * 32768u + v * 8) >> 5] >> ((v * 8) & 31u)) & 0xffu;
            uint32_t prev = v ? stage[v * 8 - 1] : (half ? last_of_half0 : 0);
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t cur = (w[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                if (v * 8 + j < cntH && !((hb >> j) & 1u) && prev >= cur) bad = true;
                prev = cur;
            }
        }
        if (half == 0 && nin > 32768u) { __syncthreads(); last_of_half0 = stage[32767]; } // (uniform branch)
    }
    if (bad) atomicOr(fault, 1u);
}

// ------------------------------------------------------------------------------------------------- K1c (round 4)
// sort3_kernel with the ranks kept where they are made.  sort3 writes every position's rank to HBM in pass A (rk[]), reads it back and rewrites it as the
// index in pass C0 (which also reads the input a second time for the hashes), and reads the index twice more in pass C1: 43 GB of the sort's 77 GB per 4 GiB,
// for a scratch array that exists only because a lane's passes work on different positions.  Here a lane keeps ITS positions through all passes -- the 64 of
// pass A's turns (turn T = wave + 16 j, step u: position 512 T + 64 u + lane) -- one register each: hash, then hash | rank << 16, then the position's `ir` word
// (index | rank << 16), from which pass C1 scatters.  The input is read once, rk[] is not touched, what goes to HBM is S and ir.  Same token round, same
// self-check, same results (tests/test_gpu_deflate.py runs every golden vector through it; ZGPU_SORT=3 is the older kernel).
__global__ void __launch_bounds__(kS3Threads) sort4_kernel(ChunkGeom g, uint16_t *__restrict__ S_all, uint32_t *__restrict__ heads_all, uint32_t *__restrict__ fault,
                                                           uint32_t *__restrict__ ir_all, ChunkMeta *__restrict__ meta)
{
    __shared__ uint32_t ad1[kS3Waves];
    __shared__ uint64_t ad2[kS3Waves];
    __shared__ __attribute__((aligned(16))) uint32_t cnt[kHashSize / 2];
    __shared__ uint32_t wave_tot[kS3Waves];
    __shared__ uint32_t token;
    __shared__ __attribute__((aligned(8))) uint32_t heads[kChunkMax / 32];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    uint16_t *S = S_all + (size_t)c * kSStride + kSPad;
    uint32_t *hd = heads_all + (size_t)c * kHeadStride, *ir = ir_all + (size_t)c * kChunkMax;
    if (threadIdx.x < kSPad) S[-(int)kSPad + (int)threadIdx.x] = 0;
    uint32_t last_of_half0 = 0;
    const uint32_t npos = n >= 3 ? n - 2 : 0, nturns = (npos + kS3TurnPos - 1) / kS3TurnPos;
    const uint32_t nout = g.nexcl ? excl_lower(g, lo + npos) - excl_lower(g, lo) : 0u, nin = npos - nout;
    const uint32_t cnt_a = lds_off(cnt), tok_a = lds_off(&token);
    {
        uint4 *z = reinterpret_cast<uint4 *>(cnt);
        for (uint32_t i = tid; i < kHashSize * 2 / 16; i += kS3Threads) z[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0) token = 0;
        for (uint32_t i = tid; i < kChunkMax / 32; i += kS3Threads) heads[i] = 0;
    }
    __syncthreads();
    auto bytes3 = [&](uint32_t p) -> uint32_t {
        if (p + 4 <= n) return reinterpret_cast<const U32u *>(src + p)->v & 0xffffffu;
        return (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
    };
    auto hash_of = [](uint32_t v) { return hash3(v & 255u, (v >> 8) & 255u, v >> 16); };
    constexpr uint32_t kTurnsPerWave = kChunkMax / kS3TurnPos / kS3Waves; // 8
    constexpr uint32_t kNone32 = 0xffffffffu;
    uint32_t pk[kTurnsPerWave][kS3TurnSteps]; // this lane's 64 positions: hash -> hash | rank << 16 -> index | rank << 16; kNone32: no position, or not in the chains

    // ---- pass A: rank(p) ----
    {
        uint32_t s1 = 0, s2 = 0;
        auto preload = [&](uint32_t T, uint32_t (&h)[kS3TurnSteps]) {
#pragma unroll
            for (uint32_t u = 0; u < kS3TurnSteps; u++) {
                const uint32_t p = T * kS3TurnPos + 64 * u + lane, v = p < npos ? bytes3(p) : 0u;
                h[u] = p < npos ? hash_of(v) : kNone32;
                if (nout && p < npos && excl_has(g, lo + p)) h[u] = kNone32;
                s1 += v & 255u; s2 += (n - p) * (v & 255u);
            }
        };
#pragma unroll
        for (uint32_t j = 0; j < kTurnsPerWave; j++)
#pragma unroll
            for (uint32_t u = 0; u < kS3TurnSteps; u++) pk[j][u] = kNone32;
        if (wave < nturns) preload(wave, pk[0]);
#pragma unroll
        for (uint32_t j = 0; j < kTurnsPerWave; j++) {
            const uint32_t T = wave + j * kS3Waves;
            __builtin_amdgcn_sched_barrier(0);
            if (T < nturns) { // (uniform)
                uint32_t aa[kS3TurnSteps], vv[kS3TurnSteps], old[kS3TurnSteps];
#pragma unroll
                for (uint32_t u = 0; u < kS3TurnSteps; u++) {
                    const bool ok = pk[j][u] != kNone32;
                    aa[u] = cnt_a + (ok ? (pk[j][u] >> 1) << 2 : 0);
                    vv[u] = ok ? 1u << ((pk[j][u] & 1u) << 4) : 0;
                }
                while (lds_ld32(tok_a) != T) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (uint32_t u = 0; u < kS3TurnSteps; u++) old[u] = lds_add_rtn32_nowait(aa[u], vv[u]);
                lds_st32(tok_a, T + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])::"memory");
#pragma unroll
                for (uint32_t u = 0; u < kS3TurnSteps; u++)
                    if (pk[j][u] != kNone32) pk[j][u] |= ((old[u] >> ((pk[j][u] & 1u) << 4)) & 0xffffu) << 16;
                if (j + 1 < kTurnsPerWave && T + kS3Waves < nturns) preload(T + kS3Waves, pk[j + 1 < kTurnsPerWave ? j + 1 : j]);
            }
        }
        uint64_t t2 = s2;
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); t2 += __shfl_down(t2, o); }
        if (lane == 0) { ad1[wave] = s1; ad2[wave] = t2; }
    }
    __syncthreads();
    if (tid == 0) {
        uint64_t a = 1, b = n;
        for (uint32_t w = 0; w < kS3Waves; w++) { a += ad1[w]; b += ad2[w] % 65521u; }
        for (uint32_t p = npos; p < n; p++) { a += src[p]; b += (uint64_t)(n - p) * src[p]; }
        ChunkMeta &mc = meta[chunk_of(g, c)];
        mc.adler_a = (uint32_t)(a % 65521u); mc.adler_b = (uint32_t)(b % 65521u); mc.in_bytes = n;
    }

    // ---- pass B: exclusive scan of the 32768 counts -> bucket starts (in place) ----
    {
        constexpr uint32_t per = kHashSize / kS3Threads, nv = per * 2 / 16;
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt) + tid * nv;
        uint32_t v[nv * 4], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < nv; i++) { const uint4 q = c4[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
#pragma unroll
        for (uint32_t i = 0; i < nv * 4; i++) sum += (v[i] & 0xffffu) + (v[i] >> 16);
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        if (lane == 63) wave_tot[wave] = x;
        __syncthreads();
        uint32_t basev = x - sum;
        for (uint32_t w = 0; w < wave; w++) basev += wave_tot[w];
#pragma unroll
        for (uint32_t i = 0; i < nv * 4; i++) {
            const uint32_t c0 = v[i] & 0xffffu, c1 = v[i] >> 16;
            v[i] = (basev & 0xffffu) | ((basev + c0) << 16);
            basev += c0 + c1;
        }
#pragma unroll
        for (uint32_t i = 0; i < nv; i++) c4[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    __syncthreads();

    // ---- pass C0: idx(p) = start(hash) + rank: the position's ir word, written and kept; a bit per bucket head ----
    const uint16_t *start = reinterpret_cast<const uint16_t *>(cnt);
#pragma unroll
    for (uint32_t j = 0; j < kTurnsPerWave; j++) {
        const uint32_t T = wave + j * kS3Waves;
        __builtin_amdgcn_sched_barrier(0); // one turn's eight at a time: all 64 in flight do not fit the registers
        if (T < nturns) {
#pragma unroll
            for (uint32_t u = 0; u < kS3TurnSteps; u++) {
                const uint32_t p = T * kS3TurnPos + 64 * u + lane;
                if (pk[j][u] != kNone32) {
                    const uint32_t r = pk[j][u] >> 16, id = (uint32_t)start[pk[j][u] & 0x7fffu] + r;
                    pk[j][u] = id | (r << 16);
                    ir[p] = pk[j][u];
                    if (r == 0) atomicOr(&heads[id >> 5], 1u << (id & 31u));
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < kChunkMax / 32; i += kS3Threads) hd[i] = heads[i];
    reinterpret_cast<uint16_t *>(hd + kHeadWords)[tid] = (uint16_t)heads_below(reinterpret_cast<const unsigned long long *>(heads)[tid], tid, wave_tot);

    // ---- pass C1: the scatter through LDS, half of S at a time, and the self-check (see sort3_kernel) ----
    uint16_t *stage = reinterpret_cast<uint16_t *>(cnt);
    bool bad = false;
    for (uint32_t half = 0; half < 2 && half * 32768u < nin; half++) {
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < kTurnsPerWave; j++) {
            const uint32_t T = wave + j * kS3Waves;
            __builtin_amdgcn_sched_barrier(0);
            if (T < nturns) {
#pragma unroll
                for (uint32_t u = 0; u < kS3TurnSteps; u++) {
                    asm volatile("" : "+v"(pk[j][u])); // (nothing derived from it is kept across the two halves: there are no registers for that)
                    if (pk[j][u] != kNone32 && ((pk[j][u] >> 15) & 1u) == half) stage[pk[j][u] & 32767u] = (uint16_t)(T * kS3TurnPos + 64 * u + lane);
                }
            }
        }
        __syncthreads();
        const uint32_t cntH = nin - half * 32768u < 32768u ? nin - half * 32768u : 32768u;
        for (uint32_t v = tid; v * 8 < cntH; v += kS3Threads) {
            const uint4 q = reinterpret_cast<const uint4 *>(stage)[v];
            *reinterpret_cast<uint4 *>(S + half * 32768u + v * 8) = q;
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            const uint32_t hb = (heads[(half * 32768u + v * 8) >> 5] >> ((v * 8) & 31u)) & 0xffu;
            uint32_t prev = v ? stage[v * 8 - 1] : (half ? last_of_half0 : 0);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t cur = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                if (v * 8 + k < cntH && !((hb >> k) & 1u) && prev >= cur) bad = true;
                prev = cur;
            }
        }
        if (half == 0 && nin > 32768u) { __syncthreads(); last_of_half0 = stage[32767]; }
    }
    if (bad) atomicOr(fault, 1u);
}

// ------------------------------------------------------------------------------------------------- K2'
__device__ inline uint32_t lds_ld32u(uint32_t a) // 4 bytes at any LDS byte offset: aligned ds_read2_b32 + v_alignbyte
{
    uint64_t v; const uint32_t al = a & ~3u;
    asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(al) : "memory");
    return __builtin_amdgcn_alignbyte((uint32_t)(v >> 32), (uint32_t)v, a & 3);
}
// xa, xb = the first and the second four bytes of (8 bytes at LDS offset a) ^ (8 bytes at LDS offset b), any alignment: three aligned
// dwords of each string in flight together, one wait
__device__ inline void lds_cmp8(uint32_t a, uint32_t b, uint32_t &xa, uint32_t &xb)
{
    uint64_t va, vb; uint32_t va2, vb2;
    const uint32_t aa = a & ~3u, ba = b & ~3u;
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read_b32 %1, %4 offset:8\n\tds_read2_b32 %2, %5 offset1:1\n\tds_read_b32 %3, %5 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(va), "=&v"(va2), "=&v"(vb), "=&v"(vb2) : "v"(aa), "v"(ba) : "memory");
    const uint32_t a0 = __builtin_amdgcn_alignbyte((uint32_t)(va >> 32), (uint32_t)va, a & 3), a1 = __builtin_amdgcn_alignbyte(va2, (uint32_t)(va >> 32), a & 3);
    const uint32_t b0 = __builtin_amdgcn_alignbyte((uint32_t)(vb >> 32), (uint32_t)vb, b & 3), b1 = __builtin_amdgcn_alignbyte(vb2, (uint32_t)(vb >> 32), b & 3);
    xa = a0 ^ b0; xb = a1 ^ b1;
}
__device__ inline void lds_ld2bytes(uint32_t a, uint32_t &b0, uint32_t &b1)
{
    asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %2 offset:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(a) : "memory");
}
struct __attribute__((packed, aligned(2))) U64u { uint64_t v; };
struct __attribute__((packed, aligned(2))) U128u { uint4 v; }; // 8 bytes at 2-byte alignment: one global_load_dwordx2 on gfx950 (scripts/micro/global_unaligned.hip)
constexpr uint32_t kM2Threads = 1024;

// chunk bytes -> LDS (zero padded to kChunkMax + 64), by all T lanes of the workgroup
template <uint32_t T> __device__ inline void stage_chunk(const uint8_t *src, uint32_t n, uint32_t *d32, uint32_t tid)
{
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *s128 = reinterpret_cast<const uint4 *>(src);
        uint4 *d128 = reinterpret_cast<uint4 *>(d32);
        const uint32_t nv = n >> 4;
        for (uint32_t i = tid; i < (kChunkMax + 64) / 16; i += T) {
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
        for (uint32_t i = tid; i < (kChunkMax + 64) / 4; i += T) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < 4; k++) { uint32_t a = (i << 2) + k; if (a < n) v |= (uint32_t)src[a] << (8 * (k & 3)); }
            d32[i] = v;
        }
    }
}

// Lane masks straight from a vector compare (SGPR pair).  `__ballot(a && b)` of two conditions goes through
// v_cndmask + v_cmp_ne to rebuild a mask the compares had already produced; the walk step below keeps the set of lanes
// still walking as such a mask and combines compare results with scalar ANDs.
__device__ inline unsigned long long mask_eq_u32(uint32_t a, uint32_t b) { unsigned long long m; asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ inline unsigned long long mask_le_i32(int a, int b) { unsigned long long m; asm volatile("v_cmp_le_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ inline unsigned long long mask_gt_u32(uint32_t a, uint32_t b) { unsigned long long m; asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ inline unsigned long long mask_lt_i32(int a, int b) { unsigned long long m; asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
// the lanes of m append `entry` to the wave's stack at LDS byte offset rt, in lane order; the other lanes store to a dummy
// word instead (narrowing exec for the store costs more than the select: writes to exec stall the vector pipe)
__device__ inline void stack_push(unsigned long long m, uint32_t rt, uint32_t dummy, uint32_t entry)
{
    uint32_t t;
    asm volatile("v_mbcnt_lo_u32_b32 %0, %2, 0\n\tv_mbcnt_hi_u32_b32 %0, %3, %0\n\tv_lshl_add_u32 %0, %0, 2, %4\n\t"
                 "v_cndmask_b32_e64 %0, %5, %0, %1\n\tds_write_b32 %0, %6"
                 : "=&v"(t) : "s"(m), "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "s"(rt), "v"(dummy), "v"(entry) : "memory");
}

// LDS byte offset of the slot the lanes of m append to a stack whose top is at byte offset rt, in lane order (the other lanes get a slot they must not use)
__device__ inline uint32_t stack_slot(unsigned long long m, uint32_t rt)
{
    uint32_t t;
    asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0\n\tv_lshl_add_u32 %0, %0, 2, %3" : "=&v"(t) : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "s"(rt));
    return t;
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
#define ZGPU_M3_PERIOD 128 // steps between forced folds (keeps the walkers' best length fresh); a power of two <= 256
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

__global__ void __launch_bounds__(kM2Threads, 8) match3_kernel(ChunkGeom g, LevelCfg cfg, const uint16_t *__restrict__ S_all, const uint32_t *__restrict__ heads_all,
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
    const uint16_t *S = S_all + (size_t)c * kSStride + kSPad;
    const unsigned long long *hd = reinterpret_cast<const unsigned long long *>(heads_all + (size_t)c * kHeadStride); // bit i: S[i] starts a bucket
    uint2 *rec = recs + (size_t)c * kChunkMax;
    const uint32_t npos = n >= 3 ? n - 2 : 0;
    const uint32_t base = chunk_base(g, c);
    stage_chunk<kM2Threads>(src, n, d32, tid);
    if (tid == 0) *work_next = 0;
    for (uint32_t q2 = npos + tid; q2 < n; q2 += kM2Threads) rec[q2] = make_uint2((uint32_t)src[q2] << 24, 0); // the last two positions carry no hash
    __syncthreads();

    const uint32_t chainF = cfg.chain, chainQ = cfg.chain >> 2, dbase = lds_off(d32), nblk = (npos + 63) >> 6;
    const uint16_t *hb = reinterpret_cast<const uint16_t *>(heads_all + (size_t)c * kHeadStride + kHeadWords); // heads_below per block
    auto grab = [&]() { uint32_t b = 0; if (lane == 0) b = atomicAdd(work_next, 1u); return (uint32_t)__builtin_amdgcn_readfirstlane(b); };
    // the next block's entry, head bits and heads_below are fetched while the current block is searched
    uint32_t blk = grab(), p_nx = 0, below_nx = 0;
    unsigned long long m_nx = 0;
    if (blk < nblk) { const uint32_t wi = (blk << 6) + lane; p_nx = wi < npos ? S[wi] : 0; m_nx = hd[blk]; below_nx = hb[blk]; }
    while (blk < nblk) {
        const uint32_t wi = (blk << 6) + lane;
        const bool valid = wi < npos;
        const uint32_t p = p_nx, below = below_nx;
        const unsigned long long m = m_nx;
        const uint32_t blk_nx = grab();
        if (blk_nx < nblk) { const uint32_t wn = (blk_nx << 6) + lane; p_nx = wn < npos ? S[wn] : 0; m_nx = hd[blk_nx]; below_nx = hb[blk_nx]; }
        uint32_t avail = 0;
        {   // rank of an entry in its bucket = distance from the last bucket head at or below it; only min(rank, chain) matters
            const unsigned long long mine = m & ((2ull << lane) - 1);
            const uint32_t rank = mine ? (uint32_t)__builtin_clzll(mine) - (63 - lane) : lane + below;
            if (valid) avail = rank < chainF ? rank : chainF;
        }
        const int w = (int)(p + base);
        int thr = (w - (int)kMaxDist > 1 ? w - (int)kMaxDist : 1) - (int)base;                 // first candidate: dist <= MAX_DIST, not NIL
        const int thr_next = (w - (int)kMaxDist + 1 > 1 ? w - (int)kMaxDist + 1 : 1) - (int)base; // later ones: strictly inside
        if (cfg.strategy == kRle && thr < (int)p - 1) thr = (int)p - 1; // Z_RLE: the head of the chain counts only at distance 1 (deflate.c:1596)
        uint32_t best = kMinMatch - 1, key_seen = 0, snapkey = 0;
        uint32_t scan2 = (uint32_t)d8[p + 1] | ((uint32_t)d8[p + 2] << 8);
        uint32_t boff = dbase + best - 1;
        unsigned long long amask = mask_gt_u32(avail, 0u); // lanes still walking (wave-uniform)
        bool snap_taken = false;
        const uint16_t *sp = S + wi; // candidate k is sp[-1-k]
        lds_st32(slot + lane * 4, 0);
        lds_st16(pw + lane * 2, p);
        uint32_t tail = 0; // entries on the ring (wave-uniform)
        const uint32_t lanebits = lane << 16, dummy = ring + (kRing - 1) * 4; // (the stack never holds more than 127 entries: its last word is free)
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
                best = (key >> 31) ? key & 0x1ffu : key >> 16;
                boff = dbase + best - 1;
                scan2 = (uint32_t)d8[p + best - 1] | ((uint32_t)d8[p + best] << 8);
            }
            amask &= ~mask_lt_i32((int)key, 0); // a nice_match key: the walk of that lane is over (deflate.c:1224)
        };

        // four candidates per load, the nearest in the top 16 bits; the index is clamped so that lanes whose walk is over load
        // something harmless instead of being masked off
        // ... and lanes whose walk is over read the zeroed pad in front of S: candidate "position 0" for all of them, so their
        // (unused) quick-check reads fall on a handful of LDS words instead of 64 random ones (the LDS pipe is 80 % busy)
        auto group = [&](uint32_t k0) -> uint64_t {
            int gi = (int)wi - 4 - (int)k0, gsel;
            gi = gi < -(int)kSPad ? -(int)kSPad : gi;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(gsel) : "v"(-(int)kSPad), "v"(gi), "s"(amask));
            return reinterpret_cast<const U64u *>(S + gsel)->v;
        };
        uint64_t cq = group(0);
        uint32_t k = 0;
        for (;; k += 4) {
            if (amask == 0) break;
            if (k == chainQ) {
                while (tail) fold(k);
                snapkey = key_seen; snap_taken = true;
            } else if ((k & (ZGPU_M3_PERIOD - 1)) == 0) { while (tail) fold(k); }
            const uint64_t cqn = group(k + 4);
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                // no divergent control flow in the step: every lane reads (idle ones a harmless in-range address), masks decide
                const uint32_t q = (uint32_t)(cq >> (48 - 16 * j)) & 0xffffu;
                uint32_t b0, b1;
                lds_ld2bytes(boff + q, b0, b1);
                amask &= mask_le_i32(thr, (int)q); // beyond MAX_DIST (or the NIL position): the chain ends here (deflate.c:1163)
                if (k == 0 && j == 0) thr = thr_next;
                const unsigned long long m = amask & mask_eq_u32(b0 | (b1 << 8), scan2);
                M3_STAT(0, 1); M3_STAT(1, __popcll(amask)); M3_STAT(2, __popcll(m));
                if (m) {
                    stack_push(m, (uint32_t)__builtin_amdgcn_readfirstlane(ring + (tail << 2)), dummy, q | lanebits | ((k + j) << 22));
                    tail = (uint32_t)__builtin_amdgcn_readfirstlane(tail + (uint32_t)__popcll(m));
                    if (tail >= 64) fold(k + j);
                }
                amask &= mask_gt_u32(avail, k + j + 1);
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
        blk = blk_nx;
    }
}

// ------------------------------------------------------------------------------------------------- K2w
// Parse-driven search (levels 4-9; replaces match3_kernel + the game stage of parse2_kernel).  match3 searches every
// position with the full chain budget because the parse that decides which searches matter runs after it; deflate_slow
// itself (deflate.c:1554-1674) calls longest_match at a quarter of the positions, often with a quarter of the budget, and
// walks a sixth of the candidates (tests/tools/walk_model.c).  Here the parse drives the search:
//
//   * The chunk is cut into blocks of 64 positions; a WALKER (one lane) takes a block and runs the reference's loop from the
//     block's first position in the neutral state (nothing in hand, prev_length = MIN_MATCH-1), calling for searches as the
//     loop does: at a neutral position with the full budget, at the position behind a match shorter than max_lazy with the
//     match as the seed (and a quarter of the budget from good_match on).
//   * Whatever happens from a neutral position depends on that position alone, so a walker that arrives, neutral, at a
//     position another walker has been at in the same state stops there (one bit per position in LDS, claimed with an atomic
//     OR): every neutral position is worked on once, and parses started 64 positions apart fall into step after a match or
//     two (0.8 neutral positions on average, tests/tools/walk_model.c).  The walkers' paths form a forest in which the path
//     from position 0 -- the reference's parse -- is complete.  A walker whose path has merged takes the next block.
//   * A lane searches like a lane of match3 (same step: two-byte quick check at the best length, candidates that pass are
//     parked on the wave's stack, 64 parked candidates are compared at full occupancy and folded into their owners' keys),
//     but the lanes of a wave are at different points of different searches.  They advance in bodies of four candidates; a
//     lane whose search is over waits until the wave's next PASS (when 16 lanes wait, or none is searching), which plays
//     the parse for all of them, claims positions, hands out blocks and starts the next searches.  Everything a start needs
//     from memory (index and rank of the position: one dword of `ir`) is fetched at least one pass ahead, for the position
//     behind the current one and for the position behind the match in hand.
//   * Output: for every neutral position r that a game started from, gm[r] = (m - r) << 24 | len << 15 | dist of the match the
//     game ends with, and bit r of the chunk's bitmap gs: what parse2_kernel's stage A1 derives from match3's records,
//     restricted to the positions some walker stood on (which include the whole path).  parse2 (lite form) does the rest.
#ifndef ZGPU_WTRIG
#define ZGPU_WTRIG 48 // lanes waiting for a pass that make the wave run one
#endif
#ifndef ZGPU_WBLK
#define ZGPU_WBLK 64 // positions per block
#endif
#ifndef ZGPU_WTHREADS
#define ZGPU_WTHREADS 512
#endif
#ifndef ZGPU_WNEU_SHIFT
#define ZGPU_WNEU_SHIFT 0 // walkers meet at positions that are multiples of 1 << this (fewer bits in LDS, a little more duplicate work)
#endif
#ifndef ZGPU_WFOLD_AT
#define ZGPU_WFOLD_AT 64 // parked candidates that make a wave compare them (at most 64 more arrive with one step: the stack holds 128)
#endif
constexpr uint32_t kWFoldAt = ZGPU_WFOLD_AT;
constexpr uint32_t kWThreads = ZGPU_WTHREADS, kWWaves = kWThreads / 64, kWBlk = ZGPU_WBLK, kWTrig = ZGPU_WTRIG, kWNeuShift = ZGPU_WNEU_SHIFT;
constexpr uint32_t kWNeuBytes = (kChunkMax >> kWNeuShift) / 8;
constexpr uint32_t kWLds = kM3DataLds + 16 + kWNeuBytes + kWWaves * kM3WaveLds;
static_assert(2 * kWLds <= 160 * 1024, "two walker workgroups per CU");
enum : uint32_t { W_NEED = 0, W_LIMBO = 1, W_READY = 2, W_SEARCH = 3, W_DONE = 4 };
#ifdef ZGPU_WALK_STATS // debug build only: 0 bodies, 1 active lane-steps, 2 passes, 3 lanes served by passes, 4 searches, 5 folds, 6 parked, 7 limbo starts
__device__ unsigned long long walk_stats[8];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_walk_stats(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(walk_stats), sizeof z);
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(walk_stats), z, sizeof z);
}
#define W_STAT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&walk_stats[i], v_); } while (0)
#else
#define W_STAT(i, v) do { } while (0)
#endif
#ifdef ZGPU_WALK_TIME // debug build only (scripts/walk_time.py): clock cycles of a wave in 0 passes (rest), 1 bodies (without 2), 2 folds inside bodies, 3 setup, 4 drain folds, 5 parse + claims, 6 blocks
__device__ unsigned long long walk_time[8];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_walk_time(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(walk_time), sizeof z);
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(walk_time), z, sizeof z);
}
// (summed in registers, written once when the wave is done: an atomic per section would sit in the wave's memory counter and be waited for
// by the next section that needs a load)
#define W_T0() unsigned long long wt_prev = __builtin_readcyclecounter(), wt_fold = 0, wt_lds = 0, wt_take = 0, wt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define W_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); wt_acc[i] += t_ - wt_prev - ((i) == 1 ? wt_fold + wt_lds + wt_take : 0); \
                    if ((i) == 1) { wt_acc[2] += wt_fold; wt_acc[3] += wt_lds; wt_acc[7] += wt_take; } wt_fold = wt_lds = wt_take = 0; wt_prev = t_; } while (0)
#define W_TF(stmt) do { const unsigned long long f0_ = __builtin_readcyclecounter(); stmt; wt_fold += __builtin_readcyclecounter() - f0_; } while (0)
#define W_TL(stmt) do { const unsigned long long f0_ = __builtin_readcyclecounter(); stmt; wt_lds += __builtin_readcyclecounter() - f0_; } while (0)   // the body's batch of byte reads
#define W_TK(stmt) do { const unsigned long long f0_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stmt; wt_take += __builtin_readcyclecounter() - f0_; } while (0) // waiting for the groups a pass asked for
#define W_TEND() do { if (lane == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&walk_time[i_], wt_acc[i_]); } while (0)
#else
#define W_T0() do { } while (0)
#define W_T(i) do { } while (0)
#define W_TF(stmt) stmt
#define W_TL(stmt) stmt
#define W_TK(stmt) stmt
#define W_TEND() do { } while (0)
#endif
// b + the high / low half of w, b | the high half of w: one instruction each (SDWA picks the half), b | (w & m)
__device__ inline uint32_t add_w1(uint32_t b, uint32_t w) { uint32_t r; asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(b), "v"(w)); return r; }
__device__ inline uint32_t add_w0(uint32_t b, uint32_t w) { uint32_t r; asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(b), "v"(w)); return r; }
__device__ inline uint32_t or_w1(uint32_t b, uint32_t w) { uint32_t r; asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(b), "v"(w)); return r; }
__device__ inline uint32_t and_or(uint32_t w, uint32_t m, uint32_t b) { uint32_t r; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(m), "v"(b)); return r; }
__device__ inline uint32_t sel_mask(unsigned long long m, uint32_t if_set, uint32_t if_clear) { uint32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m)); return r; }

// TILE (continuous stream): where the parse leaves the tile as a function of where it enters it.  The games the walkers have played (gm, one bit per game
// start in gs) are a forest over the positions [h0, h1): a game started at r ends in the neutral position q = m + len, literals follow up to the next game
// start.  Every possible entry h0 + k is the start of a walker, so the path from it lies in the forest.  Game starts are numbered (prefix counts of gs), the
// successor of each is looked up, and pointer jumping (N[i] = N[N[i]], in place) takes every start to the exit of its path in log2(depth) rounds: no serial
// walk through the tile.  exits[k] = (neutral position the parse entered at h0 + k ends in) - h1, 0 .. 512.  T lanes, LDS from `pl` (the chunk bytes are dead).
template <uint32_t T>
__device__ inline void tile_exits(uint8_t *pl, const uint32_t *gm, const uint32_t *gs, uint32_t h0, uint32_t h1, uint32_t nent, uint16_t *exits, uint32_t tid)
{
    constexpr uint32_t kWords = kChunkMax / 32;
    static_assert(kWords % T == 0 && T % 64 == 0 && T <= 1024, "workgroup size");
    uint32_t *HAS = reinterpret_cast<uint32_t *>(pl);
    uint16_t *wcnt = reinterpret_cast<uint16_t *>(pl + kWords * 4);
    uint32_t *wtot = reinterpret_cast<uint32_t *>(pl + kWords * 6);
    uint16_t *N = reinterpret_cast<uint16_t *>(pl + kWords * 6 + 64);
    constexpr uint32_t kCapNodes = (kM3DataLds - (kWords * 6 + 64)) / 2; // game starts the table has room for
    static_assert(kCapNodes < 0x8000u && kCapNodes > kTileH1 / 3, "a path's game starts are three positions apart at least; different paths' need not be");
    const uint32_t lane = tid & 63, wave = tid >> 6;
    constexpr uint32_t per = kWords / T;
    uint32_t sum = 0, cnt[per];
#pragma unroll
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t w = tid * per + i, v = __hip_atomic_load(gs + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        HAS[w] = v; cnt[i] = (uint32_t)__builtin_popcount(v); sum += cnt[i];
    }
    uint32_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    uint32_t b = x - sum;
    for (uint32_t w = 0; w < wave; w++) b += wtot[w];
#pragma unroll
    for (uint32_t i = 0; i < per; i++) { wcnt[tid * per + i] = (uint16_t)b; b += cnt[i]; }
    __syncthreads();
    uint32_t nn = 0;
    for (uint32_t w = 0; w < T / 64; w++) nn += wtot[w]; // game starts of the tile
    auto rank_of = [&](uint32_t p) { return (uint32_t)wcnt[p >> 5] + (uint32_t)__builtin_popcount(HAS[p >> 5] & ~(~0u << (p & 31u))); };
    if (nn > kCapNodes) { // (uniform) walkers that never met -- long runs, short periods: every entry's path is walked by a lane of its own, a global load per game
        for (uint32_t k = tid; k < kTileExitStride; k += T) {
            uint32_t v = 0;
            if (k < nent) {
                const uint32_t e = h0 + k;
                uint32_t p = e < h1 ? next_bit(HAS, e, kWords) : kNone, x = e > h1 ? e : h1;
                while (p != kNone && p < h1) {
                    const uint32_t gv = __hip_atomic_load(gm + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), q = p + (gv >> 24) + ((gv >> 15) & 511u);
                    x = q > h1 ? q : h1;
                    p = q < h1 ? next_bit(HAS, q, kWords) : kNone;
                }
                v = x - h1;
            }
            exits[k] = (uint16_t)v;
        }
        return;
    }
    constexpr uint32_t kExit = 0x8000u; // N[i] = kExit | (exit - h1), or the number of the next game start on the path
    for (uint32_t p0 = h0 & ~31u; p0 < h1; p0 += T * 8) {
        uint32_t gv[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) { const uint32_t p = p0 + u * T + tid; gv[u] = p < h1 ? __hip_atomic_load(gm + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; } // (all of them: loads under a per-lane condition are waited for one by one)
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t p = p0 + u * T + tid;
            if (p < h1 && ((HAS[p >> 5] >> (p & 31u)) & 1u)) {
                const uint32_t q = p + (gv[u] >> 24) + ((gv[u] >> 15) & 511u);
                uint32_t v = kExit | (q > h1 ? q - h1 : 0u);
                if (q < h1) { const uint32_t t = next_bit(HAS, q, kWords); if (t != kNone && t < h1) v = rank_of(t); }
                N[rank_of(p)] = (uint16_t)v;
            }
        }
    }
    __syncthreads();
    for (;;) {
        bool more = false;
        for (uint32_t i = tid; i < nn; i += T) {
            const uint32_t v = N[i];
            if (!(v & kExit)) { const uint32_t w = N[v]; N[i] = (uint16_t)w; more = more || !(w & kExit); } // (another lane may move N[v] on meanwhile: old or new, both lie further down i's path)
        }
        if (!__syncthreads_or(more)) break;
    }
    for (uint32_t k = tid; k < kTileExitStride; k += T) {
        uint32_t v = 0;
        if (k < nent) {
            const uint32_t e = h0 + k, t = e < h1 ? next_bit(HAS, e, kWords) : kNone;
            v = (t == kNone || t >= h1) ? (e > h1 ? e - h1 : 0u) : ((uint32_t)N[rank_of(t)] & 0x7fffu);
        }
        exits[k] = (uint16_t)v;
    }
}

// FUSE: when its walkers are done the workgroup goes on with the rest of the parse itself (parse_chunk, zgpu_lz_parse.h, in the LDS the
// chunk bytes lived in).  That part is all latency -- a window at a time, one wave threading the path -- and leaves the CU's vector
// units to the other workgroup's walkers, which are bound by exactly those; as a kernel of its own it cost as much as a third of the walk.
static_assert(kP2LdsBytes <= kM3DataLds, "the parse works in the memory of the chunk bytes");
// MODE 0: the walkers alone; 1: the rest of the parse behind them (FUSE); 2: a TILE of a continuous stream (zgpu_cont.hip) -- walkers start at every possible
// entry of the tile and at every 64th position of its range [h0, h1), stop at h1, and the workgroup ends with the tile's exit as a function of its entry.
template <int MODE>
__global__ void __launch_bounds__(kWThreads, kWThreads / 128) walk_kernel(ChunkGeom g, LevelCfg cfg, const uint16_t *__restrict__ S_all, const uint32_t *__restrict__ ir_all,
                                                            uint32_t *__restrict__ gm_all, uint32_t *__restrict__ gs_all, uint32_t *__restrict__ tokens, ChunkMeta *meta, TileGeom tg)
{
    constexpr bool FUSE = MODE == 1, TILE = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *d32 = lds;
    uint32_t *ctrl = lds + kM3DataLds / 4; // [0]: next block to hand out
    uint32_t *NEU = ctrl + 4;              // bit p: some walker stands, or stood, at p with nothing in hand
    const uint8_t *d8 = reinterpret_cast<const uint8_t *>(d32);
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ring = (uint32_t)__builtin_amdgcn_readfirstlane(lds_off(lds) + kM3DataLds + 16 + kWNeuBytes + wave * kM3WaveLds), slot = ring + kRing * 4, pw = slot + 64 * 4;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    const uint16_t *S = S_all + (size_t)c * kSStride + kSPad;
    const uint32_t *ir = ir_all + (size_t)c * kChunkMax;
    uint32_t *gm = gm_all + (size_t)c * kChunkMax, *gs = gs_all + (size_t)c * (kChunkMax / 32);
    uint32_t th0 = 0, th1 = n, tnent = 0, tnent_all = 0, nil_local = ~0u; // TILE: the range the tile parses, its entries, the position whose first candidate at MAX_DIST is NIL
    uint32_t base = chunk_base(g, c);
    if (TILE) {
        uint64_t wb; uint32_t nl;
        tile_span(g, tg, c, wb, nl, th0, th1, tnent);
        tnent_all = tnent;
        if (tnent > th1 - th0) tnent = th1 - th0; // (walkers start inside the range only; an entry at or behind h1 passes the tile by)
        base = (tg.abs0_nil && tg.abs0 + wb == 0) ? 0u : 1u; // only the stream's own position 0 is NIL; a tile's local 0 is out of every parsed position's reach
        if (tg.nil_pos >= wb && tg.nil_pos - wb < kChunkMax) nil_local = (uint32_t)(tg.nil_pos - wb);
    }
    const uint32_t npos = n >= 3 ? n - 2 : 0;
    stage_chunk<kWThreads>(src, n, d32, tid);
    for (uint32_t i = tid; i < kChunkMax / 32; i += kWThreads) { if (i < kWNeuBytes / 4) NEU[i] = 0; gs[i] = 0; }
    if (tid < 8) d32[(kChunkMax + 64) / 4 + tid] = 0xffffffffu; // the word the quick check reads where a zero word would look like a hit
    if (tid == 0) ctrl[0] = 0;
    __syncthreads();
    int slide_at; // visited positions >= slide_at see the slid window (ParseCtx::slide_at, zgpu_lz_parse.hip)
    {
        const int room = (int)(2 * kWSize - base), b0 = (int)n < room ? (int)n : room;
        const int a = b0 - (int)kMinLookahead + 1, b = (int)(kWSize + kMaxDist) - (int)base;
        slide_at = a > b ? a : b;
    }
    // blocks: TILE -- one per entry, then one per 64 positions of the rest of the range
    const uint32_t chainF = cfg.chain, chainQ = cfg.chain >> 2, dbase = lds_off(d32), nblk = TILE ? tnent + (th1 - th0 - tnent + kWBlk - 1) / kWBlk : (n + kWBlk - 1) / kWBlk;
    const uint32_t wlim = TILE ? th1 : n; // a walker that arrives here, neutral, is done with the chunk / the tile
    const bool vexact = (chainF & 7u) != 0 || (chainQ & 7u) != 0 || cfg.strategy == kRle; // a chain may end inside a group of eight with entries of its own bucket behind it (see the bodies)
    const uint32_t lanebits = lane << 16, dummy = ring + (kRing - 1) * 4;
    // walker
    uint32_t st = W_NEED, x = 0, handL = kMinMatch - 1, handM = 0, handD = 0, gstart = 0;
    uint32_t irx = 0, irn = 0, irE = 0, irl = 0; // idx | rank << 16 of: x, x + 1, the position behind the match in hand, a position asked for in the last pass
    bool haveE = false;
    // search (as in match3_kernel); keys: len >= nice: 1<<31 | q<<9 | len, else len<<16 | q -- nearest candidate first = largest q
    uint32_t rem = 0, best = kMinMatch - 1, key_seen = 0, sentinel = 0, scan2 = 0, boff = dbase + 1;
    // candidates arrive in groups of eight (16 bytes of S).  A lane's loads are scattered over S: nothing is shared between lanes and
    // every group comes from L2 or beyond, a latency of several bodies; streaming the groups one body ahead (as match3 does with
    // its coalesced loads) left every body waiting for memory.  So a pass fetches the first 32 candidates of the searches it
    // starts at once (F0..F3; most chains are no longer), the lanes take them over one body later (G0..G3, consumed by rotation),
    // and a chain that goes on after 32 candidates is refilled by the next pass.
    int gi = -(int)kSPad; // gi: S index of the next group to fetch
    uint32_t firstb = 0; // 1 << 22 while the lane's search has not examined its first candidate (the one the reference allows at MAX_DIST exactly)
    uint4 G0 = make_uint4(0, 0, 0, 0), G1 = G0, G2 = G0, G3 = G0, F0 = G0, F1 = G0, F2 = G0, F3 = G0;
    uint32_t left = 0;  // groups in G0..G3
    bool cont = false;  // out of candidates in registers, chain not finished
    unsigned long long amask = 0, jmask = 0; // lanes walking a chain; lanes that join them when their first group has arrived
    uint32_t tail = 0;

    auto fold = [&]() {
        const uint32_t cnt = tail < 64 ? tail : 64;
        W_STAT(5, 1); W_STAT(6, cnt);
        tail = (uint32_t)__builtin_amdgcn_readfirstlane(tail - cnt);
        if (lane < cnt) {
            const uint32_t e = lds_ld32(ring + ((tail + lane) << 2));
            const uint32_t q = e & 0xffffu, o = (e >> 16) & 63u;
            const uint32_t po = lds_ld16(pw + o * 2), look = n - po, cap = look < kMaxMatch ? look : kMaxMatch, nice = cfg.nice < look ? cfg.nice : look;
            // the candidate must be in reach: the first one of a chain at most MAX_DIST back (deflate.c:1588), the others strictly less
            // (deflate.c:1163), none at the NIL position.  Positions descend along a chain, so refusing them here one by one is the
            // reference's "the chain ends at the first candidate out of reach".
            const bool reach = po - q <= kMaxDist - ((e >> 22) & 1u ? 0u : 1u) && q + base >= 1u;
            uint32_t l = 0, xa, xb;
            for (;;) { // eight bytes of both strings per round trip
                lds_cmp8(dbase + q + l, dbase + po + l, xa, xb);
                if ((xa | xb) != 0 || l + 8 >= cap) break;
                l += 8;
            }
            uint32_t len = xa ? l + ((uint32_t)__builtin_ctz(xa) >> 3) : xb ? l + 4 + ((uint32_t)__builtin_ctz(xb) >> 3) : l + 8;
            len = len < cap ? len : cap;
            if (len >= kMinMatch && reach) lds_max32(slot + o * 4, len >= nice ? (0x80000000u | (q << 9) | len) : ((len << 16) | q));
        }
        const uint32_t key = lds_ld32(slot + lane * 4);
        if (key != key_seen) {
            key_seen = key;
            best = (key >> 31) ? key & 0x1ffu : key >> 16;
            boff = dbase + best - 1;
            scan2 = (uint32_t)d8[x + best - 1] | ((uint32_t)d8[x + best] << 16);
        }
        amask &= ~mask_lt_i32((int)key, 0); // nice_match: that walk is over (deflate.c:1224)
    };
    auto gload = [&](int gi) -> uint4 { gi = gi < -(int)kSPad ? -(int)kSPad : gi; return reinterpret_cast<const U128u *>(S + gi)->v; };

    W_T0();
    W_T(3);
    for (;;) {
        // ================================================= pass: the parse for every lane whose search is over =================================================
        while (tail) fold();
        W_T(4);
        if (cont && (key_seen >> 31)) cont = false; // (the drain found a nice_match: no refill)
        const bool fin = st == W_SEARCH && !(((amask | jmask) >> lane) & 1ull) && !cont;
        W_STAT(2, 1); W_STAT(3, __popcll(__builtin_amdgcn_ballot_w64(fin || st == W_LIMBO || st == W_NEED)));
        if (st == W_LIMBO) { irx = irl; st = W_READY; } // what the last pass asked for has arrived
        uint32_t y = 0, irY = 0;
        bool toN = false, haveIr = false;
        if (fin) {
            uint32_t len = kMinMatch - 1, dist = 0;
            if (key_seen != sentinel) {
                const bool nz = key_seen >> 31;
                len = nz ? key_seen & 0x1ffu : key_seen >> 16;
                dist = x - (nz ? (key_seen >> 9) & 0xffffu : key_seen & 0xffffu);
            }
            if (len <= 5 && (cfg.strategy == kFiltered || (len == kMinMatch && dist > kTooFar))) len = kMinMatch - 1; // deflate.c:1601-1611
            bool emit = false;
            if (len > handL) { // a (longer) match at x: the byte before it, if a match was in hand, becomes a literal (deflate.c:1642-1651)
                if (handL < kMinMatch) gstart = x;
                handL = len; handD = dist; handM = x;
                if (len < cfg.lazy && x + 1 < npos) { x = x + 1; irx = irn; st = W_READY; } // look one position further (deflate.c:1588)
                else emit = true;
                haveE = false; // (the position behind the match in hand has moved)
            } else if (handL >= kMinMatch) emit = true;            // the match in hand stands (deflate.c:1611-1634)
            else { y = x + 1; irY = irn; toN = haveIr = true; }    // a literal
            if (emit) {
                gm[gstart] = ((handM - gstart) << 24) | (handL << 15) | handD;
                atomicOr(&gs[gstart >> 5], 1u << (gstart & 31u));
                y = handM + handL; irY = irE; haveIr = haveE; toN = true;
                handL = kMinMatch - 1;
            }
        }
        if (toN) { // neutral at y
            st = W_NEED;
            if (y < wlim) {
                const uint32_t yc = y >> kWNeuShift, bit = 1u << (yc & 31u);
                const bool meet = (y & ((1u << kWNeuShift) - 1u)) == 0; // (elsewhere walkers pass each other unseen: the same work twice, the same result)
                if (!meet || !(atomicOr(&NEU[yc >> 5], bit) & bit)) { // nobody has been here: go on
                    x = y;
                    if (haveIr) { irx = irY; st = W_READY; }
                    else { irl = y < npos ? ir[y] : 0; st = W_LIMBO; W_STAT(7, 1); }
                }
            }
        }
        W_T(5);
        {   // blocks for the walkers that have none
            const unsigned long long nm = __builtin_amdgcn_ballot_w64(st == W_NEED);
            if (nm) {
                const int first = __builtin_ctzll(nm);
                uint32_t b0 = 0;
                if ((int)lane == first) b0 = atomicAdd(&ctrl[0], (uint32_t)__popcll(nm));
                b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, first);
                if (st == W_NEED) {
                    const uint32_t b = b0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0));
                    if (b >= nblk) st = W_DONE;
                    else {
                        const uint32_t y2 = TILE ? (b < tnent ? th0 + b : th0 + tnent + (b - tnent) * kWBlk) : b * kWBlk, bit = 1u << ((y2 >> kWNeuShift) & 31u);
                        if (!(atomicOr(&NEU[(y2 >> kWNeuShift) >> 5], bit) & bit)) { x = y2; handL = kMinMatch - 1; irl = y2 < npos ? ir[y2] : 0; st = W_LIMBO; }
                        // (else: a walker from further down passed through here; the next pass asks for another block)
                    }
                }
            }
        }
        W_T(6);
        bool fresh = false;
        if (st == W_READY) { // start the search at x with the match in hand (length handL, 2: none) as the seed
            const uint32_t seed = handL, idx = irx & 0xffffu, rank = irx >> 16, budget = seed >= cfg.good ? (chainQ ? chainQ : 0xffffu) : chainF; // (a quarter of fewer than 4 never runs out, deflate.c:1163)
            uint32_t avail = (x < npos && seed < cfg.lazy) ? (rank < budget ? rank : budget) : 0; // (deflate.c:1588: no search once the match in hand reaches max_lazy)
            irn = x + 1 < npos ? ir[x + 1] : 0;
            if (seed >= kMinMatch) { const uint32_t E = handM + handL; irE = E < npos ? ir[E] : 0; haveE = true; }
            // the one position whose first candidate can sit at window index 32768: NIL after the slide (deflate.c:1309-1312)
            if (TILE) {
                if (x == nil_local && avail != 0 && x - (uint32_t)S[idx - 1] == kMaxDist) avail = 0;
                if (cfg.strategy == kHuffmanOnly) avail = 0;                                                            // deflate.c:1594: no search at all
                if (cfg.strategy == kRle && avail != 0) avail = x - (uint32_t)S[idx - 1] == 1 ? 1u : 0u;                // deflate.c:1596-1599: the nearest candidate, at distance 1 only
            }
            else if (x + base == kWSize + kMaxDist && avail != 0 && (int)x >= slide_at && (uint32_t)S[idx - 1] + base == kWSize) avail = 0;
            best = seed; sentinel = (seed << 16) | 0xffffu; key_seen = sentinel; boff = dbase + best - 1;
            scan2 = (uint32_t)d8[x + best - 1] | ((uint32_t)d8[x + best] << 16);
            lds_st32(slot + lane * 4, sentinel);
            lds_st16(pw + lane * 2, x);
            rem = avail; firstb = 1u << 22;
            gi = (int)idx - 8;
            st = W_SEARCH;
            fresh = avail != 0;
            W_STAT(4, 1);
        }
        if (cont) { cont = false; fresh = true; } // refill
        jmask = __builtin_amdgcn_ballot_w64(fresh);
        if (fresh) { // only the lanes that start or refill load, and only the groups the chain has (a scattered 16-byte load is 64 requests to the
                     // address unit whatever the lanes ask for; F0..F3 are read back by these lanes alone)
            F0 = gload(gi);
            if (rem > 8) F1 = gload(gi - 8);
            if (rem > 16) F2 = gload(gi - 16);
            if (rem > 24) F3 = gload(gi - 24);
            gi -= 32;
        }
        W_T(0);
        if (__builtin_amdgcn_ballot_w64(st != W_DONE) == 0) { W_TEND(); break; }
        const unsigned long long smask = __builtin_amdgcn_ballot_w64(st == W_SEARCH);
        const uint32_t idle = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(st == W_LIMBO || st == W_NEED));

        // ================================================= bodies of eight candidates until enough lanes wait =================================================
        for (bool first = true;; first = false) {
            const unsigned long long amask0 = amask; // the lanes that examine a group in this body
            if (amask) {
                W_STAT(0, 1);
                // the quick-check bytes of all eight candidates in one round trip, read at the best length the body starts with: a fold
                // inside the body moves best/boff/scan2 for the next body, the rest of this one goes on with what it read (walking with a
                // stale best length is exact, see match3_kernel).  A lane examines nv = min(rem, 8) candidates; for the others of the
                // group -- and for all eight in lanes that examine nothing -- it reads a fixed word whose two bytes differ from scan0, so
                // that a hit is a hit of a real candidate and no step needs the walking mask (whether a candidate is in reach is
                // checked when it is compared in full, fold()).
                // Addresses straight from the halves of G0 (SDWA: no extraction).  Which of the eight are candidates at all is NOT checked here:
                // a lane that examines nothing compares with a value no two bytes can make; a lane whose chain ends inside the group reads on into
                // the entries below its bucket, whose hashes -- so their first three bytes -- differ from the scan string's: the full comparison
                // (fold) finds fewer than MIN_MATCH bytes and drops them.  Only a chain cut short by the BUDGET goes on into entries of its own
                // bucket that must not be looked at: budgets are multiples of eight (whole groups) except at level 4, in tuned configurations and
                // for Z_RLE -- `vexact`: those take the sentinel address for the entries past the chain's end, as every body did before.
                const uint32_t scan0 = sel_mask(amask0, scan2, 0xffffffffu);
                uint32_t bb[8], a[8], boff_e;
                boff_e = boff;
                if (!vexact) { // the lanes that examine nothing read ONE common word (a broadcast): with whatever their registers held they were a third of the kernel's bank conflicts
                    const uint32_t boff_i = dbase + kChunkMax + 8u;
                    G0.x = sel_mask(amask0, G0.x, 0u); G0.y = sel_mask(amask0, G0.y, 0u); G0.z = sel_mask(amask0, G0.z, 0u); G0.w = sel_mask(amask0, G0.w, 0u);
                    boff_e = sel_mask(amask0, boff, boff_i);
                }
                a[0] = add_w1(boff_e, G0.w); a[1] = add_w0(boff_e, G0.w); a[2] = add_w1(boff_e, G0.z); a[3] = add_w0(boff_e, G0.z); // nearest candidate = highest address
                a[4] = add_w1(boff_e, G0.y); a[5] = add_w0(boff_e, G0.y); a[6] = add_w1(boff_e, G0.x); a[7] = add_w0(boff_e, G0.x);
                if (vexact) { // (uniform)
                    const uint32_t nv = rem < 8 ? rem : 8, sent = dbase + (scan2 ? kChunkMax + 8u : kChunkMax + 64u + 8u); // zero pad | the 0xFF words behind it: two bytes that differ from the scan string's
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) a[j] = j < nv ? a[j] : sent;
                }
                uint32_t hi[8]; // (d16 loads would fill both halves of one register, but with SRAM ECC on they clear the other half)
                W_TL(
                    asm volatile("ds_read_u8 %0, %16\n\tds_read_u8 %8, %16 offset:1\n\tds_read_u8 %1, %17\n\tds_read_u8 %9, %17 offset:1\n\t"
                                 "ds_read_u8 %2, %18\n\tds_read_u8 %10, %18 offset:1\n\tds_read_u8 %3, %19\n\tds_read_u8 %11, %19 offset:1\n\t"
                                 "ds_read_u8 %4, %20\n\tds_read_u8 %12, %20 offset:1\n\tds_read_u8 %5, %21\n\tds_read_u8 %13, %21 offset:1\n\t"
                                 "ds_read_u8 %6, %22\n\tds_read_u8 %14, %22 offset:1\n\tds_read_u8 %7, %23\n\tds_read_u8 %15, %23 offset:1\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(bb[0]), "=&v"(bb[1]), "=&v"(bb[2]), "=&v"(bb[3]), "=&v"(bb[4]), "=&v"(bb[5]), "=&v"(bb[6]), "=&v"(bb[7]),
                                   "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6]), "=&v"(hi[7])
                                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory"));
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) bb[j] |= hi[j] << 16;
                W_STAT(1, __popcll(amask0) * 8);
                // The hits of all eight comparisons are counted first (scalar), then parked in one straight run: no branch and no fold between
                // two candidates -- a wave's critical path through a body is what bounds the kernel, not the number of its instructions.  The stack
                // holds 127 entries: when these do not fit on top of what is parked it is folded empty first, and the (rare: runs, zeros) body with
                // more hits than the stack holds takes the candidates one by one as before.
                const uint32_t lf = lanebits | firstb;
                unsigned long long mm[8];
                uint32_t total = 0;
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) { mm[j] = mask_eq_u32(bb[j], scan0); total += (uint32_t)__popcll(mm[j]); }
                auto entry_of = [&](uint32_t j) -> uint32_t {
                    const uint32_t wd = j < 2 ? G0.w : j < 4 ? G0.z : j < 6 ? G0.y : G0.x, lb = j == 0 ? lf : lanebits;
                    return (j & 1) ? and_or(wd, 0xffffu, lb) : or_w1(lb, wd);
                };
                if (total) {
                    if (tail + total > kRing - 1) { while (tail) W_TF(fold()); }
                    if (total <= kRing - 1) {
                        // (only the lanes with a hit store: the others' stores to a common dummy word were the kernel's largest source of bank conflicts)
                        uint32_t ta[8], te[8];
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) {
                            ta[j] = stack_slot(mm[j], ring + (tail << 2)); te[j] = entry_of(j);
                            tail += (uint32_t)__popcll(mm[j]);
                        }
                        tail = (uint32_t)__builtin_amdgcn_readfirstlane(tail);
                        unsigned long long sv;
                        asm volatile("s_mov_b64 %0, exec\n\t"
                                     "s_mov_b64 exec, %17\n\tds_write_b32 %1, %9\n\ts_mov_b64 exec, %18\n\tds_write_b32 %2, %10\n\t"
                                     "s_mov_b64 exec, %19\n\tds_write_b32 %3, %11\n\ts_mov_b64 exec, %20\n\tds_write_b32 %4, %12\n\t"
                                     "s_mov_b64 exec, %21\n\tds_write_b32 %5, %13\n\ts_mov_b64 exec, %22\n\tds_write_b32 %6, %14\n\t"
                                     "s_mov_b64 exec, %23\n\tds_write_b32 %7, %15\n\ts_mov_b64 exec, %24\n\tds_write_b32 %8, %16\n\t"
                                     "s_mov_b64 exec, %0"
                                     : "=&s"(sv)
                                     : "v"(ta[0]), "v"(ta[1]), "v"(ta[2]), "v"(ta[3]), "v"(ta[4]), "v"(ta[5]), "v"(ta[6]), "v"(ta[7]),
                                       "v"(te[0]), "v"(te[1]), "v"(te[2]), "v"(te[3]), "v"(te[4]), "v"(te[5]), "v"(te[6]), "v"(te[7]),
                                       "s"(mm[0]), "s"(mm[1]), "s"(mm[2]), "s"(mm[3]), "s"(mm[4]), "s"(mm[5]), "s"(mm[6]), "s"(mm[7]) : "memory");
                    } else {
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) {
                            if (mm[j]) {
                                stack_push(mm[j], (uint32_t)__builtin_amdgcn_readfirstlane(ring + (tail << 2)), dummy, entry_of(j));
                                tail = (uint32_t)__builtin_amdgcn_readfirstlane(tail + (uint32_t)__popcll(mm[j]));
                                if (tail >= kWFoldAt) W_TF(fold());
                            }
                        }
                    }
                    while (tail >= kWFoldAt) W_TF(fold());
                }
                amask &= mask_gt_u32(rem, 8u); // lanes whose chain goes on (a fold may have ended others: nice_match)
                firstb = sel_mask(amask0, 0u, firstb);
            }
            G0 = G1; G1 = G2; G2 = G3;
            left = left ? left - 1 : 0;
            if (first && jmask) { // the searches the pass started or refilled: their 32 candidates have had a body's time to arrive
                W_TK(if ((jmask >> lane) & 1ull) { G0 = F0; G1 = F1; G2 = F2; G3 = F3; left = 4; });
            }
            rem = sel_mask(amask0, rem > 8 ? rem - 8 : 0, rem);
            {   // lanes that walk on but have nothing left in registers wait for the next pass
                const unsigned long long cm = amask & mask_eq_u32(left, 0u);
                cont = cont || ((cm >> lane) & 1ull);
                amask &= ~cm;
            }
            amask |= jmask; jmask = 0;
            if (amask == 0 || (uint32_t)__popcll(smask & ~amask) + idle >= kWTrig) break;
        }
        W_T(1);
    }
#if ZGPU_WTHREADS == 512 // (other workgroup sizes: experiments with the walkers alone, MODE 0)
    if (TILE) {
        __syncthreads(); // every walker of the tile is done: gm / gs are complete
        tile_exits<kWThreads>(reinterpret_cast<uint8_t *>(lds), gm, gs, th0, th1, tnent_all, tg.exits + (size_t)c * kTileExitStride, tid);
    }
    if (FUSE) {
        __syncthreads(); // every walker of the chunk is done: gm / gs are complete (and written: the barrier waits for the stores)
        constexpr uint32_t kP2Threads = kWThreads;
        constexpr bool LITE = true, FUSED = true, TILE = false;
        const uint2 *recs = nullptr;
        const uint32_t *gmv_all = gm_all, *gsv_all = gs_all;
        uint8_t *pl = reinterpret_cast<uint8_t *>(lds);
        uint16_t *const J = reinterpret_cast<uint16_t *>(pl + kP2OffJ);
        uint32_t *const HAS = reinterpret_cast<uint32_t *>(pl + kP2OffHAS), *const MARK = reinterpret_cast<uint32_t *>(pl + kP2OffMARK), *const COV = reinterpret_cast<uint32_t *>(pl + kP2OffCOV),
                 *const MAT = reinterpret_cast<uint32_t *>(pl + kP2OffMAT), *const wbase = reinterpret_cast<uint32_t *>(pl + kP2OffWbase), *const VIS = reinterpret_cast<uint32_t *>(pl + kP2OffVIS),
                 *const EXITS = reinterpret_cast<uint32_t *>(pl + kP2OffEXITS), *const wave_tot = reinterpret_cast<uint32_t *>(pl + kP2OffWtot);
        uint32_t &sh_entry = *reinterpret_cast<uint32_t *>(pl + kP2OffEntry), &sh_exit = *reinterpret_cast<uint32_t *>(pl + kP2OffEntry + 4);
#include "zgpu_lz_parse_body.inc"
    }
#endif
}

// ======================================================================================================================================
// Levels 1-3 -- deflate_fast (deflate.c:1448-1546) on the sorted buckets.
// deflate_fast leaves the positions inside a match longer than max_insert_length out of the hash chains, so its chains depend on its parse and
// the walkers above (whose state is the position alone) do not apply: one lane runs one chunk's loop (as zgpu_lz_serial.hip does).  What changes is
// what a step costs.  With head[]/prev[] every POSITION is a chain of dependent trips to memory (read head, write head and prev; then the search).
// Here the chain of p is its bucket predecessors S[idx-1], S[idx-2], ... (idx, rank = ir[p]) that carry a flag byte G[idx'] = "in the chains";
// inserting a position is one byte store nobody waits for, and only a TOKEN costs dependent trips: ir, then eight predecessors with their flags,
// then the candidates' bytes -- 11 600 tokens per chunk at level 1 against 65 536 positions.
// The window slide, NIL, MAX_DIST, nice_match and the chain budget are longest_match's (deflate.c:1027-1168), as in SerialLz::longest.
// ======================================================================================================================================
struct __attribute__((packed, aligned(1))) U128b { uint4 v; };
constexpr uint32_t kGPad = 32, kGStride = kChunkMax + 2 * kGPad; // flag bytes per chunk, room in front for the group reads
enum : uint32_t { F_IR = 0, F_GROUP, F_CAND, F_EXT, F_DONE };

// index of the first byte in which two 16-byte strings differ (16: none)
__device__ inline uint32_t first_diff16(uint4 a, uint4 b)
{
    const uint64_t x0 = (((uint64_t)(a.y ^ b.y)) << 32) | (a.x ^ b.x), x1 = (((uint64_t)(a.w ^ b.w)) << 32) | (a.z ^ b.z);
    return x0 ? (uint32_t)__builtin_ctzll(x0) >> 3 : x1 ? 8 + ((uint32_t)__builtin_ctzll(x1) >> 3) : 16u;
}

// 16 bytes at in + o where a 16-byte read would run past the end of the buffer (zeros behind it: lengths are capped, they never count)
__device__ __noinline__ uint4 tail16(const uint8_t *in, uint32_t o, uint64_t safe_end)
{
    uint32_t v[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16 && (uint64_t)o + k < safe_end; k++) v[k >> 2] |= (uint32_t)in[o + k] << (8 * (k & 3));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// One lane per chunk, and the lanes of a wave in step: a lane whose loop waits for memory would hold up the 63 others wherever their loops stand
// (one instruction stream), so the loop is turned inside out.  Every lane is in one of four states; a round of the wave lets every lane ask for
// up to five 16-byte pieces of memory, waits ONCE for all of them, and lets every lane take its state one step further:
//   F_IR     ir[p..p+7] and the 16 bytes at p (after a long match; the others carry them over)      -> the search is set up
//   F_GROUP  eight bucket predecessors and their flag bytes (+ ir and bytes for a fresh token)        -> up to four candidates picked
//   F_CAND   16 bytes at each candidate                                                               -> longest_match's bookkeeping, in order
//   F_EXT    16 more bytes of a candidate that matched all 16, and of the string at p
// then emits its token, sets the flags of what went into the chains, and starts the next token.  A token costs two to three rounds.
__global__ void __launch_bounds__(64) fast_kernel(ChunkGeom g, LevelCfg cfg, const uint16_t *__restrict__ S_all, const uint32_t *__restrict__ ir_all, uint8_t *G_all,
                                                   uint32_t *__restrict__ tokens, ChunkMeta *meta, uint32_t lanes)
{
    const uint32_t c = blockIdx.x * lanes + threadIdx.x;
    const bool live = threadIdx.x < lanes && c < g.nchunks;
    const uint32_t cc = live ? c : 0;
    uint64_t lo; uint32_t n;
    chunk_span(g, cc, lo, n);
    const uint8_t *in = g.in + lo;
    const uint64_t safe_end = g.in_bytes - lo; // bytes from the chunk's first to the end of the input buffer: what a 16-byte read may touch
    const uint16_t *S = S_all + (size_t)cc * kSStride + kSPad;
    const uint32_t *ir = ir_all + (size_t)cc * kChunkMax;
    uint8_t *G = G_all + (size_t)cc * kGStride + kGPad;
    uint32_t *tok = tokens + (size_t)cc * kChunkMax;
    const uint8_t *dummy = reinterpret_cast<const uint8_t *>(S_all); // what a lane reads when it has nothing to ask for (one cached line)
    const uint32_t base = chunk_base(g, cc);
    uint32_t off = 0, ntok = 0, blk_tok0 = 0, nblk = 0, nostore = 0, block_start = 0;
    auto cut_block = [&](uint32_t p_end) {
        if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk; // buf == NULL, deflate.c:1365-1367
        nblk++; blk_tok0 = ntok; block_start = p_end;
    };
    const uint32_t room = 2 * kWSize - base;
    uint32_t buffered = n < room ? n : room; // first fill_window (deflate.c:1275,1342)
    uint32_t p = 0, st = live ? F_IR : F_DONE;
    // the loop top of deflate_fast for the token that starts at p: fill_window (the slide, deflate.c:1293), the end of the input
    auto token_top = [&]() {
        if (buffered - p < kMinLookahead) {
            if ((int)(p + base) - (int)off >= (int)(kWSize + kMaxDist)) off += kWSize;
            buffered = n;
            if (n == p) { cut_block(p); st = F_DONE; } // the final block (its emission happens in the Huffman stage)
        }
    };
    if (live) token_top();
    uint32_t irw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint4 scan = make_uint4(0, 0, 0, 0), cb[4] = {scan, scan, scan, scan};
    bool fresh = false; // F_GROUP: ir and the bytes at p still have to be fetched (the token's idx came from the last token's ir)
    // the search at p
    uint32_t idx = 0, rank = 0, j0 = 0, chain = 0, best = 0, mstart = 0, cap = 0, nice = 0, look = 0, ncand = 0, cand[4] = {0, 0, 0, 0}, extl = 0;
    int w = 0, limit = 0;
    bool first = true, over = false, lastg = false;

    auto setup_search = [&](uint32_t iv) { // the token at p has at least three bytes
        idx = iv & 0xffffu; rank = iv >> 16;
        w = (int)(p + base) - (int)off; limit = w > (int)kMaxDist ? w - (int)kMaxDist : 0;
        cap = look < kMaxMatch ? look : kMaxMatch; nice = cfg.nice < look ? cfg.nice : look;
        chain = cfg.chain; best = kMinMatch - 1; first = true; over = false; lastg = false; j0 = 0; ncand = 0;
    };

    for (;;) {
        if (__builtin_amdgcn_ballot_w64(st != F_DONE) == 0) break;
        // ---- what every lane asks for: five 16-byte pieces, by state (kNone: a pointer into the workspace, else an offset into the chunk's bytes) ----
        constexpr uint32_t kNone = 0xffffffffu;
        const uint8_t *a0 = dummy, *a1 = dummy, *a2 = dummy, *a3 = dummy;
        uint32_t o0 = kNone, o1 = kNone, o2 = kNone, o3 = kNone, o4 = kNone;
        const bool want_ir = st == F_IR || (st == F_GROUP && fresh);
        if (want_ir) { a2 = reinterpret_cast<const uint8_t *>(ir + p); a3 = reinterpret_cast<const uint8_t *>(ir + p + 4); o4 = p; }
        if (st == F_GROUP) { const int at = (int)idx - (int)j0 - 8; a0 = reinterpret_cast<const uint8_t *>(S + at); a1 = G + at; } // predecessors idx-j0-8 .. idx-j0-1
        if (st == F_CAND) { o0 = cand[0]; o1 = ncand > 1 ? cand[1] : kNone; o2 = ncand > 2 ? cand[2] : kNone; o3 = ncand > 3 ? cand[3] : kNone; }
        if (st == F_EXT) { o0 = cand[0] + extl; o1 = p + extl; }
        // (a 16-byte read that would run past the input buffer -- the last bytes of the last chunk -- is made byte by byte below)
        const bool u0 = o0 != kNone && (uint64_t)o0 + 16 > safe_end, u1 = o1 != kNone && (uint64_t)o1 + 16 > safe_end, u2 = o2 != kNone && (uint64_t)o2 + 16 > safe_end,
                   u3 = o3 != kNone && (uint64_t)o3 + 16 > safe_end, u4 = o4 != kNone && (uint64_t)o4 + 16 > safe_end;
        if (o0 != kNone && !u0) a0 = in + o0;
        if (o1 != kNone && !u1) a1 = in + o1;
        if (o2 != kNone && !u2) a2 = in + o2;
        if (o3 != kNone && !u3) a3 = in + o3;
        const uint8_t *a4 = (o4 != kNone && !u4) ? in + o4 : dummy;
        uint4 r0 = reinterpret_cast<const U128b *>(a0)->v, r1 = reinterpret_cast<const U128b *>(a1)->v, r2 = reinterpret_cast<const U128b *>(a2)->v,
              r3 = reinterpret_cast<const U128b *>(a3)->v, r4 = reinterpret_cast<const U128b *>(a4)->v;
        if (__builtin_amdgcn_ballot_w64(u0 || u1 || u2 || u3 || u4)) {
            if (u0) r0 = tail16(in, o0, safe_end);
            if (u1) r1 = tail16(in, o1, safe_end);
            if (u2) r2 = tail16(in, o2, safe_end);
            if (u3) r3 = tail16(in, o3, safe_end);
            if (u4) r4 = tail16(in, o4, safe_end);
        }

        // ---- every lane one step further ----
        bool eval = false, fin = false;
        if (want_ir) {
            irw[0] = r2.x; irw[1] = r2.y; irw[2] = r2.z; irw[3] = r2.w; irw[4] = r3.x; irw[5] = r3.y; irw[6] = r3.z; irw[7] = r3.w;
            scan = r4;
        }
        if (st == F_IR) {
            look = n - p;
            if (look < kMinMatch) { first = true; fin = true; } // no INSERT_STRING, no search worth the name (deflate.c:1472: a stale hash_head cannot give three bytes)
            else { setup_search(irw[0]); if (rank == 0) fin = true; else st = F_GROUP; }
        } else if (st == F_GROUP) {
            fresh = false;
            const uint64_t fl = ((uint64_t)r1.y << 32) | r1.x;
            uint32_t used = 0;
            const uint32_t room4 = chain < 4 ? chain : 4;
            for (uint32_t t = 0; t < 8; t++) { // nearest first: the highest address
                if (ncand >= room4) break;
                if (j0 + t + 1 > rank) { lastg = true; break; }
                used = t + 1;
                const uint32_t slot = 7 - t;
                if (!((fl >> (8 * slot)) & 1u)) continue;
                const uint32_t wd = slot >= 6 ? r0.w : slot >= 4 ? r0.z : slot >= 2 ? r0.y : r0.x, q = (slot & 1) ? wd >> 16 : wd & 0xffffu;
                const int wq = (int)(q + base) - (int)off;
                bool stop = wq <= 0;                                                             // NIL, or gone with the slide
                if (first) { stop = stop || (uint32_t)(w - wq) > kMaxDist; first = first && stop; } // hash_head out of reach: no search (deflate.c:1481)
                else stop = stop || wq <= limit;                                                 // deflate.c:1163
                if (stop) { lastg = true; break; }                                               // (the chain ends behind the candidates picked so far)
                cand[3] = ncand == 3 ? q : cand[3]; cand[2] = ncand == 2 ? q : cand[2]; cand[1] = ncand == 1 ? q : cand[1]; cand[0] = ncand == 0 ? q : cand[0];
                ncand++;
            }
            j0 += used;
            if (ncand) st = F_CAND;
            else if (lastg || j0 >= rank) fin = true;
            // else: the next group
        } else if (st == F_CAND) {
            cb[0] = r0; cb[1] = r1; cb[2] = r2; cb[3] = r3;
            eval = true;
        } else if (st == F_EXT) {
            const uint32_t d = first_diff16(r0, r1);
            extl += d;
            if (d < 16 || extl >= cap) { st = F_CAND; eval = true; }
        }
        if (eval) { // longest_match's bookkeeping (deflate.c:1126-1163) for the candidates of the round, in order
            uint32_t l = st == F_CAND && extl ? extl : 0; // (back from F_EXT: the front candidate's length is known)
            bool known = extl != 0;
            extl = 0;
            while (ncand != 0) {
                if (!known) { l = first_diff16(cb[0], scan); if (l == 16 && cap > 16) { st = F_EXT; extl = 16; break; } }
                known = false;
                l = l < cap ? l : cap;
                bool end = false;
                if (l > best) { mstart = cand[0]; best = l; end = l >= nice; }
                end = end || --chain == 0;
                cand[0] = cand[1]; cand[1] = cand[2]; cand[2] = cand[3]; cb[0] = cb[1]; cb[1] = cb[2]; cb[2] = cb[3]; ncand--; // (registers cannot be indexed: the next one moves to the front)
                if (end) { fin = true; ncand = 0; }
            }
            if (st == F_CAND && !fin) { if (lastg || j0 >= rank) fin = true; else st = F_GROUP; }
        }
        if (fin) { // the token is decided: emit it, put what deflate_fast inserts into the chains, go to the next one
            uint32_t L = 1;
            const uint32_t match_len = first ? kMinMatch - 1 : (best <= look ? best : look); // (no search: nothing in hand, deflate.c:1478-1481)
            if (look >= kMinMatch) G[idx] = 1; // INSERT_STRING(strstart)
            if (match_len >= kMinMatch) {
                tok[ntok++] = tok_match(p - mstart, match_len - kMinMatch);
                if (match_len <= cfg.lazy && look - match_len >= kMinMatch) { // max_insert_length (h/deflate.h:176): the strings inside a short match go in
#pragma unroll
                    for (uint32_t k = 1; k < 7; k++) if (k < match_len) G[irw[k] & 0xffffu] = 1;
                }
                L = match_len;
            } else tok[ntok++] = tok_lit(scan.x & 255u);
            const bool cut = ntok - blk_tok0 == kBlockTokens;
            p += L;
            if (cut) cut_block(p);
            st = F_IR; extl = 0; ncand = 0;
            token_top();
            if (st != F_DONE && L < 8 && n - p >= kMinMatch) { // the next token's idx is at hand: its first round fetches predecessors and ir together
                uint32_t iv = irw[1];
#pragma unroll
                for (uint32_t k = 2; k < 8; k++) iv = L == k ? irw[k] : iv;
                if ((iv >> 16) != 0) { look = n - p; setup_search(iv); fresh = true; st = F_GROUP; }
            }
        }
    }
    if (live) { meta[c].ntok = ntok; meta[c].nostore = nostore; meta[c].in_bytes = n; }
}

// `exact_sort`: use the ballot-only sort (the engine sets it after sort3's pass V reported a fault, or ZGPU_SORT=1 asks)
// `walk`: parse-driven search (walk_kernel + the lite parse) instead of the all-position search (match3_kernel + parse2_kernel)
// `walk` 2: levels 1-3, deflate_fast on the sorted buckets (fast_kernel); `walk` 3: the same by a wave per chunk (zgpu_lz_fastwin.hip)
// returns true when the sort has left the chunks' Adler-32 in meta[] (sort3_kernel does; the ballot-only sort does not)
bool launch_lz_sorted(const ChunkGeom &g, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, void *prof, int exact_sort, int walk)
{
    bool adler_done = false;
    uint8_t *w = static_cast<uint8_t *>(workspace);
    const size_t nch = g.nchunks;
    uint32_t *fault = reinterpret_cast<uint32_t *>(w);
    uint16_t *S = reinterpret_cast<uint16_t *>(w + 256);
    uint16_t *rk = reinterpret_cast<uint16_t *>(w + 256 + ((nch * kSStride * 2 + 255) & ~(size_t)255));
    uint32_t *heads = reinterpret_cast<uint32_t *>(rk + nch * kChunkMax);
    uint2 *recs = reinterpret_cast<uint2 *>(heads + nch * kHeadStride);
    uint32_t *ir = reinterpret_cast<uint32_t *>(recs + nch * kChunkMax);
    hipEvent_t ev{};
    prof_span_begin(prof, st, &ev);
    static int sort_env = -1;
    if (sort_env < 0) { const char *e = getenv("ZGPU_SORT"); sort_env = e ? atoi(e) : 4; } // 4: sort4_kernel, 3: sort3_kernel, 1: the ballot-ranked sort
    if (exact_sort || sort_env == 1) {
        hipLaunchKernelGGL(sort_kernel, dim3(g.nchunks), dim3(kSortThreads), 0, st, g, S, rk, heads, ir);
        hipLaunchKernelGGL(heads_below_kernel, dim3(g.nchunks), dim3(1024), 0, st, heads);
    }
    else {
        if (sort_env == 3) hipLaunchKernelGGL(sort3_kernel, dim3(g.nchunks), dim3(kS3Threads), 0, st, g, S, rk, heads, fault, ir, meta);
        else hipLaunchKernelGGL(sort4_kernel, dim3(g.nchunks), dim3(kS3Threads), 0, st, g, S, heads, fault, ir, meta);
        adler_done = true;
        if (g_inject_sort_fault.exchange(0)) hipMemsetAsync(fault, 1, 4, st); // zgpu_debug_inject_sort_fault(): exercise the engine's fallback without a real fault
    }
    prof_span_end(prof, st, ZGPU_STAGE_CHAIN, ev);
    prof_span_begin(prof, st, &ev);
    if (walk == 3) {
        launch_lz_fastwin(g, cfg, S, ir, tokens, meta, st);
        prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
        return adler_done;
    }
    if (walk == 2) { // the records' memory holds the flag bytes
        uint8_t *G = reinterpret_cast<uint8_t *>(recs);
        hipMemsetAsync(G, 0, nch * kGStride, st);
        static int forced = -1; // chunks per wave (ZGPU_FAST_LANES): a wave's step takes as long as its slowest lane's memory access
        if (forced < 0) { const char *v = getenv("ZGPU_FAST_LANES"); forced = v ? atoi(v) : 0; if (forced < 0 || forced > 64) forced = 0; }
        uint32_t lanes = (uint32_t)forced;
        if (!lanes) { lanes = 1; while (lanes < 64 && (uint64_t)lanes * 4096 < g.nchunks) lanes <<= 1; } // (measured best at 4 GiB: 16 chunks per wave)
        hipLaunchKernelGGL(fast_kernel, dim3((g.nchunks + lanes - 1) / lanes), dim3(64), 0, st, g, cfg, S, ir, G, tokens, meta, lanes);
        prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
        return adler_done;
    }
    if (walk) { // the records' memory holds the walkers' output: gm (u32 per position), then the bitmaps gs (2048 words per chunk)
        uint32_t *gm = reinterpret_cast<uint32_t *>(recs), *gs = gm + nch * kChunkMax;
        static int fuse = -1; // ZGPU_WALK_FUSE=0: the rest of the parse as a kernel of its own (A/B runs)
        if (fuse < 0) { const char *v = getenv("ZGPU_WALK_FUSE"); fuse = v ? atoi(v) : 1; }
        static bool opt_inw = false;
        if (!opt_inw) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(walk_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWLds);
            hipFuncSetAttribute(reinterpret_cast<const void *>(walk_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWLds);
            opt_inw = true;
        }
        if (fuse) {
            hipLaunchKernelGGL(walk_kernel<1>, dim3(g.nchunks), dim3(kWThreads), kWLds, st, g, cfg, S, ir, gm, gs, tokens, meta, TileGeom{});
            prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
            return adler_done;
        }
        hipLaunchKernelGGL(walk_kernel<0>, dim3(g.nchunks), dim3(kWThreads), kWLds, st, g, cfg, S, ir, gm, gs, tokens, meta, TileGeom{});
        prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
        prof_span_begin(prof, st, &ev);
        launch_parse_lite(g, cfg, gm, gs, tokens, meta, st);
        prof_span_end(prof, st, ZGPU_STAGE_PARSE, ev);
        return adler_done;
    }
    static bool opt_in3 = false;
    if (!opt_in3) { hipFuncSetAttribute(reinterpret_cast<const void *>(match3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kM3Lds); opt_in3 = true; }
    hipLaunchKernelGGL(match3_kernel, dim3(g.nchunks), dim3(kM2Threads), kM3Lds, st, g, cfg, S, heads, recs);
    prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
    prof_span_begin(prof, st, &ev);
    launch_parse(g, cfg, recs, tokens, meta, st);

    prof_span_end(prof, st, ZGPU_STAGE_PARSE, ev);
    return adler_done;
}

// A batch of tiles of a continuous stream (zgpu_cont.hip): sort, walkers + exit functions, the chain of entries, the tiles' tokens.
void launch_chain(const uint16_t *exits, uint32_t ntiles, uint16_t *comp, uint16_t *gentry, uint16_t *entry, hipStream_t st);
void launch_parse_tile(const ChunkGeom &g, LevelCfg cfg, const uint32_t *gm, const uint32_t *gs, uint32_t *tokens, ChunkMeta *meta, const TileGeom &tg, hipStream_t st);
// the sort alone (levels 1-3 go on with fastwin_tile_kernel's rounds, zgpu_engine.hip): where S and ir of the batch's tiles are
void launch_sort_tiles(const ChunkGeom &g, void *workspace, ChunkMeta *meta, hipStream_t st, void *prof, int exact_sort, const uint16_t **S_out, const uint32_t **ir_out)
{
    uint8_t *w = static_cast<uint8_t *>(workspace);
    const size_t nch = g.nchunks;
    uint32_t *fault = reinterpret_cast<uint32_t *>(w);
    uint16_t *S = reinterpret_cast<uint16_t *>(w + 256);
    uint16_t *rk = reinterpret_cast<uint16_t *>(w + 256 + ((nch * kSStride * 2 + 255) & ~(size_t)255));
    uint32_t *heads = reinterpret_cast<uint32_t *>(rk + nch * kChunkMax);
    uint2 *recs = reinterpret_cast<uint2 *>(heads + nch * kHeadStride);
    uint32_t *ir = reinterpret_cast<uint32_t *>(recs + nch * kChunkMax);
    hipEvent_t ev{};
    prof_span_begin(prof, st, &ev);
    static int sort_env = -1;
    if (sort_env < 0) { const char *e = getenv("ZGPU_SORT"); sort_env = e ? atoi(e) : 4; } // 4: sort4_kernel, 3: sort3_kernel, 1: the ballot-ranked sort
    if (exact_sort || sort_env == 1) {
        hipLaunchKernelGGL(sort_kernel, dim3(g.nchunks), dim3(kSortThreads), 0, st, g, S, rk, heads, ir);
        hipLaunchKernelGGL(heads_below_kernel, dim3(g.nchunks), dim3(1024), 0, st, heads);
    } else {
        if (sort_env == 3) hipLaunchKernelGGL(sort3_kernel, dim3(g.nchunks), dim3(kS3Threads), 0, st, g, S, rk, heads, fault, ir, meta);
        else hipLaunchKernelGGL(sort4_kernel, dim3(g.nchunks), dim3(kS3Threads), 0, st, g, S, heads, fault, ir, meta);
        if (g_inject_sort_fault.exchange(0)) hipMemsetAsync(fault, 1, 4, st);
    }
    prof_span_end(prof, st, ZGPU_STAGE_CHAIN, ev);
    *S_out = S; *ir_out = ir;
}
void launch_lz_tiles(const ChunkGeom &g, const TileGeom &tg, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, uint16_t *comp, uint16_t *gentry, hipStream_t st,
                     void *prof, int exact_sort)
{
    uint8_t *w = static_cast<uint8_t *>(workspace);
    const size_t nch = g.nchunks;
    uint32_t *fault = reinterpret_cast<uint32_t *>(w);
    uint16_t *S = reinterpret_cast<uint16_t *>(w + 256);
    uint16_t *rk = reinterpret_cast<uint16_t *>(w + 256 + ((nch * kSStride * 2 + 255) & ~(size_t)255));
    uint32_t *heads = reinterpret_cast<uint32_t *>(rk + nch * kChunkMax);
    uint2 *recs = reinterpret_cast<uint2 *>(heads + nch * kHeadStride);
    uint32_t *ir = reinterpret_cast<uint32_t *>(recs + nch * kChunkMax);
    uint32_t *gm = reinterpret_cast<uint32_t *>(recs), *gs = gm + nch * kChunkMax;
    hipEvent_t ev{};
    prof_span_begin(prof, st, &ev);
    static int sort_env = -1;
    if (sort_env < 0) { const char *e = getenv("ZGPU_SORT"); sort_env = e ? atoi(e) : 4; } // 4: sort4_kernel, 3: sort3_kernel, 1: the ballot-ranked sort
    if (exact_sort || sort_env == 1) {
        hipLaunchKernelGGL(sort_kernel, dim3(g.nchunks), dim3(kSortThreads), 0, st, g, S, rk, heads, ir);
        hipLaunchKernelGGL(heads_below_kernel, dim3(g.nchunks), dim3(1024), 0, st, heads);
    } else {
        if (sort_env == 3) hipLaunchKernelGGL(sort3_kernel, dim3(g.nchunks), dim3(kS3Threads), 0, st, g, S, rk, heads, fault, ir, meta);
        else hipLaunchKernelGGL(sort4_kernel, dim3(g.nchunks), dim3(kS3Threads), 0, st, g, S, heads, fault, ir, meta);
        if (g_inject_sort_fault.exchange(0)) hipMemsetAsync(fault, 1, 4, st);
    }
    prof_span_end(prof, st, ZGPU_STAGE_CHAIN, ev);
    prof_span_begin(prof, st, &ev);
    static bool opt_in = false;
    if (!opt_in) { hipFuncSetAttribute(reinterpret_cast<const void *>(walk_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWLds); opt_in = true; }
    hipLaunchKernelGGL(walk_kernel<2>, dim3(g.nchunks), dim3(kWThreads), kWLds, st, g, cfg, S, ir, gm, gs, tokens, meta, tg);
    prof_span_end(prof, st, ZGPU_STAGE_MATCH, ev);
    prof_span_begin(prof, st, &ev);
    launch_chain(tg.exits, g.nchunks, comp, gentry, tg.entry + g.chunk0, st);
    prof_span_end(prof, st, ZGPU_STAGE_PARSE, ev);
    (void)tokens;
}
// ... and the tiles' tokens from their true entries: a launch of its own, so that it can run on another stream under the next batch's walkers (it is all
// latency -- a window at a time, one wave threading the path -- and they are bound by the vector units: what the chunk path gets by fusing the two)
void launch_lz_tiles_parse(const ChunkGeom &g, const TileGeom &tg, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, void *prof)
{
    uint8_t *w = static_cast<uint8_t *>(workspace);
    const size_t nch = g.nchunks;
    uint16_t *rk = reinterpret_cast<uint16_t *>(w + 256 + ((nch * kSStride * 2 + 255) & ~(size_t)255));
    uint32_t *heads = reinterpret_cast<uint32_t *>(rk + nch * kChunkMax);
    uint2 *recs = reinterpret_cast<uint2 *>(heads + nch * kHeadStride);
    uint32_t *gm = reinterpret_cast<uint32_t *>(recs), *gs = gm + nch * kChunkMax;
    hipEvent_t ev{};
    prof_span_begin(prof, st, &ev);
    launch_parse_tile(g, cfg, gm, gs, tokens, meta, tg, st);
    prof_span_end(prof, st, ZGPU_STAGE_PARSE, ev);
}

// the word sort3's pass V raises (first word of the workspace)
uint32_t *lz_sorted_fault_word(void *workspace) { return static_cast<uint32_t *>(workspace); }

} // namespace zgpu
