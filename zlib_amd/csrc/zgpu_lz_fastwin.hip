// zgpu_lz_fastwin.hip -- LZ77 stage for levels 1-3: deflate_fast (/root/reference/qcsrc/deflate.c:1448-1546, longest_match :1027-1168),
// one WAVE per chunk, 64 positions per step, the chunk's window and its hash chains in LDS.
//
// The chains of deflate_fast depend on the parse: the strings inside a match longer than max_insert_length never enter them
// (deflate.c:1510-1534).  So neither head[]/prev[] filled ahead of the parse nor the static chains of levels 4-9 apply.  What IS known ahead of
// the parse is which positions share a hash: sort3_kernel has counting-sorted the positions by hash (S; idx(p), rank(p) in `ir`), the bucket
// predecessors of p, nearest first, are S[idx-1], S[idx-2], ... S[idx-rank], and the chain of p is those of them that have been inserted.
// "Inserted" is ONE BIT PER S INDEX, kept in LDS (8 KiB): the bits of p's 32 nearest predecessors are 32 consecutive bits -- one read, and
// the first `max_chain_length` set ones are the chain.  The bytes the candidates are compared with come from a ring of the last 33 KiB of the
// chunk in LDS (MAX_DIST + lookahead); the S entries and ir words of a window are fetched from HBM one and two windows ahead of the parse (they
// do not depend on it), so no trip to memory lies on the chain of dependent steps.  47 KiB of LDS per chunk: three chunks per CU.
//
// A window: every lane evaluates longest_match at its own position under the bits as they stand -- final below the window, a guess ("set")
// inside it; a scalar walk follows the token starts (a literal run is one step); a match longer than max_insert_length clears the bits of its
// inside; then every lane reads its bits again: a token start whose search had examined a bit that is now clear is stale, the tokens in front of
// the first stale one stand, everything from there on is evaluated again (1.5 evaluations per window on the Silesia-mix,
// tests/tools/fastwin_model.c, which is this algorithm on the CPU, checked against the plain loop).  A search that needs more than 32
// predecessors (1.5 % at level 1) is done over the whole bucket by all lanes when the walk stands on it.
//
// Measured (MI355X, scripts/fw_time.py): a lone wave per SIMD issues an instruction every five to six cycles, a window costs 12 000 cycles (evaluation
// 30 %, the scalar walk 45 %), a chunk 5 ms whatever the size of the call -- against 28 ms for the lane-per-chunk loop, which needs tens of thousands
// of chunks in flight and wins only above 2.5 GiB a call (the engine picks by the size of the call).  Tried and dropped (git history, commit
// "the waves of a workgroup share one chunk"; tests/tools/fastwin_spec_model.c is its protocol on the CPU): six waves per chunk taking the
// windows in turn, each evaluating and walking its window ahead of the parse and only checking at its turn.  Exact, and slower: a wave two to
// five windows ahead sees bits that are not there yet, 0.7-0.8 evaluations per window stay on the serial path.
#include "zgpu_common.h"
#include <cstdlib>

#ifdef ZGPU_FW_TIME // timing builds only (scripts/build_variant.sh fwtime -DZGPU_FW_TIME; scripts/fw_time.py): cycles by phase, counts
__device__ unsigned long long g_fw_time[16];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_fw_time(unsigned long long *out, int reset)
{
    unsigned long long z[16] = {0};
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fw_time), z, sizeof z); else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fw_time), sizeof z);
}
#define FW_T0() unsigned long long fw_t_ = __builtin_amdgcn_s_memtime(); unsigned long long fw_acc_[16] = {0}
#define FW_LAP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); fw_acc_[k] += n_ - fw_t_; fw_t_ = n_; } while (0)
#define FW_CNT(k, v) do { fw_acc_[k] += (v); } while (0)
#define FW_END() do { if (lane == 0) for (int k_ = 0; k_ < 16; k_++) atomicAdd(&g_fw_time[k_], fw_acc_[k_]); } while (0)
#else
#define FW_T0() do { } while (0)
#define FW_LAP(k) do { } while (0)
#define FW_CNT(k, v) do { } while (0)
#define FW_END() do { } while (0)
#endif

namespace zgpu {

constexpr uint32_t kSPadF = 8, kSStrideF = kChunkMax + kSPadF; // S layout of zgpu_lz_sorted.hip (kSPad entries in front of every chunk's S)
constexpr uint32_t kFwRing = 34 * 1024;             // bytes of the chunk in LDS: MAX_DIST back, a window + MAX_MATCH ahead, filled 1 KiB at a time
constexpr uint32_t kFwMirror = 48;                  // the ring's first bytes again behind its end: a 40-byte read (nice_match 32 + 8) may start at its last byte
constexpr uint32_t kFwFlagWords = kChunkMax / 32 + 2;
constexpr uint32_t kFwStgStride = 80;               // bytes per lane of the staged S entries (64 + 16: 16-byte stores free of bank conflicts)
constexpr uint32_t kFwOffFlags = kFwRing + kFwMirror;
constexpr uint32_t kFwOffStg = (kFwOffFlags + kFwFlagWords * 4 + 15) & ~15u;
constexpr uint32_t kFwLds = kFwOffStg + 64 * kFwStgStride;
constexpr uint32_t kFwDepth = 32;                   // predecessors a lane looks at
constexpr uint32_t kResLen = 511, kResTerm = 1u << 9, kResInc = 1u << 10;

struct __attribute__((packed, aligned(1))) FwU128 { uint4 v; };
struct __attribute__((packed, aligned(1))) FwU32 { uint32_t v; };

typedef __attribute__((address_space(3))) uint8_t lds_u8;
__device__ inline uint32_t fw_lds_base(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }
// 8 / 4 bytes at any LDS byte offset (the hardware serves unaligned ds reads; their 36 clocks in the LDS cost one issue slot)
__device__ inline uint64_t fw_ld64(uint32_t a) { uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory"); return v; }
__device__ inline uint32_t fw_ld32(uint32_t a) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory"); return v; }
__device__ inline void fw_ld32x2(uint32_t a, uint32_t b, uint32_t &x, uint32_t &y) { asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(a), "v"(b) : "memory"); }
__device__ inline void fw_ld64x2(uint32_t a, uint32_t b, uint64_t &x, uint64_t &y)
{
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(a), "v"(b) : "memory");
}
__device__ inline void fw_ld64x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint64_t &x, uint64_t &y, uint64_t &z, uint64_t &w)
{
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w) : "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
// RING false (round 4): no copy of the chunk in LDS -- the bytes of a position and of its candidates come from the input itself (L2 / the Infinity Cache: a chunk's 64 KiB have just
// been streamed by the sort), which leaves 13 KiB of LDS per chunk (the bits and the staged S entries) and lets eleven chunks share a CU where the 34 KiB ring allowed three
struct __attribute__((packed, aligned(1))) FwU64 { uint64_t v; };
#ifndef ZGPU_FW_STG64
#define ZGPU_FW_STG64 0 // 1 (built at the end of round 4, scripts/build_variant.sh stg64 -DZGPU_FW_STG64=1): the staged entries of the form without the ring at 64 bytes a lane, slots swizzled -- 12 304 bytes a
                        // workgroup, TWELVE to a CU (LDS is handed out in pieces of 1 280 bytes): 4 GiB level 1 in 22 phases 673 -> 564 ms, parity tests of levels 1-3 pass; not the default before the whole suite has run on it
#endif
constexpr uint32_t kFwOffFlagsNR = 0, kFwOffStgNR = (kFwFlagWords * 4 + 15) & ~15u, kFwLdsNR = kFwOffStgNR + (ZGPU_FW_STG64 ? 64 * 64 : 63 * kFwStgStride + 64); // (13 312 bytes, without the last lane's padding.  ELEVEN workgroups run on a CU, not the twelve that 160 KiB / 13 KiB promise: a launch of 3 003 tiles takes two waves of workgroups' time, one of 2 752 one -- zgpu_engine.hip lz_tiles_fast)
__device__ inline uint64_t fw_g64(const uint8_t *src, uint32_t pos, uint64_t safe_end)
{
    if ((uint64_t)pos + 8 <= safe_end) return reinterpret_cast<const FwU64 *>(src + pos)->v;
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8 && (uint64_t)pos + k < safe_end; k++) v |= (uint64_t)src[pos + k] << (8 * k);
    return v;
}
__device__ inline uint32_t fw_g32(const uint8_t *src, uint32_t pos, uint64_t safe_end)
{
    if ((uint64_t)pos + 4 <= safe_end) return reinterpret_cast<const FwU32 *>(src + pos)->v;
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4 && (uint64_t)pos + k < safe_end; k++) v |= (uint32_t)src[pos + k] << (8 * k);
    return v;
}
__device__ inline uint32_t fw_ring(uint32_t p) { const uint32_t d = p - kFwRing; return p < d ? p : d; } // p mod ring size, p < 2 * ring
// Byte offset of 16-byte slot j (0 .. 3) of a lane's 64 bytes of staged S entries.  At 80 bytes a lane the 16-byte stores of neighbouring lanes fall into different banks; at 64
// (ZGPU_FW_STG64, the form without the ring only) the slot number is swizzled by the lane's instead.
template <bool RING> __device__ inline uint32_t fw_stg_slot(uint32_t lane, uint32_t j)
{
    if (RING || !ZGPU_FW_STG64) return lane * kFwStgStride + 16u * j;
    return lane * 64u + 16u * (j ^ ((lane >> 1) & 3u));
}
__device__ inline uint32_t fw_diff8(uint64_t x) { return x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u; }

__device__ __noinline__ uint4 fw_tail16(const uint8_t *in, uint32_t o, uint64_t safe_end)
{
    uint32_t v[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16 && (uint64_t)o + k < safe_end; k++) v[k >> 2] |= (uint32_t)in[o + k] << (8 * (k & 3));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// CHAIN: max_chain_length; NICE: nice_match (a multiple of 8: the lanes compare that many bytes of every candidate, a match that reaches it is
// measured by all lanes together when the walk takes it)
template <int CHAIN, int NICE, bool RING>
__global__ void __launch_bounds__(64) fastwin_kernel(ChunkGeom g, uint32_t max_insert, const uint16_t *__restrict__ S_all, const uint32_t *__restrict__ ir_all,
                                                      uint32_t *__restrict__ tokens, ChunkMeta *__restrict__ meta)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fw_lds[];
    static_assert(NICE % 8 == 0 && NICE >= 8 && NICE <= 32 && CHAIN % 4 == 0, "whole groups of four candidates, whole 8-byte steps");
    constexpr int NW = NICE / 8;
    constexpr uint32_t OFF_FLAGS = RING ? kFwOffFlags : kFwOffFlagsNR, OFF_STG = RING ? kFwOffStg : kFwOffStgNR;
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    const uint64_t safe_end = g.in_bytes - lo;
    const uint16_t *S = S_all + (size_t)c * kSStrideF + kSPadF;
    const uint32_t *ir = ir_all + (size_t)c * kChunkMax;
    const uint32_t cm = chunk_of(g, c); // (a launch over a list of chunks: input, tokens and meta are the listed chunk's, the sorted buckets are slot c's)
    uint32_t *tok = tokens + (size_t)cm * kChunkMax;
    const uint32_t base = chunk_base(g, c), npos = n >= 3 ? n - 2 : 0;
    uint32_t *flags = reinterpret_cast<uint32_t *>(fw_lds + OFF_FLAGS);
    const uint32_t ring_a = fw_lds_base(fw_lds);
    const uint64_t lane_bit = 1ull << lane, lanes_below = lane_bit - 1;

    for (uint32_t i = lane; i < kFwFlagWords; i += 64) flags[i] = 0;
    FW_T0();

    // ---- the ring: block k (1 KiB of the chunk) lives at (k mod 34) KiB ----
    uint32_t filled = 0;
    auto fill_to = [&](uint32_t need) { // (uniform)
        while (RING && filled < need) {
            const uint32_t o = filled + 16 * lane;
            uint4 v;
            if ((uint64_t)o + 16 <= safe_end) v = reinterpret_cast<const FwU128 *>(src + o)->v; else v = fw_tail16(src, o, safe_end);
            const uint32_t ro = fw_ring(filled) + 16 * lane;
            *reinterpret_cast<uint4 *>(fw_lds + ro) = v;
            if (ro < kFwMirror) *reinterpret_cast<uint4 *>(fw_lds + kFwRing + ro) = v;
            filled += 1024;
        }
    };
    fill_to(1024);

    // ---- block bookkeeping (cut_block of zgpu_lz_serial.hip; deflate.c:1365-1367 for `nostore`) ----
    uint32_t off = 0, ntok = 0, blk_tok0 = 0, nblk = 0, nostore = 0, block_start = 0;
    const uint32_t room = 2 * kWSize - base;
    uint32_t buffered = n < room ? n : room; // first fill_window (deflate.c:1275,1342)
    auto token_top = [&](uint32_t p) { // the loop top of deflate_fast for the token that starts at p: fill_window and its slide (deflate.c:1293)
        if (buffered - p < kMinLookahead) { if ((int)(p + base) - (int)off >= (int)(kWSize + kMaxDist)) off += kWSize; buffered = n; }
    };
    auto cut_block = [&](uint32_t p_end) {
        if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk;
        nblk++; blk_tok0 += kBlockTokens; block_start = p_end;
    };

    // ---- what is fetched ahead: ir of the window after next, the S entries of the next window ----
    auto load_ir = [&](uint32_t w) -> uint32_t { const uint32_t p = w * 64 + lane; return p < npos ? ir[p] : 0u; };
    uint4 sq0, sq1, sq2, sq3; // S[idx-32 .. idx-1] of this lane's position: 64 bytes, the nearest predecessor last
    auto load_s = [&](uint32_t iv) {
        const uint32_t rk = iv >> 16;
        const uint8_t *a = reinterpret_cast<const uint8_t *>(S) + 2 * (int)(iv & 0xffffu) - 64; // (below S[0]: the pad and the chunk in front, or the header -- masked by the rank)
        sq0 = sq1 = sq2 = sq3 = make_uint4(0, 0, 0, 0);
        if (rk > 0) sq3 = reinterpret_cast<const FwU128 *>(a + 48)->v;
        if (rk > 8) sq2 = reinterpret_cast<const FwU128 *>(a + 32)->v;
        if (rk > 16) sq1 = reinterpret_cast<const FwU128 *>(a + 16)->v;
        if (rk > 24) sq0 = reinterpret_cast<const FwU128 *>(a)->v;
    };
    uint32_t ir_cur = load_ir(0), ir_nxt = load_ir(1);
    load_s(ir_cur);

    uint32_t pos = 0;        // where the next token starts
    bool cross_short = false; // the match that reaches into this window was a short one: its strings are in the chains
    const uint32_t nwin = (n + 63) / 64;
    __syncthreads();

    for (uint32_t win = 0; win < nwin; win++) {
        const uint32_t w0 = win * 64, p = w0 + lane;
        const bool skip = pos >= w0 + 64; // the window lies inside a match (a long one: a short one ends within six positions)
        if (!skip) { // the staged entries of this window
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 0)) = sq0;
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 1)) = sq1;
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 2)) = sq2;
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 3)) = sq3;
        }
        const uint32_t iv = ir_cur;
        load_s(ir_nxt);
        ir_cur = ir_nxt;
        ir_nxt = load_ir(win + 2);
        fill_to(w0 + 64 + kMaxMatch + NICE + 16 < kChunkMax + 1024 ? w0 + 64 + kMaxMatch + NICE + 16 : kChunkMax + 1024);
        FW_LAP(0);
        if (skip) continue;
        FW_CNT(8, 1);

        const uint32_t idx = iv & 0xffffu, rk = iv >> 16;
        const bool haspos = p < npos;
        const uint32_t B0 = 65536u - idx;                       // bit B0 + k of the (reversed) bitmap belongs to predecessor k
        const uint32_t own_w = (B0 - 1) >> 5, own_b = 1u << ((B0 - 1) & 31u);
        const uint32_t entry = pos - w0;
        if (haspos && (lane >= entry || cross_short)) atomicOr(&flags[own_w], own_b);
        const uint32_t look = n > p ? n - p : 0, cap = look < kMaxMatch ? look : kMaxMatch, ni = (uint32_t)NICE < look ? (uint32_t)NICE : look;
        const uint32_t cmp_max = cap < (uint32_t)NICE ? cap : (uint32_t)NICE;
        const int w = (int)(p + base), limit = w > (int)kMaxDist ? w - (int)kMaxDist : 0;
        uint64_t own[NW];
        if (!RING) {
#pragma unroll
            for (int t = 0; t < NW; t++) own[t] = fw_g64(src, p + 8 * t, safe_end);
        } else {
            const uint32_t own_a = ring_a + fw_ring(p);
            if (NW == 1) own[0] = fw_ld64(own_a);
            else { fw_ld64x2(own_a, own_a + 8, own[0], own[1]); if (NW == 4) fw_ld64x2(own_a + 16, own_a + 24, own[2], own[3]); }
        }
        const uint32_t own_byte = (uint32_t)own[0] & 255u;

        uint32_t res = 1, mstart = 0, mex = 0; // this lane's search: length | kResTerm | kResInc; where the match starts; the bits it examined
        const uint32_t lend = n - w0 < 64 ? n - w0 : 64; // lanes of this window that are positions of the chunk
        auto read_bits = [&]() -> uint32_t {
            const uint32_t wa = B0 >> 5, lo32 = flags[wa], hi32 = flags[wa + 1];
            uint32_t m = __builtin_amdgcn_alignbit(hi32, lo32, B0 & 31u);
            if (rk < 32) m &= (1u << rk) - 1u;
            return m;
        };

        uint32_t start = entry;
        bool need_eval = true, dirty = false; // dirty: bits have been cleared since the lanes from `start` on were evaluated
        for (;;) { // rounds
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (need_eval) dirty = false;
            FW_LAP(1);
            FW_CNT(9, 1);
            if (need_eval) FW_CNT(10, 1);
            if (need_eval && lane >= start) {
                res = 1; mstart = 0; mex = 0;
                if (haspos && rk != 0) {
                    uint32_t m = read_bits(), best = kMinMatch - 1, nsel = 0;
                    bool first = true, stopped = false, term = false;
#pragma unroll 1
                    for (int g0 = 0; g0 < CHAIN; g0 += 4) {
                        if (__builtin_amdgcn_ballot_w64(!stopped && m != 0) == 0) break;
                        uint32_t k[4], q[4]; bool v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { v[u] = m != 0; k[u] = v[u] ? (uint32_t)__builtin_ctz(m) : 0u; m &= m - 1u; }
#pragma unroll
                        for (int u = 0; u < 4; u++) { const uint32_t o = 62u - 2u * k[u]; q[u] = *reinterpret_cast<const uint16_t *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, o >> 4) + (o & 15u)); }
                        uint64_t cb[4][NW];
                        if (!RING) {
#pragma unroll
                            for (int t = 0; t < NW; t++)
#pragma unroll
                                for (int u = 0; u < 4; u++) cb[u][t] = fw_g64(src, q[u] + 8 * t, safe_end);
                        } else {
                            const uint32_t a0 = ring_a + fw_ring(q[0]), a1 = ring_a + fw_ring(q[1]), a2 = ring_a + fw_ring(q[2]), a3 = ring_a + fw_ring(q[3]);
#pragma unroll
                            for (int t = 0; t < NW; t++) fw_ld64x4(a0 + 8 * t, a1 + 8 * t, a2 + 8 * t, a3 + 8 * t, cb[0][t], cb[1][t], cb[2][t], cb[3][t]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (v[u] && !stopped) { // longest_match's bookkeeping for this candidate (deflate.c:1126-1163)
                                mex |= 1u << k[u];
                                const int wq = (int)(q[u] + base);
                                const bool far = first ? (wq <= 0 || (uint32_t)(w - wq) > kMaxDist) : wq <= limit; // hash_head out of reach: no search (deflate.c:1481); the chain ends (:1163)
                                if (far) stopped = true;
                                else {
                                    first = false; nsel++;
                                    uint32_t l = 0;
#pragma unroll
                                    for (int t = 0; t < NW; t++) if (l == 8u * t) l += fw_diff8(cb[u][t] ^ own[t]);
                                    l = l < cmp_max ? l : cmp_max;
                                    if (l > best) { best = l; mstart = q[u]; if (l >= ni) { stopped = true; term = l < cap; } }
                                    if (nsel == (uint32_t)CHAIN) stopped = true;
                                }
                            }
                        }
                    }
                    if (!first && best >= kMinMatch) res = best | (term ? kResTerm : 0u);
                    if (!stopped && nsel < (uint32_t)CHAIN && rk > kFwDepth) res = kResInc; // the chain may go on below the bits this lane has
                }
            }
            FW_LAP(2);
            // ---- the walk over the token starts, from `start` ----
            // What the scalar loop needs of a lane comes packed: where the token behind this lane's starts (lane + length, unclamped) and the two flags.
            // Whether a match's strings stay out of the chains is a mask of its own (a measured match is NICE or longer, i.e. longer than
            // max_insert_length at the three levels this kernel serves), and the positions inside such matches (C) are found by all lanes behind the loop.
            const uint64_t nonlit = __builtin_amdgcn_ballot_w64(res != 1u);
            uint32_t pk = (lane + (res & kResLen)) | (res & (kResTerm | kResInc));
            const uint32_t len_e = res & kResLen;
            const uint64_t longm = __builtin_amdgcn_ballot_w64(len_e >= kMinMatch && !(len_e <= max_insert && n - p - len_e >= kMinMatch));
            uint64_t T = 0;
            uint32_t L = start, stop_inc = 64, Llast = 64;
            while (L < lend) {
                const uint64_t ahead = nonlit >> L;
                if (ahead == 0) { T |= (~0ull << L) & (lend >= 64 ? ~0ull : (1ull << lend) - 1ull); L = lend; Llast = 64; break; } // literals to the end of the window
                const uint32_t run = (uint32_t)__builtin_ctzll(ahead); // literals up to the next lane with a match (a lane below lend: others hold 1)
                T |= ((1ull << run) - 1ull) << L;
                L += run;
                uint32_t x = __builtin_amdgcn_readlane(pk, L);
                if (x & (kResInc | kResTerm)) {
                    if (x & kResInc) { stop_inc = L; break; }
                    FW_CNT(11, 1);
                    // all lanes measure the match: four bytes each behind the NICE the lane has compared
                    const uint32_t p0 = w0 + L, q0 = __builtin_amdgcn_readlane(mstart, L), cap0 = n - p0 < kMaxMatch ? n - p0 : kMaxMatch;
                    const uint32_t o = NICE + 4 * lane;
                    uint32_t xa, xb;
                    if (!RING) { xa = fw_g32(src, q0 + o, safe_end); xb = fw_g32(src, p0 + o, safe_end); }
                    else fw_ld32x2(ring_a + fw_ring(q0 + o), ring_a + fw_ring(p0 + o), xa, xb);
                    xa ^= xb;
                    const uint64_t ne = __builtin_amdgcn_ballot_w64(xa != 0);
                    uint32_t len = cap0;
                    if (ne != 0) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(ne), xf = __builtin_amdgcn_readlane(xa, f);
                        len = NICE + 4 * f + ((uint32_t)__builtin_ctz(xf) >> 3);
                        len = len < cap0 ? len : cap0;
                    }
                    res = lane == L ? len : res; // (the length is known now, whatever becomes of this round)
                    pk = lane == L ? L + len : pk;
                    x = L + len;
                }
                T |= 1ull << L;
                Llast = L;
                L = x & kResLen;
            }
            bool cs_new = cross_short;
            if (L >= 64 && Llast < 64) cs_new = !((longm >> Llast) & 1ull); // the match that reaches into the next window
            uint64_t C = 0;
            {
                const uint64_t below = T & lanes_below; // the token starts in front of this lane
                const uint32_t owner = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                const bool inside = below != 0 && !(T & lane_bit) && lane < L && ((longm >> owner) & 1ull);
                C = __builtin_amdgcn_ballot_w64(inside);
            }
            FW_LAP(3);
            // ---- which of these tokens stand ----
            if ((C & lane_bit) && haspos) atomicAnd(&flags[own_w], ~own_b);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            dirty = dirty || C != 0;
            bool stale = false;
            if (dirty && lane >= start && mex != 0) stale = (mex & ~read_bits()) != 0;
            const uint64_t st_mask = __builtin_amdgcn_ballot_w64(stale) & T;
            const uint32_t Lstale = st_mask ? (uint32_t)__builtin_ctzll(st_mask) : 64u;
            const uint32_t Lacc = Lstale < stop_inc ? Lstale : stop_inc; // the tokens that start below Lacc are the reference's
            const uint64_t Tacc = Lacc >= 64 ? T : T & ((1ull << Lacc) - 1ull);
            const uint32_t end_pos = Lacc >= 64 ? w0 + L : w0 + Lacc; // where the token behind the accepted ones starts
            FW_LAP(4);
            // the accepted tokens
            {
                const uint32_t cnt = (uint32_t)__builtin_popcountll(Tacc);
                if (Tacc & lane_bit) {
                    const uint32_t ln = res & kResLen;
                    tok[ntok + (uint32_t)__builtin_popcountll(Tacc & lanes_below)] = ln == 1 ? tok_lit(own_byte) : tok_match(p - mstart, ln - kMinMatch);
                }
                const bool near_end = w0 + 64 + kMinLookahead > (buffered < n ? buffered : n);
                if (near_end || ntok + cnt - blk_tok0 >= kBlockTokens) { // the rare windows in which the slide or a block cut happens: token by token
                    uint64_t t = Tacc; uint32_t k = 0;
                    while (t) {
                        const uint32_t l0 = (uint32_t)__builtin_ctzll(t); t &= t - 1;
                        token_top(w0 + l0);
                        k++;
                        if (ntok + k - blk_tok0 == kBlockTokens) cut_block(t ? w0 + (uint32_t)__builtin_ctzll(t) : end_pos);
                    }
                }
                ntok += cnt;
            }
            FW_LAP(5);
            if (Lacc >= 64 || Lacc >= lend) { pos = w0 + L; cross_short = cs_new; break; }
            // the bits behind the accepted tokens are guesses again
            if (lane >= Lacc && (C & lane_bit) && haspos) atomicOr(&flags[own_w], own_b);
            start = Lacc;
            if (Lstale <= stop_inc) { need_eval = true; continue; }
            FW_CNT(12, 1);
            // ---- the search at Lacc over its whole bucket, by all lanes (every bit below it is final now) ----
            {
                const uint32_t p0 = w0 + Lacc, i0 = __builtin_amdgcn_readlane(idx, Lacc), r0 = __builtin_amdgcn_readlane(rk, Lacc);
                const uint32_t look0 = n - p0, cap0 = look0 < kMaxMatch ? look0 : kMaxMatch, ni0 = (uint32_t)NICE < look0 ? (uint32_t)NICE : look0;
                const int w00 = (int)(p0 + base), limit0 = w00 > (int)kMaxDist ? w00 - (int)kMaxDist : 0;
                uint32_t best = kMinMatch - 1, ms0 = 0, ch = CHAIN;
                bool first = true, done = false;
                const uint32_t pa = ring_a;
                for (uint32_t k0 = 0; k0 < r0 && !done; k0 += 64) {
                    const uint32_t kk = k0 + lane;
                    const bool valid = kk < r0;
                    const uint32_t q = valid ? S[(int)i0 - 1 - (int)kk] : 0u;
                    const uint32_t rv = 65536u - i0 + kk;
                    const bool ins = valid && ((flags[rv >> 5] >> (rv & 31u)) & 1u);
                    uint32_t l = 0;
                    if (ins && (int)(q + base) > limit0 - 1) { // (a candidate out of reach is turned away before its length is asked for; its bytes may have left the ring)
                        for (;;) {
                            const uint32_t d = RING ? fw_diff8(fw_ld64(pa + fw_ring(q + l)) ^ fw_ld64(pa + fw_ring(p0 + l))) : fw_diff8(fw_g64(src, q + l, safe_end) ^ fw_g64(src, p0 + l, safe_end));
                            l += d;
                            if (d < 8 || l >= cap0) break;
                        }
                        l = l < cap0 ? l : cap0;
                    }
                    uint64_t im = __builtin_amdgcn_ballot_w64(ins);
                    while (im) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(im); im &= im - 1;
                        const uint32_t qf = __builtin_amdgcn_readlane(q, f), lf = __builtin_amdgcn_readlane(l, f);
                        const int wq = (int)(qf + base);
                        if (first ? (wq <= 0 || (uint32_t)(w00 - wq) > kMaxDist) : wq <= limit0) { done = true; break; }
                        first = false;
                        if (lf > best) { best = lf; ms0 = qf; if (lf >= ni0) { done = true; break; } }
                        if (--ch == 0) { done = true; break; }
                    }
                }
                const uint32_t r1 = (!first && best >= kMinMatch) ? best : 1u;
                if (lane == Lacc) { res = r1; mstart = ms0; mex = 0; }
            }
            need_eval = false;
            FW_LAP(6);
        }
    }
    FW_END();
    token_top(n); // the loop top that finds the input at its end (deflate.c:1459-1466): the slide may still happen here
    if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk; // the final block (its emission happens in the Huffman stage)
    if (lane == 0) { meta[cm].ntok = ntok; meta[cm].nostore = nostore; meta[cm].in_bytes = n; }
}

// ======================================================================================================================================
// The same parse for a TILE of a continuous stream (zgpu_cont.hip; round 4).  deflate_fast's state at a token start is the position AND which of the
// 32506 positions in front of it are in the hash chains -- the strings inside a long match are not -- so a tile cannot be parsed before the tile in
// front of it has been.  It can be GUESSED, though: in round 0 every tile but the batch's first parses its own history as well, from nothing (a warm-up:
// 32512 positions later the parse and the chains have almost always fallen into step with the stream's); from round 1 on a tile whose predecessor's
// results have changed is parsed again from where the predecessor's parse ended, with the predecessor's "inserted" bits as its history; a round in which
// no tile's results (exit, bits) change has reached the one consistent assignment, which is the reference's parse (the recursion tile i = F(tile i - 1)
// has exactly one solution).  The engine runs rounds until no tile is active (zgpu_engine.hip, deflate_cont).
//   ins (per tile, two buffers): one bit per local position 32512 .. 65535: "in the chains", valid below the tile's exit -- read by the tile behind it,
//   for which these are its local positions 0 .. 33023.
// Block cuts, the window's slides and the stored-block veto are the stream's business (cont_table_kernel), not the tile's.
template <int CHAIN, int NICE, bool RING>
__global__ void __launch_bounds__(64) fastwin_tile_kernel(ChunkGeom g, TileGeom tg, FastTiles ft, uint32_t max_insert, const uint16_t *__restrict__ S_all, const uint32_t *__restrict__ ir_all,
                                                           uint32_t *__restrict__ tokens, ChunkMeta *__restrict__ meta)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fw_lds[];
    constexpr int NW = NICE / 8;
    constexpr uint32_t OFF_FLAGS = RING ? kFwOffFlags : kFwOffFlagsNR, OFF_STG = RING ? kFwOffStg : kFwOffStgNR;
    const uint32_t c = ft.list ? ft.list[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
    const bool warm = ft.warm_mode && c != 0;
    uint64_t wb; uint32_t n, h0, h1, nent;
    tile_span(g, tg, c, wb, n, h0, h1, nent);
    const uint8_t *src = g.in + wb;
    const uint64_t safe_end = g.in_bytes - wb;
    const uint16_t *S = S_all + (size_t)c * kSStrideF + kSPadF;
    const uint32_t *ir = ir_all + (size_t)c * kChunkMax;
    uint32_t *tok = tokens + (size_t)c * kChunkMax;
    const uint32_t base = (tg.abs0_nil && tg.abs0 + wb == 0) ? 0u : 1u, npos = n >= 3 ? n - 2 : 0;
    const uint32_t nil_local = (tg.nil_pos >= wb && tg.nil_pos - wb < kChunkMax) ? (uint32_t)(tg.nil_pos - wb) : ~0u;
    uint32_t *flags = reinterpret_cast<uint32_t *>(fw_lds + OFF_FLAGS);
    const uint32_t ring_a = fw_lds_base(fw_lds);
    const uint64_t lane_bit = 1ull << lane, lanes_below = lane_bit - 1;
    const uint32_t entry_pos = warm ? 0u : (c == 0 ? h0 + tg.entry[g.chunk0] : kTileStride + ft.exit_cur[c - 1]); // where this parse starts (the batch's first tile: the chain's hand-over)
    const uint32_t tok_from = warm ? h0 : 0u;                                                            // tokens in front of this position are the warm-up's
    const uint32_t *ins_in = warm ? nullptr : (c == 0 ? ft.prev_ins : (ft.cur[c - 1] ? ft.ins1 : ft.ins0) + (size_t)(c - 1) * kInsWords);

    // A tile that is parsed again because its predecessor's results changed usually finds nothing of that within its reach: the changes sit where the predecessor's
    // own start was a guess, ~32 KiB back.  With the same entry as last time and the highest changed history bit at position md, a token that starts behind
    // md + MAX_DIST cannot examine a changed bit (deflate.c:1037-1038, 1163: candidates at most MAX_DIST back) -- so once the parse has passed that point having
    // made, token for token, what it made last time, the rest of the tile comes out as before (same history within reach, same own bits, by induction) and the
    // parse stops: the tile's results stand.  A token that differs sends the parse to the tile's end as before.
    uint32_t stop_at = ~0u, old_ntok = 0, old_ent = 0;
    uint32_t *used = ft.used_ins ? ft.used_ins + (size_t)c * kInsWords : nullptr;
    if (used && !warm && c != 0 && ft.round != 0 && entry_pos == ft.entry_used[c]) {
        uint32_t md = 0;
        for (uint32_t i = lane; i < kInsWords; i += 64) {
            uint32_t x = ins_in[i] ^ used[i];
            const uint32_t lo = i * 32;
            if (lo >= entry_pos) x = 0; else if (entry_pos - lo < 32) x &= (1u << (entry_pos - lo)) - 1u; // (the bits are valid below the entry)
            if (x) { const uint32_t hi = lo + 32u - (uint32_t)__builtin_clz(x); md = hi > md ? hi : md; }  // (highest changed position + 1)
        }
        for (int o = 32; o > 0; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)md, o); md = y > md ? y : md; }
        stop_at = md ? md - 1 + kMaxDist : 0u;
        if (ft.stat && lane == 0) { atomicAdd(&ft.stat[0], 1u); if (stop_at >= h1) atomicAdd(&ft.stat[3], 1u); }
        old_ntok = meta[c].ntok; old_ent = tg.entry[g.chunk0 + c];
    }
    bool same = true; // the tokens made so far are the ones the tile had

    for (uint32_t i = lane; i < kFwFlagWords; i += 64) flags[i] = 0;
    __syncthreads();
    if (ins_in && !(stop_at == 0u)) { // the history's bits, by S index: every position in front of the entry that the predecessor found in the chains
        for (uint32_t q0 = 0; q0 < entry_pos; q0 += 256) {
            uint32_t iv[4]; bool on[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t q = q0 + u * 64 + lane;
                on[u] = q < entry_pos && q < npos && ((ins_in[q >> 5] >> (q & 31u)) & 1u);
                iv[u] = on[u] ? ir[q] : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) if (on[u]) { const uint32_t b = 65536u - (iv[u] & 0xffffu) - 1u; atomicOr(&flags[b >> 5], 1u << (b & 31u)); }
        }
    }

    uint32_t filled = 0;
    auto fill_to = [&](uint32_t need) {
        while (RING && filled < need) {
            const uint32_t o = filled + 16 * lane;
            uint4 v;
            if ((uint64_t)o + 16 <= safe_end) v = reinterpret_cast<const FwU128 *>(src + o)->v; else v = fw_tail16(src, o, safe_end);
            const uint32_t ro = fw_ring(filled) + 16 * lane;
            *reinterpret_cast<uint4 *>(fw_lds + ro) = v;
            if (ro < kFwMirror) *reinterpret_cast<uint4 *>(fw_lds + kFwRing + ro) = v;
            filled += 1024;
        }
    };
    auto load_ir = [&](uint32_t w) -> uint32_t { const uint32_t p = w * 64 + lane; return p < npos ? ir[p] : 0u; };
    uint4 sq0, sq1, sq2, sq3;
    auto load_s = [&](uint32_t iv) {
        const uint32_t rk = iv >> 16;
        const uint8_t *a = reinterpret_cast<const uint8_t *>(S) + 2 * (int)(iv & 0xffffu) - 64;
        sq0 = sq1 = sq2 = sq3 = make_uint4(0, 0, 0, 0);
        if (rk > 0) sq3 = reinterpret_cast<const FwU128 *>(a + 48)->v;
        if (rk > 8) sq2 = reinterpret_cast<const FwU128 *>(a + 32)->v;
        if (rk > 16) sq1 = reinterpret_cast<const FwU128 *>(a + 16)->v;
        if (rk > 24) sq0 = reinterpret_cast<const FwU128 *>(a)->v;
    };
    uint32_t pos = entry_pos, ntok = 0, ent_used = ~0u, tail_from = ~0u;
    const uint32_t win0 = pos / 64, nwin = (n + 63) / 64;
    uint32_t ir_cur = load_ir(win0), ir_nxt = load_ir(win0 + 1);
    load_s(ir_cur);
    bool cross_short = false;
    __syncthreads();

    for (uint32_t win = win0; win < nwin && pos < h1; win++) {
        if (stop_at != ~0u && same && pos > stop_at) { // (uniform) nothing that changed is within reach any more, and nothing has come out differently: the results stand
            for (uint32_t i = lane; i < kInsWords; i += 64) used[i] = ins_in[i];
            if (lane == 0) { ft.kept[c] = 1; ft.changed[c] = 0; ft.exit_new[c] = ft.exit_cur[c]; if (ft.stat) atomicAdd(&ft.stat[1], 1u); }
            return;
        }
        const uint32_t w0 = win * 64, p = w0 + lane;
        const bool skip = pos >= w0 + 64;
        if (!skip) {
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 0)) = sq0;
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 1)) = sq1;
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 2)) = sq2;
            *reinterpret_cast<uint4 *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, 3)) = sq3;
        }
        const uint32_t iv = ir_cur;
        load_s(ir_nxt);
        ir_cur = ir_nxt;
        ir_nxt = load_ir(win + 2);
        fill_to(w0 + 64 + kMaxMatch + NICE + 16 < kChunkMax + 1024 ? w0 + 64 + kMaxMatch + NICE + 16 : kChunkMax + 1024);
        if (skip) continue;

        const uint32_t idx = iv & 0xffffu, rk = iv >> 16;
        const bool haspos = p < npos;
        const uint32_t B0 = 65536u - idx;
        const uint32_t own_w = (B0 - 1) >> 5, own_b = 1u << ((B0 - 1) & 31u);
        const uint32_t entl = pos - w0;
        if (haspos && (lane >= entl || cross_short)) atomicOr(&flags[own_w], own_b);
        const uint32_t look = n > p ? n - p : 0, cap = look < kMaxMatch ? look : kMaxMatch, ni = (uint32_t)NICE < look ? (uint32_t)NICE : look;
        const uint32_t cmp_max = cap < (uint32_t)NICE ? cap : (uint32_t)NICE;
        const int w = (int)(p + base), limit = w > (int)kMaxDist ? w - (int)kMaxDist : 0;
        uint64_t own[NW];
        if (!RING) {
#pragma unroll
            for (int t = 0; t < NW; t++) own[t] = fw_g64(src, p + 8 * t, safe_end);
        } else {
            const uint32_t own_a = ring_a + fw_ring(p);
            if (NW == 1) own[0] = fw_ld64(own_a);
            else { fw_ld64x2(own_a, own_a + 8, own[0], own[1]); if (NW == 4) fw_ld64x2(own_a + 16, own_a + 24, own[2], own[3]); }
        }
        const uint32_t own_byte = (uint32_t)own[0] & 255u;

        uint32_t res = 1, mstart = 0, mex = 0;
        uint32_t lend = n - w0 < 64 ? n - w0 : 64;              // lanes of this window that are positions of the input ...
        if (h1 - w0 < lend) lend = h1 - w0;                     // ... and of the tile's range: a token that starts at or behind h1 is the next tile's
        auto read_bits = [&]() -> uint32_t {
            const uint32_t wa = B0 >> 5, lo32 = flags[wa], hi32 = flags[wa + 1];
            uint32_t m = __builtin_amdgcn_alignbit(hi32, lo32, B0 & 31u);
            if (rk < 32) m &= (1u << rk) - 1u;
            return m;
        };

        uint32_t start = entl;
        bool need_eval = true, dirty = false;
        for (;;) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (need_eval) dirty = false;
            if (need_eval && lane >= start) {
                res = 1; mstart = 0; mex = 0;
                if (haspos && rk != 0) {
                    uint32_t m = read_bits(), best = kMinMatch - 1, nsel = 0;
                    bool first = true, stopped = false, term = false;
#pragma unroll 1
                    for (int g0 = 0; g0 < CHAIN; g0 += 4) {
                        if (__builtin_amdgcn_ballot_w64(!stopped && m != 0) == 0) break;
                        uint32_t k[4], q[4]; bool v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { v[u] = m != 0; k[u] = v[u] ? (uint32_t)__builtin_ctz(m) : 0u; m &= m - 1u; }
#pragma unroll
                        for (int u = 0; u < 4; u++) { const uint32_t o = 62u - 2u * k[u]; q[u] = *reinterpret_cast<const uint16_t *>(fw_lds + OFF_STG + fw_stg_slot<RING>(lane, o >> 4) + (o & 15u)); }
                        uint64_t cb[4][NW];
                        if (!RING) {
#pragma unroll
                            for (int t = 0; t < NW; t++)
#pragma unroll
                                for (int u = 0; u < 4; u++) cb[u][t] = fw_g64(src, q[u] + 8 * t, safe_end);
                        } else {
                            const uint32_t a0 = ring_a + fw_ring(q[0]), a1 = ring_a + fw_ring(q[1]), a2 = ring_a + fw_ring(q[2]), a3 = ring_a + fw_ring(q[3]);
#pragma unroll
                            for (int t = 0; t < NW; t++) fw_ld64x4(a0 + 8 * t, a1 + 8 * t, a2 + 8 * t, a3 + 8 * t, cb[0][t], cb[1][t], cb[2][t], cb[3][t]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (v[u] && !stopped) {
                                mex |= 1u << k[u];
                                const int wq = (int)(q[u] + base);
                                const bool far = first ? (wq <= 0 || (uint32_t)(w - wq) > kMaxDist || (p == nil_local && (uint32_t)(w - wq) == kMaxDist)) : wq <= limit;
                                if (far) stopped = true;
                                else {
                                    first = false; nsel++;
                                    uint32_t l = 0;
#pragma unroll
                                    for (int t = 0; t < NW; t++) if (l == 8u * t) l += fw_diff8(cb[u][t] ^ own[t]);
                                    l = l < cmp_max ? l : cmp_max;
                                    if (l > best) { best = l; mstart = q[u]; if (l >= ni) { stopped = true; term = l < cap; } }
                                    if (nsel == (uint32_t)CHAIN) stopped = true;
                                }
                            }
                        }
                    }
                    if (!first && best >= kMinMatch) res = best | (term ? kResTerm : 0u);
                    if (!stopped && nsel < (uint32_t)CHAIN && rk > kFwDepth) res = kResInc;
                }
            }
            const uint64_t nonlit = __builtin_amdgcn_ballot_w64(res != 1u);
            uint32_t pk = (lane + (res & kResLen)) | (res & (kResTerm | kResInc));
            const uint32_t len_e = res & kResLen;
            const uint64_t longm = __builtin_amdgcn_ballot_w64(len_e >= kMinMatch && !(len_e <= max_insert && n - p - len_e >= kMinMatch));
            uint64_t T = 0;
            uint32_t L = start, stop_inc = 64, Llast = 64;
            while (L < lend) {
                const uint64_t ahead = (nonlit >> L) & (lend - L >= 64 ? ~0ull : (1ull << (lend - L)) - 1ull); // (a match behind the range's end starts no token of this tile)
                if (ahead == 0) { T |= (~0ull << L) & (lend >= 64 ? ~0ull : (1ull << lend) - 1ull); L = lend; Llast = 64; break; }
                const uint32_t run = (uint32_t)__builtin_ctzll(ahead);
                T |= ((1ull << run) - 1ull) << L;
                L += run;
                uint32_t x = __builtin_amdgcn_readlane(pk, L);
                if (x & (kResInc | kResTerm)) {
                    if (x & kResInc) { stop_inc = L; break; }
                    const uint32_t p0 = w0 + L, q0 = __builtin_amdgcn_readlane(mstart, L), cap0 = n - p0 < kMaxMatch ? n - p0 : kMaxMatch;
                    const uint32_t o = NICE + 4 * lane;
                    uint32_t xa, xb;
                    if (!RING) { xa = fw_g32(src, q0 + o, safe_end); xb = fw_g32(src, p0 + o, safe_end); }
                    else fw_ld32x2(ring_a + fw_ring(q0 + o), ring_a + fw_ring(p0 + o), xa, xb);
                    xa ^= xb;
                    const uint64_t ne = __builtin_amdgcn_ballot_w64(xa != 0);
                    uint32_t len = cap0;
                    if (ne != 0) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(ne), xf = __builtin_amdgcn_readlane(xa, f);
                        len = NICE + 4 * f + ((uint32_t)__builtin_ctz(xf) >> 3);
                        len = len < cap0 ? len : cap0;
                    }
                    res = lane == L ? len : res;
                    pk = lane == L ? L + len : pk;
                    x = L + len;
                }
                T |= 1ull << L;
                Llast = L;
                L = x & kResLen;
            }
            bool cs_new = cross_short;
            if (L >= 64 && Llast < 64) cs_new = !((longm >> Llast) & 1ull);
            uint64_t C = 0;
            {
                const uint64_t below = T & lanes_below;
                const uint32_t owner = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                const bool inside = below != 0 && !(T & lane_bit) && lane < L && ((longm >> owner) & 1ull);
                C = __builtin_amdgcn_ballot_w64(inside);
            }
            if ((C & lane_bit) && haspos) atomicAnd(&flags[own_w], ~own_b);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            dirty = dirty || C != 0;
            bool stale = false;
            if (dirty && lane >= start && mex != 0) stale = (mex & ~read_bits()) != 0;
            const uint64_t st_mask = __builtin_amdgcn_ballot_w64(stale) & T;
            const uint32_t Lstale = st_mask ? (uint32_t)__builtin_ctzll(st_mask) : 64u;
            const uint32_t Lacc = Lstale < stop_inc ? Lstale : stop_inc;
            const uint64_t Tacc = Lacc >= 64 ? T : T & ((1ull << Lacc) - 1ull);
            {   // the accepted tokens (the warm-up's, in front of h0, are dropped)
                const uint32_t lo_l = tok_from > w0 ? (tok_from - w0 < 64 ? tok_from - w0 : 64u) : 0u;
                const uint64_t Tv = lo_l >= 64 ? 0ull : Tacc & (~0ull << lo_l);
                bool neq = false;
                if (Tv & lane_bit) {
                    const uint32_t ln = res & kResLen, ti = ntok + (uint32_t)__builtin_popcountll(Tv & lanes_below);
                    const uint32_t nv = ln == 1 ? tok_lit(own_byte) : tok_match(p - mstart, ln - kMinMatch);
                    if (stop_at != ~0u && same) neq = ti >= old_ntok || tok[ti] != nv;
                    tok[ti] = nv;
                }
                if (stop_at != ~0u && same && __builtin_amdgcn_ballot_w64(neq) != 0) { same = false; if (ft.stat && lane == 0) { atomicAdd(&ft.stat[2], 1u); atomicAdd(&ft.stat[4], (w0 - (entry_pos & ~63u)) >> 6); atomicAdd(&ft.stat[5], (h1 - entry_pos) >> 6); } }
                if (ent_used == ~0u && Tv != 0) { ent_used = w0 + (uint32_t)__builtin_ctzll(Tv) - h0; if (stop_at != ~0u && ent_used != old_ent) same = false; }
                ntok += (uint32_t)__builtin_popcountll(Tv);
            }
            if (Lacc >= 64 || Lacc >= lend) {
                pos = w0 + L; cross_short = cs_new;
                // the tile's last token, when it is a short match that reaches into the next window: its strings ARE in the chains (deflate.c:1510-1520), and no
                // window of this tile will set their bits
                tail_from = (L > 64 && Llast < 64 && !((longm >> Llast) & 1ull)) ? w0 + 64 : ~0u;
                break;
            }
            if (lane >= Lacc && (C & lane_bit) && haspos) atomicOr(&flags[own_w], own_b);
            start = Lacc;
            if (Lstale <= stop_inc) { need_eval = true; continue; }
            {   // the search at Lacc over its whole bucket, by all lanes
                const uint32_t p0 = w0 + Lacc, i0 = __builtin_amdgcn_readlane(idx, Lacc), r0 = __builtin_amdgcn_readlane(rk, Lacc);
                const uint32_t look0 = n - p0, cap0 = look0 < kMaxMatch ? look0 : kMaxMatch, ni0 = (uint32_t)NICE < look0 ? (uint32_t)NICE : look0;
                const int w00 = (int)(p0 + base), limit0 = w00 > (int)kMaxDist ? w00 - (int)kMaxDist : 0;
                uint32_t best = kMinMatch - 1, ms0 = 0, ch = CHAIN;
                bool first = true, done = false;
                const uint32_t pa = ring_a;
                for (uint32_t k0 = 0; k0 < r0 && !done; k0 += 64) {
                    const uint32_t kk = k0 + lane;
                    const bool valid = kk < r0;
                    const uint32_t q = valid ? S[(int)i0 - 1 - (int)kk] : 0u;
                    const uint32_t rv = 65536u - i0 + kk;
                    const bool ins = valid && ((flags[rv >> 5] >> (rv & 31u)) & 1u);
                    uint32_t l = 0;
                    if (ins && (int)(q + base) > limit0 - 1) {
                        for (;;) {
                            const uint32_t d = RING ? fw_diff8(fw_ld64(pa + fw_ring(q + l)) ^ fw_ld64(pa + fw_ring(p0 + l))) : fw_diff8(fw_g64(src, q + l, safe_end) ^ fw_g64(src, p0 + l, safe_end));
                            l += d;
                            if (d < 8 || l >= cap0) break;
                        }
                        l = l < cap0 ? l : cap0;
                    }
                    uint64_t im = __builtin_amdgcn_ballot_w64(ins);
                    while (im) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(im); im &= im - 1;
                        const uint32_t qf = __builtin_amdgcn_readlane(q, f), lf = __builtin_amdgcn_readlane(l, f);
                        const int wq = (int)(qf + base);
                        if (first ? (wq <= 0 || (uint32_t)(w00 - wq) > kMaxDist || (p0 == nil_local && (uint32_t)(w00 - wq) == kMaxDist)) : wq <= limit0) { done = true; break; }
                        first = false;
                        if (lf > best) { best = lf; ms0 = qf; if (lf >= ni0) { done = true; break; } }
                        if (--ch == 0) { done = true; break; }
                    }
                }
                const uint32_t r1 = (!first && best >= kMinMatch) ? best : 1u;
                if (lane == Lacc) { res = r1; mstart = ms0; mex = 0; }
            }
            need_eval = false;
        }
    }
    // ---- what the tile hands on: where its parse ends, which of its positions are in the chains ----
    if (tail_from != ~0u && pos >= h1) { const uint32_t q = tail_from + lane; if (q < pos && q < npos) { const uint32_t b = 65536u - (ir[q] & 0xffffu) - 1u; atomicOr(&flags[b >> 5], 1u << (b & 31u)); } }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (ent_used == ~0u) ent_used = (pos > h0 ? pos : h0) - h0; // (no token: the parse entered at or behind the range's end)
    const uint32_t exit_k = (pos > h1 ? pos : h1) - h1;
    const uint32_t *old_ins = (ft.cur[c] ? ft.ins1 : ft.ins0) + (size_t)c * kInsWords;
    uint32_t *new_ins = (ft.cur[c] ? ft.ins0w : ft.ins1w) + (size_t)c * kInsWords;
    bool diff = exit_k != ft.exit_cur[c];
    uint32_t dbg_first = ~0u, dbg_last = 0, dbg_n = 0;
    for (uint32_t q0 = kTileStride; q0 < kChunkMax; q0 += 64) {
        const uint32_t q = q0 + lane;
        bool on = false;
        if (q < pos && q < npos) { const uint32_t b = 65536u - (ir[q] & 0xffffu) - 1u; on = (flags[b >> 5] >> (b & 31u)) & 1u; }
        const uint64_t m = __builtin_amdgcn_ballot_w64(on);
        const uint32_t wi = (q0 - kTileStride) >> 5;
        if (lane == 0) {
            const bool dw = old_ins[wi] != (uint32_t)m || old_ins[wi + 1] != (uint32_t)(m >> 32);
            if (dw) { if (dbg_first == ~0u) dbg_first = q0; dbg_last = q0; dbg_n++; }
            diff = diff || dw; new_ins[wi] = (uint32_t)m; new_ins[wi + 1] = (uint32_t)(m >> 32);
        }
    }
    if (c == 0 && g.chunk0 == 0 && ft.low_out) { // (uniform)
        for (uint32_t q0 = 0; q0 < kTileStride; q0 += 64) {
            const uint32_t q = q0 + lane;
            bool on = false;
            if (q < pos && q < npos) { const uint32_t b = 65536u - (ir[q] & 0xffffu) - 1u; on = (flags[b >> 5] >> (b & 31u)) & 1u; }
            const uint64_t m = __builtin_amdgcn_ballot_w64(on);
            if (lane == 0) { ft.low_out[q0 >> 5] = (uint32_t)m; ft.low_out[(q0 >> 5) + 1] = (uint32_t)(m >> 32); }
        }
        for (uint32_t i = kTileStride / 32 + lane; i < kInsWords; i += 64) ft.low_out[i] = 0;
    }
    if (used) { // what these results were made from
        if (warm) { // the warm-up's own account of the history
            for (uint32_t q0 = 0; q0 < kInsWords * 32; q0 += 64) {
                const uint32_t q = q0 + lane;
                bool on = false;
                if (q < pos && q < npos) { const uint32_t b = 65536u - (ir[q] & 0xffffu) - 1u; on = (flags[b >> 5] >> (b & 31u)) & 1u; }
                const uint64_t m = __builtin_amdgcn_ballot_w64(on);
                if (lane == 0) { used[q0 >> 5] = (uint32_t)m; used[(q0 >> 5) + 1] = (uint32_t)(m >> 32); }
            }
        } else if (ins_in && c != 0) for (uint32_t i = lane; i < kInsWords; i += 64) used[i] = ins_in[i];
        if (lane == 0) { ft.entry_used[c] = (uint16_t)(warm ? h0 + ent_used : entry_pos); ft.kept[c] = 0; }
    }
    if (lane == 0 && ft.dbg) { uint32_t *d = ft.dbg + (size_t)c * 8; d[0] = ft.round; d[1] = dbg_n; d[2] = dbg_first; d[3] = dbg_last; d[4] = exit_k; d[5] = ft.exit_cur[c]; d[6] = entry_pos; d[7] = ntok; }
    if (lane == 0) {
        meta[c].ntok = ntok; meta[c].in_bytes = 0;
        tg.entry[g.chunk0 + c] = (uint16_t)ent_used;
        ft.exit_new[c] = (uint16_t)exit_k;
        ft.changed[c] = diff ? 1 : 0;
    }
}

// Which form a launch takes: with fewer chunks than three per CU the ring form (a chunk 5.1 ms instead of 5.6: its latency is what a small call pays), from there on the one
// without the ring (eleven chunks per CU: 256 MiB 32.9 -> 19.3 ms, 1 GiB 116 -> 59, 4 GiB 450 -> 217 at level 1).  ZGPU_FW_RING=1 / 0 forces one of them.
static bool fw_use_ring(uint32_t nlaunch)
{
    static int v = -2;
    if (v == -2) { const char *e = getenv("ZGPU_FW_RING"); v = e ? atoi(e) : -1; }
    return v < 0 ? nlaunch < 768 : v != 0;
}
// the levels' own parameters only (deflate.c:137-149): a tuned stream goes to the lane-per-chunk loop
bool lz_fastwin_serves(const LevelCfg &cfg)
{
    if (cfg.slow || cfg.good != 4) return false;
    return (cfg.chain == 4 && cfg.nice == 8) || (cfg.chain == 8 && cfg.nice == 16) || (cfg.chain == 32 && cfg.nice == 32);
}

void launch_lz_fastwin(const ChunkGeom &g, LevelCfg cfg, const uint16_t *S, const uint32_t *ir, uint32_t *tokens, ChunkMeta *meta, hipStream_t st)
{
    static bool opt_in = false;
    if (!opt_in) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_kernel<4, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_kernel<8, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_kernel<32, 32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        opt_in = true;
    }
    if (!fw_use_ring(g.nchunks)) {
        if (cfg.chain == 4) hipLaunchKernelGGL((fastwin_kernel<4, 8, false>), dim3(g.nchunks), dim3(64), kFwLdsNR, st, g, cfg.lazy, S, ir, tokens, meta);
        else if (cfg.chain == 8) hipLaunchKernelGGL((fastwin_kernel<8, 16, false>), dim3(g.nchunks), dim3(64), kFwLdsNR, st, g, cfg.lazy, S, ir, tokens, meta);
        else hipLaunchKernelGGL((fastwin_kernel<32, 32, false>), dim3(g.nchunks), dim3(64), kFwLdsNR, st, g, cfg.lazy, S, ir, tokens, meta);
        return;
    }
    if (cfg.chain == 4) hipLaunchKernelGGL((fastwin_kernel<4, 8, true>), dim3(g.nchunks), dim3(64), kFwLds, st, g, cfg.lazy, S, ir, tokens, meta);
    else if (cfg.chain == 8) hipLaunchKernelGGL((fastwin_kernel<8, 16, true>), dim3(g.nchunks), dim3(64), kFwLds, st, g, cfg.lazy, S, ir, tokens, meta);
    else hipLaunchKernelGGL((fastwin_kernel<32, 32, true>), dim3(g.nchunks), dim3(64), kFwLds, st, g, cfg.lazy, S, ir, tokens, meta);
}

// ---- the rounds' bookkeeping (one lane per tile) ----
__global__ void __launch_bounds__(256) fast_init_kernel(uint8_t *cur, uint8_t *active, uint16_t *exit_cur, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { cur[i] = 0; active[i] = 1; exit_cur[i] = 0xffffu; }
}
// behind a round: the tiles that were parsed make their new results current; a tile is parsed again when its predecessor's results have changed
__global__ void __launch_bounds__(256) fast_flip_kernel(uint8_t *cur, const uint8_t *active, uint8_t *active_next, const uint8_t *changed, uint16_t *exit_cur, const uint16_t *exit_new,
                                                        uint32_t n, uint32_t round, uint32_t *count, uint32_t *list, const uint8_t *kept)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (round == 0 || (active[i] && !(kept && kept[i]))) { cur[i] ^= 1; exit_cur[i] = exit_new[i]; } // (a tile whose parse stopped early keeps its results where they are)
    const bool nx = i >= 1 && (round == 0 || (active[i - 1] && changed[i - 1]));
    active_next[i] = nx ? 1 : 0;
    if (nx) list[atomicAdd(count, 1u)] = i; // (any order: the tiles of a round do not depend on each other)
}
// behind the last round of a batch: the entry of the tile behind the batch, and the batch's last bits as the next batch's history
__global__ void __launch_bounds__(256) fast_finish_kernel(const uint8_t *cur, const uint16_t *exit_cur, const uint32_t *ins0, const uint32_t *ins1, uint32_t n, uint16_t *entry_after,
                                                          uint32_t *prev_ins, uint32_t *prev_prev_ins, const uint32_t *low)
{
    const uint32_t tid = threadIdx.x;
    if (tid == 0) *entry_after = exit_cur[n - 1];
    // (the bits in front of the last tile: the tile before it, or what was in front of the batch -- kept for the feed's hand-over, fast_hist_kernel)
    if (n >= 2) { const uint32_t *b = (cur[n - 2] ? ins1 : ins0) + (size_t)(n - 2) * kInsWords; for (uint32_t i = tid; i < kInsWords; i += 256) prev_prev_ins[i] = b[i]; }
    else for (uint32_t i = tid; i < kInsWords; i += 256) prev_prev_ins[i] = low ? low[i] : prev_ins[i]; // (the feed's first tile: its own account of its positions 0 .. 32511)
    __syncthreads();
    const uint32_t *a = (cur[n - 1] ? ins1 : ins0) + (size_t)(n - 1) * kInsWords;
    for (uint32_t i = tid; i < kInsWords; i += 256) prev_ins[i] = a[i];
}
// the feed's hand-over: "in the chains" for the 32512 positions in front of where the parse stands (bit j: position new_w0 + j), from the bits of the
// last tile (its local positions 32512 ..) and of what lay in front of it (its local positions 0 .. 33023).  x0: new_w0 in the last tile's local coordinates.
__global__ void __launch_bounds__(256) fast_hist_kernel(const uint32_t *before, const uint32_t *last, uint32_t x0, uint32_t count, uint32_t *out)
{
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= kInsWords) return;
    uint32_t v = 0;
    for (uint32_t b = 0; b < 32; b++) {
        const uint32_t j = w * 32 + b, x = x0 + j;
        if (j >= count) break;
        const bool on = x < kTileStride ? ((before[x >> 5] >> (x & 31u)) & 1u) : x < kChunkMax ? ((last[(x - kTileStride) >> 5] >> ((x - kTileStride) & 31u)) & 1u) : false;
        v |= (on ? 1u : 0u) << b;
    }
    out[w] = v;
}
// behind a launch over a list of tiles that all make their new results current (the phases of round 0, zgpu_engine.hip lz_tiles_fast)
__global__ void __launch_bounds__(256) fast_flip_list_kernel(uint8_t *cur, uint16_t *exit_cur, const uint16_t *exit_new, const uint32_t *list, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = list[i];
    cur[c] ^= 1; exit_cur[c] = exit_new[c];
}
void launch_fast_flip_list(uint8_t *cur, uint16_t *exit_cur, const uint16_t *exit_new, const uint32_t *list, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(fast_flip_list_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cur, exit_cur, exit_new, list, n);
}
void launch_fast_init(uint8_t *cur, uint8_t *active, uint16_t *exit_cur, uint32_t n, hipStream_t st) { hipLaunchKernelGGL(fast_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cur, active, exit_cur, n); }
void launch_fast_flip(uint8_t *cur, const uint8_t *active, uint8_t *active_next, const uint8_t *changed, uint16_t *exit_cur, const uint16_t *exit_new, uint32_t n, uint32_t round,
                      uint32_t *count, uint32_t *list, const uint8_t *kept, hipStream_t st)
{
    hipMemsetAsync(count, 0, 4, st);
    hipLaunchKernelGGL(fast_flip_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cur, active, active_next, changed, exit_cur, exit_new, n, round, count, list, kept);
}
void launch_fast_finish(const uint8_t *cur, const uint16_t *exit_cur, const uint32_t *ins0, const uint32_t *ins1, uint32_t n, uint16_t *entry_after, uint32_t *prev_ins, uint32_t *prev_prev_ins,
                        const uint32_t *low, hipStream_t st)
{
    hipLaunchKernelGGL(fast_finish_kernel, dim3(1), dim3(256), 0, st, cur, exit_cur, ins0, ins1, n, entry_after, prev_ins, prev_prev_ins, low);
}
void launch_fast_hist(const uint32_t *before, const uint32_t *last, uint32_t x0, uint32_t count, uint32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(fast_hist_kernel, dim3((kInsWords + 255) / 256), dim3(256), 0, st, before, last, x0, count, out);
}

void launch_lz_fastwin_tiles(const ChunkGeom &g0, const TileGeom &tg, const FastTiles &ft, LevelCfg cfg, const uint16_t *S, const uint32_t *ir, uint32_t *tokens, ChunkMeta *meta, uint32_t ngrid,
                             hipStream_t st)
{
    ChunkGeom g = g0; // (g.nchunks stays the batch's: the launch is over `ngrid` of its tiles)
    static bool opt_in = false;
    if (!opt_in) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_tile_kernel<4, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_tile_kernel<8, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_tile_kernel<32, 32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        opt_in = true;
    }
    if (!fw_use_ring(ngrid)) {
        if (cfg.chain == 4) hipLaunchKernelGGL((fastwin_tile_kernel<4, 8, false>), dim3(ngrid), dim3(64), kFwLdsNR, st, g, tg, ft, cfg.lazy, S, ir, tokens, meta);
        else if (cfg.chain == 8) hipLaunchKernelGGL((fastwin_tile_kernel<8, 16, false>), dim3(ngrid), dim3(64), kFwLdsNR, st, g, tg, ft, cfg.lazy, S, ir, tokens, meta);
        else hipLaunchKernelGGL((fastwin_tile_kernel<32, 32, false>), dim3(ngrid), dim3(64), kFwLdsNR, st, g, tg, ft, cfg.lazy, S, ir, tokens, meta);
        return;
    }
    if (cfg.chain == 4) hipLaunchKernelGGL((fastwin_tile_kernel<4, 8, true>), dim3(ngrid), dim3(64), kFwLds, st, g, tg, ft, cfg.lazy, S, ir, tokens, meta);
    else if (cfg.chain == 8) hipLaunchKernelGGL((fastwin_tile_kernel<8, 16, true>), dim3(ngrid), dim3(64), kFwLds, st, g, tg, ft, cfg.lazy, S, ir, tokens, meta);
    else hipLaunchKernelGGL((fastwin_tile_kernel<32, 32, true>), dim3(ngrid), dim3(64), kFwLds, st, g, tg, ft, cfg.lazy, S, ir, tokens, meta);
}

} // namespace zgpu
