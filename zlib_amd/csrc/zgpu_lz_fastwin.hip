// zgpu_lz_fastwin.hip -- LZ77 stage for levels 1-3: deflate_fast (/root/reference/qcsrc/deflate.c:1448-1546, longest_match :1027-1168),
// one WORKGROUP per chunk, the chunk's window and its hash chains in LDS, 64 positions per wave and step, the waves ahead of the parse.
//
// The chains of deflate_fast depend on the parse: the strings inside a match longer than max_insert_length never enter them
// (deflate.c:1510-1534).  So neither head[]/prev[] filled ahead of the parse nor the static chains of levels 4-9 apply.  What IS known ahead of
// the parse is which positions share a hash: sort3_kernel has counting-sorted the positions by hash (S; idx(p), rank(p) in `ir`), the bucket
// predecessors of p, nearest first, are S[idx-1], S[idx-2], ... S[idx-rank], and the chain of p is those of them that have been inserted.
// "Inserted" is ONE BIT PER S INDEX, kept in LDS (8 KiB): the bits of p's 32 nearest predecessors are 32 consecutive bits -- one read, and
// the first `max_chain_length` set ones are the chain.  The bytes the candidates are compared with come from a ring of the last 36 KiB of the
// chunk in LDS (MAX_DIST + lookahead); the S entries and ir words of a window are fetched from HBM ahead of the parse (they do not depend on
// it), so no trip to memory lies on the chain of dependent steps.
//
// A window (64 positions, one wave): every lane evaluates longest_match at its own position under the bits as they stand; a walk follows the
// token starts; a match longer than max_insert_length clears the bits of its inside; then every lane compares the bits it SAW with the bits as
// they are now, over the part of them its search examined: a token start for which they differ is stale, the tokens in front of the first stale
// one stand, everything from there on is evaluated again.  A search that needs more than 32 predecessors (1.5 % at level 1) is done over the
// whole bucket by all lanes when the walk stands on it.
//
// A lone wave per SIMD issues an instruction every five to six cycles, and 47 KiB of LDS per chunk admit three chunks per CU: one wave per chunk
// is 12 000 cycles per window (measured), slower than the lane-per-chunk loop.  So the W waves of a workgroup share one chunk's ring and bits and
// take the windows in turn (wave w: windows w, w+W, ...), each AHEAD of the parse: it evaluates and walks its window from a guessed entry
// position (the exit the wave in front has published, if it has) under whatever the bits of the windows in front hold at the time.  When its
// turn comes -- the windows in front are final, the true entry is known -- it walks again from the true entry over the lengths it has, brings
// its own bits in line and lets the lanes make the comparison above: what stands is the reference's parse whatever the guesses were
// (tests/tools/fastwin_spec_model.c: this protocol on the CPU, fed wrong entries and damaged bits, against the plain loop).  Only that check --
// a walk, one read of the bits -- is serial; a wrong guess costs an evaluation at the turn.
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

#ifdef ZGPU_FW_DUMP // debug builds only (scripts/fw_tokens.py): the tokens of the launch's first chunk
#ifndef ZGPU_FW_DUMP_WIN
#define ZGPU_FW_DUMP_WIN 16
#endif
__device__ uint32_t g_fw_dbg[64 * 8 + 64];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_fw_lanes(uint32_t *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fw_dbg), sizeof(uint32_t) * (64 * 8 + 64)); }
static uint32_t g_fw_dump_tok[65536]; static zgpu::ChunkMeta g_fw_dump_meta;
extern "C" __attribute__((visibility("default"))) uint32_t zgpu_debug_fw_tokens(uint32_t *out) { for (uint32_t i = 0; i < g_fw_dump_meta.ntok; i++) out[i] = g_fw_dump_tok[i]; return g_fw_dump_meta.ntok; }
#endif

namespace zgpu {

#ifndef ZGPU_FW_WAVES
#define ZGPU_FW_WAVES 6
#endif
constexpr uint32_t kFwW = ZGPU_FW_WAVES;            // waves per chunk
constexpr uint32_t kSPadF = 8, kSStrideF = kChunkMax + kSPadF; // S layout of zgpu_lz_sorted.hip (kSPad entries in front of every chunk's S)
#ifndef ZGPU_FW_RING
#define ZGPU_FW_RING 36
#endif
constexpr uint32_t kFwRingBlocks = ZGPU_FW_RING, kFwRing = kFwRingBlocks * 1024; // bytes of the chunk in LDS: MAX_DIST back, the windows in flight + MAX_MATCH ahead; 1 KiB blocks
constexpr uint32_t kFwMirror = 48;                  // the ring's first bytes again behind its end: a 40-byte read (nice_match 32 + 8) may start at its last byte
constexpr uint32_t kFwFlagWords = kChunkMax / 32 + 2;
constexpr uint32_t kFwStgStride = 80;               // bytes per lane of the staged S entries (64 + 16: 16-byte stores free of bank conflicts)
constexpr uint32_t kFwOffFlags = kFwRing + kFwMirror;
constexpr uint32_t kFwOffCtl = (kFwOffFlags + kFwFlagWords * 4 + 15) & ~15u;
constexpr uint32_t kFwCtlWords = 32 + 2 * kFwW;
constexpr uint32_t kFwOffStg = (kFwOffCtl + kFwCtlWords * 4 + 15) & ~15u;
constexpr uint32_t kFwLds = kFwOffStg + kFwW * 64 * kFwStgStride;
constexpr uint32_t kFwDepth = 32;                   // predecessors a lane looks at
constexpr int kFwExtSteps = 4;                      // 8-byte steps a lane measures of a match that reached nice_match; a longer one is measured by all lanes together
constexpr uint32_t kResLen = 511, kResInc = 1u << 10, kResLong = 1u << 11; // a lane's search: length; "needs the whole bucket"; "longer than measured"
// control words (LDS)
enum { C_TURN = 0, C_CLAIMED, C_READY0, C_READY1, C_READY2, C_POS, C_CS, C_NTOK, C_BLKTOK0, C_NBLK, C_NOSTORE, C_BLKSTART, C_OFF, C_BUFFERED, C_SPEC = 32 };

struct __attribute__((packed, aligned(1))) FwU128 { uint4 v; };

__device__ inline uint32_t fw_lds_base(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }
// 8 / 4 bytes at any LDS byte offset (the hardware serves unaligned ds reads; their 36 clocks in the LDS cost one issue slot)
__device__ inline uint64_t fw_ld64(uint32_t a) { uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory"); return v; }
__device__ inline uint32_t fw_ld32(uint32_t a) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory"); return v; }
__device__ inline void fw_st32(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ inline void fw_ld64x2(uint32_t a, uint32_t b, uint64_t &x, uint64_t &y)
{
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(a), "v"(b) : "memory");
}
__device__ inline void fw_ld64x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint64_t &x, uint64_t &y, uint64_t &z, uint64_t &w)
{
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w) : "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
__device__ inline uint32_t fw_ring(uint32_t p) { const uint32_t d = p - kFwRing; return p < d ? p : d; } // p mod ring size, p < 2 * ring
__device__ inline uint32_t fw_diff8(uint64_t x) { return x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u; }

__device__ __noinline__ uint4 fw_tail16(const uint8_t *in, uint32_t o, uint64_t safe_end)
{
    uint32_t v[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16 && (uint64_t)o + k < safe_end; k++) v[k >> 2] |= (uint32_t)in[o + k] << (8 * (k & 3));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// CHAIN: max_chain_length; NICE: nice_match (a multiple of 8: the lanes compare that many bytes of every candidate, then measure on)
template <int CHAIN, int NICE>
__global__ void __launch_bounds__(64 * kFwW) fastwin_kernel(ChunkGeom g, uint32_t max_insert, const uint16_t *__restrict__ S_all, const uint32_t *__restrict__ ir_all,
                                                             uint32_t *__restrict__ tokens, ChunkMeta *__restrict__ meta)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fw_lds[];
    static_assert(NICE % 8 == 0 && NICE >= 8 && NICE <= 32 && CHAIN % 4 == 0, "whole groups of four candidates, whole 8-byte steps");
    constexpr int NW = NICE / 8;
    const uint32_t c = blockIdx.x, lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint8_t *src = g.in + lo;
    const uint64_t safe_end = g.in_bytes - lo;
    const uint16_t *S = S_all + (size_t)c * kSStrideF + kSPadF;
    const uint32_t *ir = ir_all + (size_t)c * kChunkMax;
    uint32_t *tok = tokens + (size_t)c * kChunkMax;
    const uint32_t base = chunk_base(g, c), npos = n >= 3 ? n - 2 : 0;
    uint32_t *flags = reinterpret_cast<uint32_t *>(fw_lds + kFwOffFlags), *ctl = reinterpret_cast<uint32_t *>(fw_lds + kFwOffCtl);
    const uint32_t ring_a = fw_lds_base(fw_lds), ctl_a = fw_lds_base(fw_lds + kFwOffCtl);
    uint8_t *stg = fw_lds + kFwOffStg + wave * (64 * kFwStgStride) + lane * kFwStgStride;
    const uint64_t lane_bit = 1ull << lane, lanes_below = lane_bit - 1, lanes_upto = lanes_below | lane_bit;
    const uint32_t nwin = (n + 63) / 64;
    const uint32_t room = 2 * kWSize - base;

    for (uint32_t i = threadIdx.x; i < kFwFlagWords; i += 64 * kFwW) flags[i] = 0;
    if (threadIdx.x < kFwCtlWords) {
        uint32_t v = 0;
        if (threadIdx.x == C_BUFFERED) v = n < room ? n : room; // first fill_window (deflate.c:1275,1342)
        if (threadIdx.x >= C_SPEC && threadIdx.x < C_SPEC + kFwW) v = ~0u; // (window numbers of the published exits: none yet)
        reinterpret_cast<uint32_t *>(fw_lds + kFwOffCtl)[threadIdx.x] = v;
    }
    __syncthreads();
    if (nwin == 0) { if (threadIdx.x == 0) { meta[c].ntok = 0; meta[c].nostore = 0; meta[c].in_bytes = n; } return; }
    if (wave >= nwin) return;
    FW_T0();

    // ---- the ring: block k (1 KiB of the chunk) lives at (k mod 36) KiB.  Blocks are claimed in order by whichever wave gets there first; a block
    //      may be loaded once the bytes it overwrites lie more than MAX_DIST behind the window whose turn it is ----
    const uint32_t nblocks = (((n + kMaxMatch + NICE + 80 < kChunkMax + 1024 ? n + kMaxMatch + NICE + 80 : kChunkMax + 1024)) + 1023) / 1024;
    auto ring_ahead = [&](uint32_t win) { // claim and load what the windows up to win + W need (never waits)
        const uint32_t want0 = (64 * (win + kFwW) + 64 + kMaxMatch + NICE + 16 + 1023) / 1024, want = want0 < nblocks ? want0 : nblocks;
        for (;;) {
            // (every branch of this loop is taken by all lanes or by none: a value that only lane 0 computes, handed on with readfirstlane and
            // followed by an `if (lane == 0)` block at the loop's end, compiled into a loop that only lane 0 stayed in)
            const uint32_t cl = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_CLAIMED)), turn = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_TURN));
            if (!(cl < want && (cl < kFwRingBlocks || 1024 * (cl + 1) - kFwRing + kMaxDist + 8 <= 64 * turn))) break;
            // all lanes try to move the counter from cl to cl + 1: at most one of them finds cl there (the lanes of one instruction are served
            // one after the other), none if another wave was faster
            uint32_t old;
            asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(ctl_a + 4 * C_CLAIMED), "v"(cl), "v"(cl + 1) : "memory");
            if (__builtin_amdgcn_ballot_w64(old == cl) == 0) continue;
            const uint32_t k = cl;
            const uint32_t o = 1024 * k + 16 * lane;
            uint4 v;
            if ((uint64_t)o + 16 <= safe_end) v = reinterpret_cast<const FwU128 *>(src + o)->v; else v = fw_tail16(src, o, safe_end);
            const uint32_t ro = 1024 * (k % kFwRingBlocks) + 16 * lane;
            *reinterpret_cast<uint4 *>(fw_lds + ro) = v;
            if (ro < kFwMirror) *reinterpret_cast<uint4 *>(fw_lds + kFwRing + ro) = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_or_b32 %0, %1" ::"v"(ctl_a + 4 * (C_READY0 + (k >> 5))), "v"(1u << (k & 31u)) : "memory"); // (all lanes, the same bit)
        }
    };
    auto ring_wait = [&](uint32_t win) { // until the blocks window `win` reads are in the ring
        const uint32_t need0 = (64 * win + 64 + kMaxMatch + NICE + 16 + 1023) / 1024, need = need0 < nblocks ? need0 : nblocks;
        for (;;) {
            const uint32_t r0 = fw_ld32(ctl_a + 4 * C_READY0), r1 = fw_ld32(ctl_a + 4 * C_READY1), r2 = fw_ld32(ctl_a + 4 * C_READY2);
            const uint32_t have = ~r0 ? (uint32_t)__builtin_ctz(~r0) : ~r1 ? 32u + (uint32_t)__builtin_ctz(~r1) : 64u + (uint32_t)__builtin_ctz(~r2);
            if (__builtin_amdgcn_readfirstlane(have) >= need) break;
            ring_ahead(win);
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // ---- block bookkeeping of the parse (cut_block of zgpu_lz_serial.hip; deflate.c:1365-1367 for `nostore`): handed from turn to turn ----
    uint32_t off = 0, ntok = 0, blk_tok0 = 0, nblk = 0, nostore = 0, block_start = 0, buffered = 0;
    auto token_top = [&](uint32_t p) { // the loop top of deflate_fast for the token that starts at p: fill_window and its slide (deflate.c:1293)
        if (buffered - p < kMinLookahead) { if ((int)(p + base) - (int)off >= (int)(kWSize + kMaxDist)) off += kWSize; buffered = n; }
    };
    auto cut_block = [&](uint32_t p_end) {
        if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk;
        nblk++; blk_tok0 += kBlockTokens; block_start = p_end;
    };

    // ---- what is fetched ahead: ir of this wave's window after next, the S entries of its next window ----
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
    uint32_t ir_cur = load_ir(wave), ir_nxt = load_ir(wave + kFwW);
    load_s(ir_cur);

    for (uint32_t win = wave; win < nwin; win += kFwW) {
        const uint32_t w0 = win * 64, p = w0 + lane;
        *reinterpret_cast<uint4 *>(stg) = sq0;
        *reinterpret_cast<uint4 *>(stg + 16) = sq1;
        *reinterpret_cast<uint4 *>(stg + 32) = sq2;
        *reinterpret_cast<uint4 *>(stg + 48) = sq3;
        const uint32_t iv = ir_cur;
        load_s(ir_nxt);
        ir_cur = ir_nxt;
        ir_nxt = load_ir(win + 2 * kFwW);
        ring_ahead(win);
        ring_wait(win);
        FW_LAP(0);
        FW_CNT(8, 1);

        const uint32_t idx = iv & 0xffffu, rk = iv >> 16;
        const bool haspos = p < npos;
        const uint32_t B0 = 65536u - idx;                       // bit B0 + k of the (reversed) bitmap belongs to predecessor k
        const uint32_t own_w = (B0 - 1) >> 5, own_b = 1u << ((B0 - 1) & 31u);
        const uint32_t look = n > p ? n - p : 0, cap = look < kMaxMatch ? look : kMaxMatch, ni = (uint32_t)NICE < look ? (uint32_t)NICE : look;
        const uint32_t cmp_max = cap < (uint32_t)NICE ? cap : (uint32_t)NICE;
        const int w = (int)(p + base), limit = w > (int)kMaxDist ? w - (int)kMaxDist : 0;
        uint64_t own[NW];
        const uint32_t own_a = ring_a + fw_ring(p);
        if (NW == 1) own[0] = fw_ld64(own_a);
        else { fw_ld64x2(own_a, own_a + 8, own[0], own[1]); if (NW == 4) fw_ld64x2(own_a + 16, own_a + 24, own[2], own[3]); }
        const uint32_t own_byte = (uint32_t)own[0] & 255u;
        const uint32_t lend = n - w0 < 64 ? n - w0 : 64; // lanes of this window that are positions of the chunk

        // this lane's search: length | flags; position behind its token; where the match starts; the bits it saw and the part of them it examined
        uint32_t res = 1, nxt = lane + 1, mstart = 0, seen = 0, range = 0;
        bool mybit = false; // this lane's bit in the bitmap as it stands (nobody else writes it)
        auto set_bit = [&](bool d) {
            if (haspos && d != mybit) { if (d) atomicOr(&flags[own_w], own_b); else atomicAnd(&flags[own_w], ~own_b); mybit = d; }
        };
        auto read_bits = [&]() -> uint32_t {
            const uint32_t wa = B0 >> 5, lo32 = flags[wa], hi32 = flags[wa + 1];
            uint32_t m = __builtin_amdgcn_alignbit(hi32, lo32, B0 & 31u);
            if (rk < 32) m &= (1u << rk) - 1u;
            return m;
        };
        auto evaluate = [&]() { // longest_match at this lane's position
            res = 1; nxt = lane + 1; mstart = 0; seen = 0; range = 0;
            if (haspos && rk != 0) {
                uint32_t m = read_bits(), best = kMinMatch - 1, nsel = 0, klast = 0;
                bool first = true, stopped = false, term = false;
                seen = m;
#pragma unroll 1
                for (int g0 = 0; g0 < CHAIN; g0 += 4) {
                    if (__builtin_amdgcn_ballot_w64(!stopped && m != 0) == 0) break;
                    uint32_t k[4], q[4]; bool v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { v[u] = m != 0; k[u] = v[u] ? (uint32_t)__builtin_ctz(m) : 0u; m &= m - 1u; }
#pragma unroll
                    for (int u = 0; u < 4; u++) q[u] = *reinterpret_cast<const uint16_t *>(stg + 62 - 2 * k[u]);
                    uint64_t cb[4][NW];
                    {
                        const uint32_t a0 = ring_a + fw_ring(q[0]), a1 = ring_a + fw_ring(q[1]), a2 = ring_a + fw_ring(q[2]), a3 = ring_a + fw_ring(q[3]);
#pragma unroll
                        for (int t = 0; t < NW; t++) fw_ld64x4(a0 + 8 * t, a1 + 8 * t, a2 + 8 * t, a3 + 8 * t, cb[0][t], cb[1][t], cb[2][t], cb[3][t]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (v[u] && !stopped) { // longest_match's bookkeeping for this candidate (deflate.c:1126-1163)
                            klast = k[u];
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
                range = stopped ? (klast >= 31 ? ~0u : (2u << klast) - 1u) : ~0u; // a search that did not stop: any of the bits matters
                if (!first && best >= kMinMatch) {
                    uint32_t l = best;
                    bool open = term; // the match reached nice_match: how long it is (the candidate's bytes against this lane's, 8 at a time)
                    for (int it = 0; it < kFwExtSteps; it++) {
                        if (__builtin_amdgcn_ballot_w64(open) == 0) break;
                        if (open) {
                            uint64_t x, y;
                            fw_ld64x2(ring_a + fw_ring(mstart + l), ring_a + fw_ring(p + l), x, y);
                            const uint32_t d = fw_diff8(x ^ y);
                            l += d;
                            if (d < 8 || l >= cap) { open = false; l = l < cap ? l : cap; }
                        }
                    }
                    res = l | (open ? kResLong : 0u);
                    nxt = lane + l;
                }
                if (!stopped && rk > kFwDepth) { res = kResInc; nxt = lane + 1; } // the chain may go on below the bits this lane has
            }
        };

        uint32_t pos = 0; bool cross_short = false; // the parse's state when it enters this window (a guess, then the true one)
        uint32_t exit_pos = 0; bool exit_cs = false;
        bool have_evals = false;
        // ---- phase 0: ahead of the parse, from a guessed entry; phase 1: this window's turn ----
#pragma unroll 1
        for (int phase = 0; phase < 2; phase++) {
            const bool commit = phase == 1;
            if (!commit) {
#ifdef ZGPU_FW_NOSPEC // (debug builds: nothing is done ahead of the parse)
                continue;
#endif
                const uint32_t turn = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_TURN));
                if (turn == win) continue; // its turn already: no guessing
                const uint32_t slot = ctl_a + 4 * (C_SPEC + (win + kFwW - 1) % kFwW);
                const uint32_t sw = __builtin_amdgcn_readfirstlane(fw_ld32(slot)), sv = __builtin_amdgcn_readfirstlane(fw_ld32(slot + 4 * kFwW));
                const uint32_t sw2 = __builtin_amdgcn_readfirstlane(fw_ld32(slot));
                if (win != 0 && sw == win - 1 && sw2 == sw) { pos = sv >> 1; cross_short = sv & 1u; }
                else { pos = w0; cross_short = false; }
                if (pos >= w0 + 64) continue; // (the guess says the window lies inside a match: nothing to prepare)
                if (pos < w0) pos = w0;
            } else {
                while (__builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_TURN)) != win) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_s_setprio(3);
                pos = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_POS)); cross_short = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_CS)) != 0;
                ntok = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_NTOK)); blk_tok0 = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_BLKTOK0));
                nblk = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_NBLK)); nostore = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_NOSTORE));
                block_start = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_BLKSTART)); off = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_OFF));
                buffered = __builtin_amdgcn_readfirstlane(fw_ld32(ctl_a + 4 * C_BUFFERED));
                FW_LAP(7);
            }
            exit_pos = pos; exit_cs = cross_short;
            if (pos >= w0 + 64) { // (at its turn only) the window lies inside a match: a long one, its strings stay out of the chains
                set_bit(cross_short);
            } else {
                const uint32_t entry = pos - w0;
                // the lanes in front of the entry are the inside of the match that reaches into the window; the others are guesses: in the chains
                // (at the turn the bits of the lanes behind the entry stand as the walk ahead of the parse left them: the walk below brings them in line)
                if (commit && have_evals) { if (lane < entry) set_bit(cross_short); } else set_bit(lane >= entry || cross_short);
                uint32_t start = entry;
                bool need_eval = !have_evals;
                uint32_t eval_from = commit ? start : 0; // (ahead of the parse all lanes are evaluated: the true entry may lie in front of the guess)
                if (entry >= lend) { // nothing to walk
                    if (need_eval && !commit) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); evaluate(); have_evals = true; FW_CNT(10, 1); }
                } else
                for (;;) { // rounds
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    FW_LAP(1);
                    FW_CNT(commit ? 9 : 13, 1);
                    if (need_eval) { FW_CNT(commit ? 14 : 10, 1); if (lane >= eval_from) evaluate(); have_evals = true; }
#ifdef ZGPU_FW_DUMP
                    if (c == 0 && win == ZGPU_FW_DUMP_WIN && commit && g_fw_dbg[512] < 4) {
                        const uint32_t r_ = g_fw_dbg[512];
                        if (r_ == 0) { g_fw_dbg[lane * 8 + 0] = iv; g_fw_dbg[lane * 8 + 1] = res; g_fw_dbg[lane * 8 + 2] = mstart; g_fw_dbg[lane * 8 + 3] = seen; g_fw_dbg[lane * 8 + 4] = range;
                                       g_fw_dbg[lane * 8 + 5] = (uint32_t)own[0]; g_fw_dbg[lane * 8 + 6] = *reinterpret_cast<const uint16_t *>(stg + 62); g_fw_dbg[lane * 8 + 7] = nxt; }
                        if (lane == 0) { g_fw_dbg[513 + r_ * 4] = start; g_fw_dbg[514 + r_ * 4] = need_eval; g_fw_dbg[515 + r_ * 4] = pos; g_fw_dbg[516 + r_ * 4] = wave; }
                        __builtin_amdgcn_s_waitcnt(0);
                        if (lane == 0) g_fw_dbg[512] = r_ + 1;
                    }
#endif
                    FW_LAP(2);
                    // ---- the walk over the token starts, from `start` ----
                    uint64_t stopm = __builtin_amdgcn_ballot_w64((res & (kResInc | kResLong)) != 0);
                    uint64_t T = 0;
                    uint32_t L = start, stop_inc = 64;
                    while (L < lend) {
                        if ((stopm >> L) & 1ull) {
                            const uint32_t x = __builtin_amdgcn_readlane(res, L);
                            if (x & kResInc) { stop_inc = L; break; }
                            // all lanes measure the match: four bytes each behind what the lane has measured
                            FW_CNT(11, 1);
                            const uint32_t p0 = w0 + L, q0 = __builtin_amdgcn_readlane(mstart, L), cap0 = n - p0 < kMaxMatch ? n - p0 : kMaxMatch, have = x & kResLen;
                            const uint32_t o = have + 4 * lane;
                            const uint32_t xa = fw_ld32(ring_a + fw_ring(q0 + o)) ^ fw_ld32(ring_a + fw_ring(p0 + o));
                            const uint64_t ne = __builtin_amdgcn_ballot_w64(xa != 0);
                            uint32_t len = cap0;
                            if (ne != 0) {
                                const uint32_t f = (uint32_t)__builtin_ctzll(ne), xf = __builtin_amdgcn_readlane(xa, f);
                                len = have + 4 * f + ((uint32_t)__builtin_ctz(xf) >> 3);
                                len = len < cap0 ? len : cap0;
                            }
                            if (lane == L) { res = len; nxt = lane + len; }
                            stopm &= ~(1ull << L);
                        }
                        T |= 1ull << L;
                        L = __builtin_amdgcn_readlane(nxt, L);
                    }
                    // the strings inside a match longer than max_insert_length (or too near the end) stay out of the chains (deflate.c:1510-1534)
                    const uint32_t mlen = res & kResLen;
                    const bool is_long = mlen > 1 && !(mlen <= max_insert && n - p - mlen >= kMinMatch);
                    const uint64_t LONG = __builtin_amdgcn_ballot_w64(is_long);
                    const uint64_t tb = T & lanes_upto; // the token this lane lies in starts at the highest set bit
                    const uint32_t ts = tb ? 63u - (uint32_t)__builtin_clzll(tb) : 64u;
                    const bool inside = ts < 64 && ts != lane && lane < L && ((LONG >> ts) & 1ull); // (the tokens tile [start, L))
                    if (lane >= start) set_bit(!inside);
                    if (L >= 64 && T) { const uint32_t tl = 63u - (uint32_t)__builtin_clzll(T); exit_cs = L > 64 ? !((LONG >> tl) & 1ull) : cross_short; }
                    FW_LAP(3);
                    // ---- which of these tokens stand ----
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    bool stale = false;
                    if (lane >= start && range != 0) stale = ((seen ^ read_bits()) & range) != 0;
                    const uint64_t st_mask = __builtin_amdgcn_ballot_w64(stale) & T;
                    const uint32_t Lstale = st_mask ? (uint32_t)__builtin_ctzll(st_mask) : 64u;
                    const uint32_t Lacc = Lstale < stop_inc ? Lstale : stop_inc; // the tokens that start below Lacc are the reference's
                    const uint64_t Tacc = Lacc >= 64 ? T : T & ((1ull << Lacc) - 1ull);
                    const uint32_t end_pos = Lacc >= lend ? w0 + L : w0 + Lacc; // where the token behind the accepted ones starts
                    FW_LAP(4);
                    if (commit) { // the accepted tokens
                        const uint32_t cnt = (uint32_t)__builtin_popcountll(Tacc);
                        if (Tacc & lane_bit) {
                            const uint32_t ln = res & kResLen;
                            tok[ntok + (uint32_t)__builtin_popcountll(Tacc & lanes_below)] = ln == 1 ? tok_lit(own_byte) : tok_match(p - mstart, ln - kMinMatch);
                        }
                        const bool near_end = w0 + 64 + kMinLookahead > buffered;
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
                    if (Lacc >= lend) { exit_pos = w0 + L; break; }
                    // the bits behind the accepted tokens are guesses again
                    if (lane >= Lacc) set_bit(true);
                    start = Lacc; eval_from = Lacc;
                    if (Lstale <= stop_inc) { need_eval = true; continue; }
                    if (!commit) { exit_pos = ~0u; break; } // ahead of the parse the bits below are not final: the whole-bucket search waits for the turn
                    FW_CNT(12, 1);
                    // ---- the search at Lacc over its whole bucket, by all lanes (every bit below it is final now) ----
                    {
                        const uint32_t p0 = w0 + Lacc, i0 = __builtin_amdgcn_readlane(idx, Lacc), r0 = __builtin_amdgcn_readlane(rk, Lacc);
                        const uint32_t look0 = n - p0, cap0 = look0 < kMaxMatch ? look0 : kMaxMatch, ni0 = (uint32_t)NICE < look0 ? (uint32_t)NICE : look0;
                        const int w00 = (int)(p0 + base), limit0 = w00 > (int)kMaxDist ? w00 - (int)kMaxDist : 0;
                        uint32_t best = kMinMatch - 1, ms0 = 0, ch = CHAIN;
                        bool first = true, done = false;
                        for (uint32_t k0 = 0; k0 < r0 && !done; k0 += 64) {
                            const uint32_t kk = k0 + lane;
                            const bool valid = kk < r0;
                            const uint32_t q = valid ? S[(int)i0 - 1 - (int)kk] : 0u;
                            const uint32_t rv = 65536u - i0 + kk;
                            const bool ins = valid && ((flags[rv >> 5] >> (rv & 31u)) & 1u);
                            uint32_t l = 0;
                            if (ins && (int)(q + base) >= limit0) { // (a candidate out of reach is turned away before its length is asked for; its bytes may have left the ring)
                                for (;;) {
                                    const uint32_t d = fw_diff8(fw_ld64(ring_a + fw_ring(q + l)) ^ fw_ld64(ring_a + fw_ring(p0 + l)));
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
                        if (lane == Lacc) { res = r1; nxt = lane + r1; mstart = ms0; seen = 0; range = 0; }
                    }
                    need_eval = false;
                    FW_LAP(6);
                }
            }
            if (!commit) { // what the wave behind may start from
                if (exit_pos != ~0u) { // (all lanes store the same words: no lane-dependent branch in front of the loops' back edges)
                    const uint32_t slot = ctl_a + 4 * (C_SPEC + win % kFwW);
                    fw_st32(slot, ~0u); fw_st32(slot + 4 * kFwW, (exit_pos << 1) | (exit_cs ? 1u : 0u)); fw_st32(slot, win);
                }
            } else {
                if (win + 1 == nwin) {
                    token_top(n); // the loop top that finds the input at its end (deflate.c:1459-1466): the slide may still happen here
                    if (off != 0 && block_start + base < kWSize) nostore |= 1u << nblk; // the final block (its emission happens in the Huffman stage)
                    meta[c].ntok = ntok; meta[c].nostore = nostore; meta[c].in_bytes = n;
                } else {
                    fw_st32(ctl_a + 4 * C_POS, exit_pos); fw_st32(ctl_a + 4 * C_CS, exit_cs ? 1u : 0u); fw_st32(ctl_a + 4 * C_NTOK, ntok); fw_st32(ctl_a + 4 * C_BLKTOK0, blk_tok0);
                    fw_st32(ctl_a + 4 * C_NBLK, nblk); fw_st32(ctl_a + 4 * C_NOSTORE, nostore); fw_st32(ctl_a + 4 * C_BLKSTART, block_start); fw_st32(ctl_a + 4 * C_OFF, off);
                    fw_st32(ctl_a + 4 * C_BUFFERED, buffered);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fw_st32(ctl_a + 4 * C_TURN, win + 1);
                __builtin_amdgcn_s_setprio(0);
                FW_LAP(15);
            }
        }
    }
    FW_END();
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
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_kernel<8, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        hipFuncSetAttribute(reinterpret_cast<const void *>(fastwin_kernel<32, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds);
        opt_in = true;
    }
    const dim3 grid(g.nchunks), block(64 * kFwW);
    if (cfg.chain == 4) hipLaunchKernelGGL((fastwin_kernel<4, 8>), grid, block, kFwLds, st, g, cfg.lazy, S, ir, tokens, meta);
    else if (cfg.chain == 8) hipLaunchKernelGGL((fastwin_kernel<8, 16>), grid, block, kFwLds, st, g, cfg.lazy, S, ir, tokens, meta);
    else hipLaunchKernelGGL((fastwin_kernel<32, 32>), grid, block, kFwLds, st, g, cfg.lazy, S, ir, tokens, meta);
#ifdef ZGPU_FW_DUMP
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(&g_fw_dump_meta, meta, sizeof g_fw_dump_meta, hipMemcpyDeviceToHost);
    (void)hipMemcpy(g_fw_dump_tok, tokens, sizeof g_fw_dump_tok, hipMemcpyDeviceToHost);
#endif
}

} // namespace zgpu
