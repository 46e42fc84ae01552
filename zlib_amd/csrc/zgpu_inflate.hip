// zgpu_inflate.hip -- decode path: one wave per chunk segment.
//
// Restates (file:line under /root/reference):
//   block header / stored / dynamic-table states of inflate()   qcsrc/inflate.c:773-949
//   code-length validation of inflate_table                     qcsrc/inftrees.c:106-138
//   symbol decode + match copy (inflate_fast and the slow path) qcsrc/inffast.c:67-302, qcsrc/inflate.c:950-1076
//   length / distance bases and extra bits                      qcsrc/inftrees.c:60-73
// The reference decodes through 2-level tables of `code` structs; only the produced bytes and the error class are
// observable, so the table layout here is the engine's own: a 10-bit (literal/length) and a 9-bit (distance) direct
// table in LDS, codes longer than that resolved by a canonical first-code walk.
//
// Work split: the bit reader and symbol decode are wave-uniform (every lane computes the same values: the decode of one
// deflate stream is sequential); lanes cooperate on table fill, input staging (1 KiB per refill, 16 bytes per lane),
// match copies (64 bytes per step) and the final store of the chunk (16 bytes per lane).  The whole output chunk
// (<= 64 KiB) lives in LDS while it is decoded, so back-references never touch global memory.
#include "zgpu_common.h"
#include <type_traits>
#include "../../include/zamd_gpu.h"
#include <cstdio>
#include <cstdlib>

struct zgpu_engine;

namespace zgpu {

int fail_hip(zgpu_engine *e, hipError_t err, const char *what, const char *file, int line);
void prof_span_begin(void *eng, hipStream_t st, hipEvent_t *a);
void prof_span_end(void *eng, hipStream_t st, int stage, hipEvent_t a);
const uint8_t *engine_inflate_dict(zgpu_engine *e);
uint32_t engine_inflate_dict_len(zgpu_engine *e);
uint32_t engine_inflate_checks(zgpu_engine *e);
void launch_adler(const ChunkGeom &g, ChunkMeta *meta, hipStream_t st);
void launch_crc(const ChunkGeom &g, ChunkMeta *meta, hipStream_t st);
void launch_scan(const ChunkMeta *meta, uint32_t nchunks, uint64_t chunk0, uint64_t *offsets, void *run, uint64_t out_cap, hipStream_t st, bool with_crc = false);
void launch_stitch(const uint8_t *slots, const ChunkMeta *meta, const uint64_t *offsets, uint64_t chunk0, uint32_t nchunks, uint8_t *out,
                   uint64_t out_cap, uint32_t slot_stride, hipStream_t st);

enum InfMsg : uint32_t {
    kMsgNone = 0, kMsgBlockType, kMsgStoredLen, kMsgTooMany, kMsgCodeLens, kMsgRepeat, kMsgLitLens, kMsgDists, kMsgLitCode, kMsgDistCode,
    kMsgTooFar, kMsgTruncated, kMsgOutput, kMsgTrailing, kMsgShort, kMsgTable, kMsgCount
};
static const char *const kInfMessages[kMsgCount] = {
    "", "invalid block type", "invalid stored block lengths", "too many length or distance symbols", "invalid code lengths set",
    "invalid bit length repeat", "invalid literal/lengths set", "invalid distances set", "invalid literal/length code", "invalid distance code",
    "invalid distance too far back", "segment ends inside a block", "segment decodes to more than chunk_size bytes",
    "segment holds data after its last block", "segment decodes to fewer than chunk_size bytes", "segment table out of range"};

// bits of the last byte that belong to a stream whose final block the last decode reached (0: all eight): inflate_stream_host's fallback for a stream taken up at a bit offset maps the end back with it
static thread_local uint32_t t_end_bits = 0;
struct InfStatus { int32_t code; uint32_t msg; uint32_t out_bytes; uint32_t used; }; // used: input bytes up to the end of the last block | final block seen << 31

constexpr uint32_t kLBits = 9, kDBits = 9, kStageDwords = 256;
#ifndef ZGPU_INF_RING
#define ZGPU_INF_RING 32768
#endif
#ifndef ZGPU_INF_PARCOPY
#define ZGPU_INF_PARCOPY 1 // the independent matches of a pass copied together (0: one after the other, A/B builds)
#endif
#ifndef ZGPU_INF_RING_DEFAULT_KB
#define ZGPU_INF_RING_DEFAULT_KB 8 // the ring of chunks decoded straight into place (zgpu_inflate_device); ZGPU_INF_RING_KB at run time
#endif
constexpr uint32_t kOutRing = ZGPU_INF_RING, kOutHalf = kOutRing / 2; // the last 32 KiB of output live in LDS (the farthest a distance reaches)

// Decoding table entries of the literal/length and distance codes carry everything the symbol loop needs:
//   bits 0-3 code length, 4-7 extra bits, 8 literal, 9 end of block, 10 length/distance, 11 invalid symbol, 16-31 byte / base value
constexpr uint32_t kEntLit = 1u << 8, kEntEob = 1u << 9, kEntLen = 1u << 10, kEntBad = 1u << 11;

// RingT: uint8_t, or uint16_t for the speculative decode of a stream's middle (spec_* below): values >= 0x8000 are markers, "the byte
// at index v & 0x7fff of the 32 KiB in front of this segment", which nobody knows yet
template <typename RingT, uint32_t kRing = kOutRing> struct InflateLdsT {
    RingT out[kRing];
    uint32_t ltab[1 << kLBits]; // 0 = code longer than kLBits (or unassigned)
    uint32_t dtab[1 << kDBits];
    uint32_t stage[kStageDwords]; // ring of input dwords
    uint32_t tok[128];            // token ring, reader -> writer, handed over in halves of 64
    uint32_t abort_flag, end_bits; // writer -> reader: stop, the output is void; reader -> writer: bits of the segment used when it ended
    uint16_t lens[320];
    uint16_t lsym[288], dsym[32]; // symbols sorted by (length, symbol) for the long-code walk
    uint16_t lcount[16], dcount[16];
    uint16_t work_offs[16], work_first[16], work_start[16];
    uint32_t build_rc, build_n;
    uint32_t end_final, pad1;     // reader -> writer: the segment ended with a final block
};
using InflateLds = InflateLdsT<uint8_t>;
using InflateLdsSpec = InflateLdsT<uint16_t>;
constexpr uint32_t kScanBytes = 4096; // the block finder reads the input through LDS in pieces of this size (+ the 16 bytes a bit offset at the end reaches into)
constexpr uint32_t kFindList = 1024; // candidates listed between two rounds of the second sieve (a group of 2048 offsets yields 683 at most: one in three)
using InflateLdsFind = InflateLdsT<uint8_t, kScanBytes + 64 + kFindList * 2>;
static_assert(sizeof(InflateLdsFind) <= 14336, "eleven finder waves per CU");
static_assert(sizeof(InflateLds) <= 40448, "four segments per CU");
static_assert(10 * sizeof(InflateLdsT<uint8_t, 8192>) <= 160 * 1024, "ten segments per CU with the 8 KiB ring (five waves per SIMD: 96 registers)");
static_assert(sizeof(InflateLdsSpec) <= 81920, "two workgroups per CU");

// Wave-uniform bit reader over a ring of input dwords in LDS.
struct BitSrc {
    const uint32_t *g32; // aligned global dwords
    uint64_t gdwords;    // dwords that may be read from g32 (bounds the whole input buffer)
    uint64_t d0;         // index of the first dword of the segment inside g32
    uint32_t filled;     // dwords of the segment staged so far
    uint32_t rd;         // dwords consumed into hold
    uint64_t hold;
    uint32_t bits;
    uint32_t nx;         // stage[rd]: read one refill ahead so that a refill never waits for LDS
    uint32_t seg_bits;   // size of the segment in bits (from its first dword, including the leading byte offset)
};

// The bit reader's state is the same in all lanes; values that come back from LDS are declared so (v_readfirstlane), which
// moves the whole decode loop -- shifts, masks, compares, branches -- from the vector pipe to scalar instructions.
#ifdef ZGPU_INF_TIME // debug build only (scripts/inf_time.py): clock per phase, summed over chunks
__device__ unsigned long long inf_time[16];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_inf_time(unsigned long long *out, int reset)
{
    unsigned long long z[16] = {};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(inf_time), sizeof z);
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(inf_time), z, sizeof z);
}
#define INF_T(i) do { const unsigned long long t_ = wall_clock64(); t_acc[i] += t_ - t_prev; t_prev = t_; } while (0)
#define INF_T0() unsigned long long t_prev = wall_clock64(); unsigned long long t_acc[16] = {}; unsigned long long n_lit = 0, n_mat = 0, n_slow = 0
#define INF_N(x) x++
#elif defined(ZGPU_INF_EXP_A)
#define INF_T(i) do { if ((ZGPU_INF_EXP_A >> (i)) & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#define INF_T0() do { } while (0)
#define INF_N(x) do { } while (0)
#elif defined(ZGPU_INF_EXP_B)
#define INF_T(i) do { if ((ZGPU_INF_EXP_B >> (i)) & 1) asm volatile("" ::: "memory"); } while (0)
#define INF_T0() do { } while (0)
#define INF_N(x) do { } while (0)
#else
#define INF_T(i) do { } while (0)
#define INF_T0() do { } while (0)
#define INF_N(x) do { } while (0)
#endif
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ inline void settle(BitSrc &b) // (the compiler cannot see that loop-carried reader state is wave-uniform: tell it once per symbol)
{
    b.hold = (uint64_t)uni((uint32_t)b.hold) | ((uint64_t)uni((uint32_t)(b.hold >> 32)) << 32);
    b.bits = uni(b.bits); b.rd = uni(b.rd); b.filled = uni(b.filled); b.seg_bits = uni(b.seg_bits);
}

__device__ inline void stage_fill(BitSrc &b, uint32_t *stage, uint32_t lane)
{
    // keep at least 120 dwords ahead of the reader; each call loads 128 dwords (8 bytes per lane).  The ring holds 256: the
    // scalar reader's rd runs two dwords ahead of the position it will be set back to (reposition, the lane-parallel decode),
    // so a fill must leave room behind rd as well: 119 + 128 ahead at most, 9 behind at least.
    while (b.filled - b.rd < 120) {
        const uint64_t i = b.d0 + b.filled + lane * 2;
        uint32_t v[2];
#pragma unroll
        for (int k = 0; k < 2; k++) v[k] = (i + k < b.gdwords) ? b.g32[i + k] : 0u;
        const uint32_t s = (b.filled + lane * 2) & (kStageDwords - 1);
#pragma unroll
        for (int k = 0; k < 2; k++) stage[s + k] = v[k];
        b.filled += 128;
    }
}
__device__ inline void refill(BitSrc &b, const uint32_t *stage)
{
    if (b.bits <= 32) { b.hold |= (uint64_t)uni(b.nx) << b.bits; b.rd++; b.bits += 32; b.nx = stage[b.rd & (kStageDwords - 1)]; } // nx stays a vector register: the wait for it belongs to its use
}
__device__ inline void prime(BitSrc &b, const uint32_t *stage) { b.nx = stage[b.rd & (kStageDwords - 1)]; } // after (re)positioning the reader
__device__ inline uint32_t peek(const BitSrc &b, uint32_t n) { return (uint32_t)b.hold & ((1u << n) - 1); }
__device__ inline void drop(BitSrc &b, uint32_t n) { b.hold >>= n; b.bits -= n; }
__device__ inline uint32_t consumed_bits(const BitSrc &b) { return b.rd * 32 - b.bits; }

// one wave's LDS operations complete in order: ordering its own writes and reads needs the compiler held back, no barrier
__device__ inline void wave_sync() { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// set bit i of a wave-uniform mask (one scalar instruction; the shift-and-or the compiler emits is two in the walk's chain)
__device__ inline void mark_bit(uint64_t &m, uint32_t i) { asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(i)); }

// v = the lane's bit of a wave mask ? a : b
__device__ inline uint32_t sel_mask(uint64_t m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
// inclusive prefix sum over the 64 lanes (DPP: shifts inside the rows of 16, then the row totals passed on)
template <int CTRL, int ROWS> __device__ inline uint32_t dpp_or_zero(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false); }
__device__ inline uint32_t wave_prefix_sum(uint32_t v)
{
    v += dpp_or_zero<0x111, 0xf>(v); // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v); // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v); // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v); // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    return v;
}

// Build one decoding table from code lengths lens[0..n).  kind: 0 code-length code, 1 literal/length, 2 distance.
// Acceptance rules of inflate_table (inftrees.c:106-138).  Returns 0 ok, 1 rejected.  Lane 0 does the serial part
// (its small work arrays live in LDS: dynamically indexed private arrays would go to scratch memory).
// base value and extra bits of length symbol 257 + k and of distance symbol s (inflate_table's lbase/lext/dbase/dext,
// inftrees.c:46-60, in closed form: no table in memory to wait for)
__device__ inline uint32_t len_extra(uint32_t k) { return (k < 8 || k == 28) ? 0u : (k >> 2) - 1; }
__device__ inline uint32_t len_base(uint32_t k) { return k < 8 ? 3 + k : k == 28 ? 258u : 3 + ((4 + (k & 3)) << ((k >> 2) - 1)); }
__device__ inline uint32_t dist_extra(uint32_t s) { return s < 4 ? 0u : (s >> 1) - 1; }
__device__ inline uint32_t dist_base(uint32_t s) { return s < 4 ? 1 + s : 1 + ((2 + (s & 1)) << ((s >> 1) - 1)); }
__constant__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// table entry of symbol s with code length l.  kind: 0 code-length code (plain sym << 8 | len), 1 literal/length, 2 distance
__device__ inline uint32_t make_entry(uint32_t kind, uint32_t s, uint32_t l)
{
    if (kind == 0) return (s << 8) | l;
    if (kind == 1) {
        if (s < 256) return l | kEntLit | (s << 16);
        if (s == 256) return l | kEntEob;
        if (s > 285) return l | kEntBad;
        return l | (len_extra(s - 257) << 4) | kEntLen | (len_base(s - 257) << 16);
    }
    if (s > 29) return l | kEntBad;
    return l | (dist_extra(s) << 4) | kEntLen | (dist_base(s) << 16);
}

template <class LDS> __device__ __noinline__ uint32_t build_table(LDS &L, const uint16_t *lens, uint32_t n, uint32_t kind, uint32_t tbits, uint32_t *tab,
                                             uint16_t *sorted, uint16_t *count, uint32_t lane)
{
    // All lanes together (round 2; one lane walking 286 lengths twice was four fifths of a dynamic block's header): a lane holds the lengths of
    // symbols lane, 64 + lane, ...; counts per length and a symbol's place among those of its length are ballots.
    wave_sync();
    for (uint32_t i = lane; i < (1u << tbits); i += 64) tab[i] = 0;
    constexpr uint32_t kGroups = 5; // 320 lengths at most
    const uint32_t ng = (n + 63) >> 6;
    uint32_t ml[kGroups];
#pragma unroll
    for (uint32_t g = 0; g < kGroups; g++) { const uint32_t s2 = g * 64 + lane; ml[g] = s2 < n ? lens[s2] : 0u; }
    uint32_t cnt[16];
    cnt[0] = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15; l++) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t g = 0; g < kGroups; g++) if (g < ng) c += (uint32_t)__builtin_popcountll(__ballot(ml[g] == l));
        cnt[l] = c;
    }
    uint32_t maxl = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15; l++) maxl = cnt[l] ? l : maxl;
    uint32_t rc = 0;
    if (maxl > 0) { // inflate_table's rules, inftrees.c:106-138
        int left = 1;
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) { left <<= 1; left -= (int)cnt[l]; if (left < 0) rc = 1; }
        if (!rc && left > 0 && (kind == 0 || maxl != 1)) rc = 1; // incomplete set
    }
    uint32_t first[16], start[16], c = 0, o = 0;
    first[0] = 0; start[0] = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15; l++) { c = (c + cnt[l - 1]) << 1; first[l] = c; start[l] = o; o += cnt[l]; }
    // the per-length rows where the long-code walk and the fill below look for them
    {
        uint32_t mc = 0, mf = 0, ms = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) { mc = lane == l ? cnt[l] : mc; mf = lane == l ? first[l] : mf; ms = lane == l ? start[l] : ms; }
        if (lane < 16) { count[lane] = (uint16_t)mc; L.work_first[lane] = (uint16_t)mf; L.work_start[lane] = (uint16_t)ms; }
        if (lane == 0) { L.build_n = o; L.build_rc = rc; }
    }
    if (rc == 0) {
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) {
            if (cnt[l] == 0) continue;
            uint32_t base = start[l];
#pragma unroll
            for (uint32_t g = 0; g < kGroups; g++) {
                if (g >= ng) continue;
                const uint64_t m = __ballot(ml[g] == l);
                if (ml[g] == l) sorted[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(g * 64 + lane);
                base += (uint32_t)__builtin_popcountll(m);
            }
        }
    }
    wave_sync();
    if (rc == 0) {
        // symbol number j in (length, symbol) order has the canonical code first[l] + (j - start[l])
        for (uint32_t j = lane; j < o; j += 64) {
            const uint32_t s2 = sorted[j], l = lens[s2];
            if (l <= tbits) {
                const uint32_t code = (uint32_t)L.work_first[l] + (j - L.work_start[l]), rev = __brev(code) >> (32 - l), e = make_entry(kind, s2, l);
                for (uint32_t i = rev; i < (1u << tbits); i += 1u << l) tab[i] = e;
            }
        }
    }
    wave_sync();
    return rc;
}

// A code longer than the table, canonical first-code method with one code length per lane: lane l (1..15) holds, for its
// table, the first code of length l, the number of codes of that length and where they start in the (length, symbol) order
// (CodeRows, loaded after build_table); the pattern decodes at the one length whose code range holds its first l bits.
// Returns symbol | length << 16, or 0xFFFF when the bit pattern is not assigned (incomplete / empty code).
struct CodeRows { uint32_t first, count, start; };
template <class LDS> __device__ inline CodeRows load_rows(const LDS &L, const uint16_t *count, uint32_t lane)
{
    CodeRows r;
    r.first = L.work_first[lane & 15]; r.start = L.work_start[lane & 15]; r.count = (lane >= 1 && lane < 16) ? count[lane] : 0u;
    return r;
}
__device__ inline uint32_t long_code(uint32_t hbits, const CodeRows &r, const uint16_t *sorted, uint32_t lane)
{
    const uint32_t l = (lane & 15) ? (lane & 15) : 1, d = (__brev(hbits) >> (32 - l)) - r.first;
    const bool hit = d < r.count; // (count is zero in the lanes that hold no length)
    uint32_t sym = 0;
    if (hit) sym = sorted[r.start + d];
    const uint64_t m = __ballot(hit);
    if (!m) return 0xFFFFu;
    const uint32_t at = (uint32_t)__builtin_ctzll(m);
    return (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)at) | (at << 16);
}

// decode one symbol of the code-length code (plain entries sym << 8 | len; its codes all fit the 7-bit table); 0xFFFF when
// the bit pattern is not assigned
__device__ inline uint32_t decode_sym(BitSrc &b, const uint32_t *tab, uint32_t tbits)
{
    const uint32_t e = uni(tab[peek(b, tbits)]);
    if (!e) return 0xFFFFu;
    drop(b, e & 255);
    return e >> 8;
}

// The header of a dynamic block behind its three type bits (inflate.c:811-880): the counts, the code-length code, the code lengths, the
// two decoding tables.  Returns 0 or the message of the first rule broken.  Wave-uniform; shared by the reader and the block finder.
// QUICK (the block finder, which only wants yes or no): the lengths' sums are kept while they are read, and a literal/length or distance code that is
// over-subscribed already ends the parse -- a header that is none usually is within a few dozen lengths, not after three hundred.  (The decoder proper
// reads them all first: an invalid repeat further on is the error zlib reports, inflate.c:838-866 before :870-885.)
template <bool QUICK, class LDS> __device__ inline uint32_t dynamic_header(LDS &L, BitSrc &b, uint32_t lane, CodeRows &lrows, CodeRows &drows)
{
    refill(b, L.stage);
    const uint32_t nlen = peek(b, 5) + 257; drop(b, 5);
    const uint32_t ndist = peek(b, 5) + 1; drop(b, 5);
    const uint32_t ncode = peek(b, 4) + 4; drop(b, 4);
    if (nlen > 286 || ndist > 30) return kMsgTooMany;
    // The code-length code (19 symbols, codes of at most 7 bits) is built in registers: lane s holds the length of symbol s, the canonical codes come
    // from ballots, and the 128-entry decoding table lives in two registers per lane (entry `lane` and entry `64 + lane`: a look-up is a
    // v_readlane, not a round trip to LDS).  inflate_table's rules for this code (inftrees.c:106-138): over-subscribed or incomplete is an error.
    uint64_t y;
    {
        refill(b, L.stage);
        const uint32_t n0 = ncode < 10 ? ncode : 10;
        const uint64_t lo = peek(b, 3 * n0); drop(b, 3 * n0);
        refill(b, L.stage);
        const uint32_t n1 = ncode - n0;
        const uint64_t hi = n1 ? peek(b, 3 * n1) : 0u; drop(b, 3 * n1);
        y = lo | (hi << 30);
    }
    // where symbol s stands in the order the lengths are sent in (16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15): five bits each
    constexpr uint64_t kInvLo = 3ull | (17ull << 5) | (15ull << 10) | (13ull << 15) | (11ull << 20) | (9ull << 25) | (7ull << 30) | (5ull << 35) | (4ull << 40) | (6ull << 45) | (8ull << 50) | (10ull << 55);
    constexpr uint64_t kInvHi = 12ull | (14ull << 5) | (16ull << 10) | (18ull << 15) | (0ull << 20) | (1ull << 25) | (2ull << 30);
    const uint32_t where = lane < 12 ? (uint32_t)(kInvLo >> (5 * lane)) & 31u : lane < 19 ? (uint32_t)(kInvHi >> (5 * (lane - 12))) & 31u : 31u;
    const uint32_t cl_len = where < ncode ? (uint32_t)(y >> (3 * where)) & 7u : 0u;
    uint32_t cl_first[8], cl_rank = 0, kraft = 0, code = 0, prev_count = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 7; l++) {
        const uint64_t m = __ballot(cl_len == l);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m), below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        code = (code + prev_count) << 1; cl_first[l] = code; prev_count = cnt;
        kraft += cnt * (128u >> l);
        if (cl_len == l) cl_rank = cl_first[l] + below;
    }
    if (kraft != 128u && kraft != 0u) return kMsgCodeLens; // (no code at all: every look-up below fails, as the reference's empty table does)
    const uint32_t cl_rev = cl_len ? __brev(cl_rank) >> (32 - cl_len) : 0u;
    uint32_t cl0 = 0, cl1 = 0;
    for (uint32_t sy = 0; sy < 19; sy++) {
        const uint32_t ls = (uint32_t)__builtin_amdgcn_readlane((int)cl_len, (int)sy);
        if (!ls) continue;
        const uint32_t rs = (uint32_t)__builtin_amdgcn_readlane((int)cl_rev, (int)sy), mk = (1u << ls) - 1, e = (sy << 8) | ls;
        if ((lane & mk) == rs) cl0 = e;
        if (((lane + 64) & mk) == rs) cl1 = e;
    }
    wave_sync();
    for (uint32_t s = lane; s < 320; s += 64) L.lens[s] = 0;
    uint32_t have = 0, prev = 0, qkl = 0, qkd = 0;
    while (have < nlen + ndist) {
        stage_fill(b, L.stage, lane);
        refill(b, L.stage);
        const uint32_t ci = peek(b, 7);
        const uint32_t ce = ci < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)cl0, (int)ci) : (uint32_t)__builtin_amdgcn_readlane((int)cl1, (int)(ci - 64));
        if (!ce) return kMsgCodeLens;
        drop(b, ce & 255u);
        const uint32_t s = ce >> 8;
        if (s < 16) {
            if (lane == 0) L.lens[have] = (uint16_t)s;
            if (QUICK && s) { if (have < nlen) qkl += 32768u >> s; else qkd += 32768u >> s; if (qkl > 32768u || qkd > 32768u) return kMsgLitLens; }
            prev = s; have++; continue;
        }
        uint32_t rep, val = 0;
        refill(b, L.stage);
        if (s == 16) { if (have == 0) return kMsgRepeat; val = prev; rep = 3 + peek(b, 2); drop(b, 2); }
        else if (s == 17) { rep = 3 + peek(b, 3); drop(b, 3); }
        else { rep = 11 + peek(b, 7); drop(b, 7); }
        if (have + rep > nlen + ndist) return kMsgRepeat;
        if (QUICK && val) { // (a run of equal lengths may straddle the two codes)
            const uint32_t inl = have >= nlen ? 0u : (have + rep <= nlen ? rep : nlen - have);
            qkl += inl * (32768u >> val); qkd += (rep - inl) * (32768u >> val);
            if (qkl > 32768u || qkd > 32768u) return kMsgLitLens;
        }
        if (lane < rep) L.lens[have + lane] = (uint16_t)val;
        if (lane + 64 < rep) L.lens[have + lane + 64] = (uint16_t)val;
        if (lane + 128 < rep) L.lens[have + lane + 128] = (uint16_t)val;
        prev = val; have += rep;
    }
    wave_sync();
    // inflate_table's verdict on the two sets of lengths (inftrees.c:106-138: over-subscribed, or incomplete with more than a single one-bit
    // code), taken by all lanes together before lane 0 builds anything: the block finder comes here with thousands of headers that are none
    {
        uint32_t kl = 0, kd = 0, ml = 0, md = 0;
        for (uint32_t i = lane; i < nlen + ndist; i += 64) {
            const uint32_t l = L.lens[i], k = l ? (32768u >> l) : 0u;
            if (i < nlen) { kl += k; ml = l > ml ? l : ml; } else { kd += k; md = l > md ? l : md; }
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            kl += (uint32_t)__shfl_xor((int)kl, sh); kd += (uint32_t)__shfl_xor((int)kd, sh);
            const uint32_t a = (uint32_t)__shfl_xor((int)ml, sh), c = (uint32_t)__shfl_xor((int)md, sh);
            ml = a > ml ? a : ml; md = c > md ? c : md;
        }
        kl = uni(kl); kd = uni(kd); ml = uni(ml); md = uni(md);
        if (ml && (kl > 32768u || (kl < 32768u && ml != 1))) return kMsgLitLens;
        if (md && (kd > 32768u || (kd < 32768u && md != 1))) return kMsgDists;
    }
    if (build_table(L, L.lens, nlen, 1, kLBits, L.ltab, L.lsym, L.lcount, lane)) return kMsgLitLens;
    lrows = load_rows(L, L.lcount, lane);
    wave_sync();
    if (build_table(L, L.lens + nlen, ndist, 2, kDBits, L.dtab, L.dsym, L.dcount, lane)) return kMsgDists;
    drows = load_rows(L, L.dcount, lane);
    return kMsgNone;
}

// One workgroup of two waves per segment.  Wave 0 (the reader) owns the bit stream: block headers, code tables and the
// token decode; it never needs to know how many bytes came out so far.  Wave 1 (the writer) owns the output: the 32 KiB
// ring in LDS, match copies, flushes to the destination, and the limits that depend on the output position (distance too
// far back, chunk size).  Tokens travel through a ring of 128 words in LDS, handed over in halves of 64 with one workgroup
// barrier per half: while the writer executes half k the reader fills half k + 1.
// token word: bits 0-1 kind (0 nothing, 1 literal, 2 match, 3 command)
//   literal: bits 2-9 byte;  match: bits 2-10 length, 11-25 distance - 1
//   command: bits 2-3 which (1 stored bytes: bits 4-20 count, the next two words = offset of the bytes in the input;
//            2 end of the segment: bits 4-11 the reader's verdict)
enum : uint32_t { kCmdStored = 1, kCmdEnd = 2 };
constexpr uint32_t kWholeStream = 0xFFFFFFFFu; // chunk_size argument: the one segment is a whole stream of any size

__device__ inline void block_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

// SPEC (speculative decode of one stream in pieces, see spec_* below): segment gc starts at BIT offsets[gc] of the input -- a block start
// the finder believes in -- and ends at the first block boundary at or behind bit offsets[gc + 1], or with the final block; nothing is known about
// the 32 KiB in front of it, so the ring holds 16-bit symbols (byte, or marker = index into that unknown window) and goes to pages of
// kOutHalf symbols taken from a pool as it fills, the segment's place in the output being unknown too.
struct SpecEnd { uint64_t end_bit; uint32_t out_bytes, flags; }; // flags: bit 0 the segment ended with the final block, bit 1 the page pool ran dry
struct SpecArgs {
    uint16_t *mid;        // pages of kOutHalf symbols
    uint64_t *page_owner; // per page: segment << 32 | index of the page within the segment
    uint32_t *page_count; // pages taken so far
    uint32_t page_cap;
    uint16_t *tails;      // per segment: the ring when it ended, oldest symbol first = the last 32 KiB of the segment's output
    SpecEnd *ends;
};
// RING: bytes of output the workgroup keeps in LDS.  32768 is the farthest a distance reaches: every match copies from the ring.  A smaller ring
// (chunks placed directly at their offset of the destination only) lets more segments share a CU; a match that reaches farther back than RING reads its
// source from the destination itself, where every byte older than the ring has been flushed: the lanes that hold such matches ask for their first 32 bytes
// when their half of the token ring is handed over, all at once, and the copy takes them from registers when its turn comes.
template <bool SPEC, uint32_t RING = ZGPU_INF_RING>
__global__ void __launch_bounds__(128, (!SPEC && RING <= 8192) ? 5 : 4) inflate_kernel_t(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint64_t *__restrict__ offsets,
                                                        uint64_t chunk0, uint32_t nchunks, uint64_t last_chunk, uint32_t chunk_size_arg,
                                                        uint8_t *__restrict__ out, uint64_t out_cap, InfStatus *status, ChunkMeta *meta,
                                                        const uint8_t *__restrict__ dict, uint32_t dict_len, uint32_t stream_mode, SpecArgs sp)
{
    typedef typename std::conditional<SPEC, uint16_t, uint8_t>::type ring_t;
    typedef InflateLdsT<ring_t, RING> Lds;
    constexpr uint32_t kOutRing = RING, kOutHalf = RING / 2; // (this kernel's ring, not the file's default)
    constexpr bool FAR = RING < 32768;
    static_assert(!(SPEC && FAR) && RING >= 4096 && (RING & (RING - 1)) == 0, "ring size");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    Lds &L = *reinterpret_cast<Lds *>(lds_raw);
    const uint32_t c = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t role = uni(threadIdx.x >> 6); // 0 reader, 1 writer
    if (c >= nchunks) return;
    const uint64_t gc = chunk0 + c;
    uint64_t seg_lo = offsets[gc], seg_hi = offsets[gc + 1];
    uint32_t bit_lead = 0, stop_bits = 0xFFFFFFFFu; // SPEC: bits of the first byte in front of the start; where the next segment starts, in bits from seg_lo
    bool bad_table;
    bool at_len = false; // SPEC: the piece starts at the LEN field of a stored block (its header bits lie in front, the piece before checks them)
    if (SPEC) {
        // bit 62 of a start: the position is the (byte-aligned) LEN of a stored block; the piece in front of such a start stops at the first block
        // boundary that can be that block's: 3 header bits and up to 7 of padding in front of LEN
        at_len = (seg_lo >> 62) & 1u;
        const uint64_t s0 = seg_lo & ~(3ull << 62), s1 = ((seg_hi >> 62) & 1u) ? (seg_hi & ~(3ull << 62)) - 10 : seg_hi;
        bad_table = s0 >= s1 || s1 > in_bytes * 8 || s1 - (s0 & ~7ull) >= (1ull << 31);
        seg_lo = s0 >> 3; seg_hi = in_bytes - seg_lo < (1ull << 28) ? in_bytes : seg_lo + (1ull << 28);
        bit_lead = (uint32_t)(s0 & 7u); stop_bits = bad_table ? 0u : (uint32_t)(s1 - seg_lo * 8);
    } else {
        // the table may arrive next to the data from anywhere: an entry that does not lie inside the input, runs backwards or is longer than
        // a 32-bit bit count can express is an error of that segment, decoded as an empty one (nothing outside the input is ever read)
        bad_table = seg_lo > seg_hi || seg_hi > in_bytes || seg_hi - seg_lo >= (1ull << 29);
    }
    if (bad_table) { seg_lo = 0; seg_hi = 0; }
    const bool must_be_final = !SPEC && gc == last_chunk;
    // chunk_size_arg == 0: "compact" mode, segments of any size up to 64 KiB are decoded into per-chunk slots and
    // concatenated afterwards (used for streams whose chunks are not all full, e.g. flushed mid-chunk)
    const bool compact = chunk_size_arg == 0;
    // chunk_size_arg == kWholeStream: one segment of any size (a stream that was not produced in chunks): decoded from end to
    // end by this one workgroup straight into the destination, limited only by the destination's capacity (a destination
    // that is too small still gets the size that would have been needed)
    const bool whole = SPEC || chunk_size_arg == kWholeStream;
    const uint32_t chunk_size = compact ? kChunkMax : whole ? 0xFFFF0000u : chunk_size_arg;
    // a preset dictionary (inflateSetDictionary, inflate.c:1200-1236) is what the window holds before the first byte: in the ring it
    // sits right below position 0, and the first segment may reach that much farther back
    const uint32_t reach = (gc == 0) ? dict_len : SPEC ? kOutRing : 0u;
    if (threadIdx.x == 0) { L.abort_flag = 0; L.end_bits = 0; L.end_final = 0; }
    INF_T0();

    if (role == 0) {
        // =========================================== reader ===========================================
        uint32_t err = bad_table ? (uint32_t)kMsgTable : (uint32_t)kMsgNone;
        uint32_t wr = 0;   // tokens put into the ring so far
        bool stop = false; // the writer gave up (its error comes first in stream order)
        auto publish = [&]() { // the current half is complete: hand it over, the other half is free from here on
            INF_T(9);
            block_sync();
            INF_T(13);
            stop = uni(L.abort_flag) != 0;
        };
        // the marked lanes' token words, in lane order
        auto emit_tokens = [&](uint64_t marks, uint32_t word) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(marks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)marks, 0u));
            const uint32_t t = (uint32_t)__builtin_popcountll(marks), bound = (wr | 63u) + 1, at = wr + rank;
            const bool mine = sel_mask(marks, 1u, 0u) != 0;
            if (mine && at < bound) L.tok[at & 127u] = word;
            if (wr + t >= bound) {
                publish();
                if (mine && at >= bound) L.tok[at & 127u] = word;
            }
            wr += t;
        };
        // n <= 3 words that must sit in one half (lane i holds word i); `final`: pad the half and hand it over
        auto emit_command = [&](uint32_t word, uint32_t n, bool final) {
            if ((wr & 63u) + n > 64u) { // no room: pad this half with nothing-tokens
                const uint32_t bound = (wr | 63u) + 1;
                if (wr + lane < bound) L.tok[(wr + lane) & 127u] = 0u;
                wr = bound;
                publish();
            }
            if (lane < n) L.tok[(wr + lane) & 127u] = word;
            wr += n;
            if (final) {
                const uint32_t bound = ((wr - 1) | 63u) + 1;
                if (wr + lane < bound) L.tok[(wr + lane) & 127u] = 0u;
                wr = bound;
                publish();
            } else if ((wr & 63u) == 0) publish();
        };
        BitSrc b;
        const uint64_t in_addr = reinterpret_cast<uint64_t>(in);
        const uint64_t abs_lo = in_addr + seg_lo, abs_al = abs_lo & ~3ull;
        b.g32 = reinterpret_cast<const uint32_t *>(abs_al);
        b.gdwords = (in_addr + in_bytes - abs_al + 3) >> 2; // reads past the input buffer are replaced by zeros, see stage_fill
        // the dword holding the last input bytes may extend past the buffer by up to 3 bytes inside the same aligned dword
        b.d0 = 0; b.filled = 0; b.rd = 0; b.hold = 0; b.bits = 0;
        const uint32_t lead = (uint32_t)(abs_lo - abs_al);
        b.seg_bits = (uint32_t)(seg_hi - seg_lo + lead) * 8;
        stage_fill(b, L.stage, lane);
        wave_sync();
        prime(b, L.stage);
        refill(b, L.stage); refill(b, L.stage);
        drop(b, lead * 8);
        if (SPEC) drop(b, bit_lead);
        auto reposition = [&](uint32_t pos) { // the scalar reader at bit `pos` of the staged dwords
            b.rd = pos >> 5; b.hold = 0; b.bits = 0;
            prime(b, L.stage); refill(b, L.stage); refill(b, L.stage);
            drop(b, pos & 31);
        };
        bool last = false, seen_final = false;
        uint32_t org_bits = 0; // bits between the segment's (aligned) start and the bit reader's origin: stored blocks move the origin
        CodeRows lrows{}, drows{}; // per-length rows of the two codes of the current block (lanes 1..15)
        INF_T(0);
        while (!err && !last && !stop) {
            if (consumed_bits(b) >= b.seg_bits) break; // segment exhausted at a block boundary (normal end of a non-final segment)
            if (SPEC && org_bits + consumed_bits(b) - lead * 8 >= stop_bits) break; // a block boundary at or behind the next segment's start
            stage_fill(b, L.stage, lane); wave_sync();
            refill(b, L.stage);
            uint32_t type = 0;
            if (SPEC && at_len) at_len = false; // (the first block of this piece: stored, not the last, the reader stands at its LEN)
            else {
                const uint32_t hdr = peek(b, 3); drop(b, 3);
                last = hdr & 1; seen_final = seen_final || last;
                type = hdr >> 1;
            }
            if (type == 3) { err = kMsgBlockType; break; }
            if (type == 0) {
                drop(b, b.bits & 7);
                refill(b, L.stage);
                const uint32_t len = peek(b, 16); drop(b, 16);
                refill(b, L.stage);
                const uint32_t nlen = peek(b, 16); drop(b, 16);
                if (len != (nlen ^ 0xFFFFu)) { err = kMsgStoredLen; break; }
                const uint32_t bytepos = consumed_bits(b) >> 3; // from the reader's current origin (dword b.d0 of the segment)
                if ((uint64_t)bytepos + len > (b.seg_bits >> 3)) { err = kMsgTruncated; break; }
                const uint64_t src = reinterpret_cast<uint64_t>(b.g32 + b.d0) + bytepos - in_addr; // the writer copies the bytes from the input
                emit_command(lane == 0 ? (3u | (kCmdStored << 2) | (len << 4)) : lane == 1 ? (uint32_t)src : (uint32_t)(src >> 32), 3, false);
                // reposition the reader right after the stored bytes: new origin = the dword holding that byte
                const uint32_t np = bytepos + len;
                b.d0 += np >> 2; b.filled = 0; b.rd = 0; b.hold = 0; b.bits = 0;
                b.seg_bits -= (np & ~3u) * 8; // seg_bits stays relative to the origin
                org_bits += (np & ~3u) * 8;   // ... and this is where the origin stands in the segment
                wave_sync();
                stage_fill(b, L.stage, lane); wave_sync();
                prime(b, L.stage);
                refill(b, L.stage); refill(b, L.stage);
                drop(b, (np & 3) * 8);
                continue;
            }
            // ---- tables ----
            if (type == 1) {
                for (uint32_t s = lane; s < 288; s += 64) L.lens[s] = (uint16_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                wave_sync();
                build_table(L, L.lens, 288, 1, kLBits, L.ltab, L.lsym, L.lcount, lane);
                lrows = load_rows(L, L.lcount, lane);
                wave_sync();
                for (uint32_t s = lane; s < 32; s += 64) L.lens[s] = 5;
                wave_sync();
                build_table(L, L.lens, 32, 2, kDBits, L.dtab, L.dsym, L.dcount, lane);
                drows = load_rows(L, L.dcount, lane);
            } else {
                err = dynamic_header<false>(L, b, lane, lrows, drows);
                if (err) break;
            }
            wave_sync();
            INF_T(1);
            // ---- symbols ----
            // Every lane decodes the token that would start at its own bit offset behind the reader (both table lookups, extra
            // bits included); a scalar walk from offset 0 then picks the lanes that really are token starts (each token says
            // where the next one begins).  A token the tables do not resolve (code longer than the table, end of block, invalid
            // code, the end of the segment) ends the walk and goes through the one-symbol path below.
            uint32_t pos = consumed_bits(b); // the reader's position in bits from the segment's origin dword
            bool eob = false;
            while (!eob && !stop) {
                pos = uni(pos); wr = uni(wr);
                b.rd = pos >> 5; b.filled = uni(b.filled); b.seg_bits = uni(b.seg_bits);
                stage_fill(b, L.stage, lane);
                if (pos > b.seg_bits) { err = kMsgTruncated; break; }
                // two windows of 64 bit offsets per step: lane i looks at offsets i and 64 + i
                uint32_t info0, word0, info1, word1; // info: token length in bits (64 = not a token the walk may take)
                {
                    const uint32_t p = pos + lane, wi = p >> 5, sh = p & 31;
                    uint32_t w[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) w[k] = L.stage[(wi + k) & (kStageDwords - 1)];
                    // (both windows side by side, so that their table lookups are in flight together)
                    const uint32_t lo[2] = {__builtin_amdgcn_alignbit(w[1], w[0], sh), __builtin_amdgcn_alignbit(w[3], w[2], sh)};
                    const uint32_t hi[2] = {__builtin_amdgcn_alignbit(w[2], w[1], sh), __builtin_amdgcn_alignbit(w[4], w[3], sh)};
                    uint32_t e[2], ed[2], h2[2], t[2], len[2];
#pragma unroll
                    for (int k = 0; k < 2; k++) e[k] = L.ltab[lo[k] & ((1u << kLBits) - 1)];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t l1 = e[k] & 15u, xl = (e[k] >> 4) & 15u;
                        t[k] = l1 + xl;
                        len[k] = (e[k] >> 16) + __builtin_amdgcn_ubfe(lo[k] >> l1, 0, xl);
                        h2[k] = (uint32_t)((((uint64_t)hi[k] << 32) | lo[k]) >> t[k]);
                    }
#pragma unroll
                    for (int k = 0; k < 2; k++) ed[k] = L.dtab[h2[k] & ((1u << kDBits) - 1)];
                    uint32_t info[2], word[2];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t l2 = ed[k] & 15u, xd = (ed[k] >> 4) & 15u;
                        const uint32_t dist = (ed[k] >> 16) + __builtin_amdgcn_ubfe(h2[k] >> l2, 0, xd);
                        // (selects by mask arithmetic: written as ?: the compiler turns the two token kinds into branches on exec)
                        const uint32_t litm = 0u - ((e[k] >> 8) & 1u), l1 = e[k] & 15u;
                        const uint32_t tokm = litm | (0u - ((e[k] & ed[k] & kEntLen) >> 10));
                        const uint32_t nb = l1 + ((t[k] - l1 + l2 + xd) & ~litm);
                        const uint32_t as_lit = 1u | ((e[k] >> 16) << 2), as_match = 2u | (len[k] << 2) | ((dist - 1) << 11);
                        word[k] = (as_lit & litm) | (as_match & ~litm);
                        info[k] = (tokm != 0 && p + 64 * k + nb <= b.seg_bits) ? nb : 64u;
                    }
                    info0 = info[0]; word0 = word[0]; info1 = info[1]; word1 = word[1];
                }
                INF_T(8);
                // the walk: token starts from offset 0 on (a lane that is not a token is marked too and ends it)
                uint64_t marks0 = 0, marks1 = 0;
                uint32_t cur = 0;
                do { mark_bit(marks0, cur); cur += (uint32_t)__builtin_amdgcn_readlane((int)info0, (int)cur); } while (cur < 64);
                const uint64_t bad0 = __ballot(info0 == 64u) & marks0;
                if (bad0) { cur = (uint32_t)__builtin_ctzll(bad0); marks0 &= (1ull << cur) - 1; }
                else {
                    uint32_t c1 = cur - 64;
                    do { mark_bit(marks1, c1); c1 += (uint32_t)__builtin_amdgcn_readlane((int)info1, (int)c1); } while (c1 < 64);
                    const uint64_t bad1 = __ballot(info1 == 64u) & marks1;
                    if (bad1) { c1 = (uint32_t)__builtin_ctzll(bad1); marks1 &= (1ull << c1) - 1; }
                    cur = 64 + c1;
                }
                if (cur != 0) {
                    emit_tokens(marks0, word0);
                    if (marks1) emit_tokens(marks1, word1);
                    INF_N(n_lit); // (steps)
                    INF_T(9);
                    pos += cur;
                    continue;
                }
                // ---- one symbol through the scalar reader ----
                INF_N(n_slow);
                reposition(pos);
                uint32_t e = uni(L.ltab[(uint32_t)b.hold & ((1u << kLBits) - 1)]);
                if (e) drop(b, e & 15u);
                else { // a code longer than the table, or no code at all
                    const uint32_t s2 = long_code((uint32_t)b.hold, lrows, L.lsym, lane);
                    if (s2 == 0xFFFFu) { err = kMsgLitCode; break; }
                    drop(b, s2 >> 16);
                    e = make_entry(1, s2 & 0xFFFFu, 0);
                }
                uint32_t tokw;
                if (e & kEntLit) tokw = 1u | ((e >> 16) << 2);
                else {
                    if (e & kEntEob) { eob = true; pos = consumed_bits(b); if (pos > b.seg_bits) err = kMsgTruncated; break; }
                    if (e & kEntBad) { err = kMsgLitCode; break; }
                    const uint32_t xl = (e >> 4) & 15u, len = (e >> 16) + peek(b, xl);
                    drop(b, xl);
                    refill(b, L.stage);
                    uint32_t ed = uni(L.dtab[(uint32_t)b.hold & ((1u << kDBits) - 1)]);
                    if (ed) drop(b, ed & 15u);
                    else {
                        const uint32_t d2 = long_code((uint32_t)b.hold, drows, L.dsym, lane);
                        if (d2 == 0xFFFFu) { err = kMsgDistCode; break; }
                        drop(b, d2 >> 16);
                        ed = make_entry(2, d2 & 0xFFFFu, 0);
                    }
                    if (ed & kEntBad) { err = kMsgDistCode; break; }
                    const uint32_t xd = (ed >> 4) & 15u, dist = (ed >> 16) + peek(b, xd);
                    drop(b, xd);
                    tokw = 2u | (len << 2) | ((dist - 1) << 11);
                }
                pos = consumed_bits(b);
                if (pos > b.seg_bits) { err = kMsgTruncated; break; } // the token runs past the end of the segment
                emit_tokens(1ull, tokw);
                INF_T(12);
            }
            // the scalar reader takes over again at the block boundary
            if (eob && !err) reposition(pos);
            else if (err == kMsgTruncated && !eob) reposition(pos);
        }
        // an error found in bits that lie past the end of the segment is the zero padding talking: the segment is truncated
        if (err && consumed_bits(b) > b.seg_bits) err = kMsgTruncated;
        if (!err && !stop) {
            const uint32_t used = consumed_bits(b);
            if (used > b.seg_bits) err = kMsgTruncated;                       // decoded past the end of the segment
            else if (SPEC) { if (!seen_final && org_bits + used - lead * 8 < stop_bits) err = kMsgTruncated; } // the input ended first
            else if (must_be_final && !seen_final) err = kMsgTruncated;       // the stream never ends
            else if (stream_mode && seen_final) { }                           // stream mode: the stream ends where its final block ends, whatever follows
            else if (!must_be_final && seen_final) err = kMsgTrailing;        // a final block before the last segment
            else if (((b.seg_bits - used) >> 3) != 0) err = kMsgTrailing;     // whole bytes left over
            if (!err && lane == 0) { L.end_bits = org_bits + used - lead * 8; L.end_final = seen_final ? 1u : 0u; } // counted from the segment's first byte
        }
#ifdef ZGPU_INF_DEBUG2
        if (err && lane == 0) printf("chunk %u reader err %u consumed %u seg_bits %u rd %u filled %u bits %u\n", c, err, consumed_bits(b), b.seg_bits, b.rd, b.filled, b.bits);
#endif
        emit_command(3u | (kCmdEnd << 2) | (err << 4), 1, true);
#ifdef ZGPU_INF_TIME
        if (lane == 0) { t_acc[5] = n_lit; t_acc[7] = n_slow; for (int i_ = 0; i_ < 16; i_++) if (t_acc[i_]) atomicAdd(&inf_time[i_], t_acc[i_]); }
#endif
        return;
    }

    // =========================================== writer ===========================================
    uint32_t err = kMsgNone;
    uint32_t o = 0;       // bytes produced
    if (SPEC) {
        for (uint32_t i = lane; i < kOutRing; i += 64) L.out[i] = (ring_t)(0x8000u | i); // slot i, never written, is byte i of the window in front
        if (gc == 0) for (uint32_t i = lane; i < dict_len; i += 64) L.out[(kOutRing - dict_len + i) & (kOutRing - 1)] = dict[i];
    } else
    for (uint32_t i = lane; i < reach; i += 64) L.out[(kOutRing - reach + i) & (kOutRing - 1)] = dict[i];
    uint32_t flushed = 0; // bytes already copied from the LDS ring to the destination (a multiple of kOutHalf until the end)
    bool nofit = false;   // direct placement: the destination ended before the chunk did
    uint8_t *dst = compact ? out + (uint64_t)c * kChunkMax : whole ? out : out + gc * (uint64_t)chunk_size;
    const uint64_t dst_room = SPEC ? ~0ull : compact ? kChunkMax : whole ? out_cap : (out_cap > gc * (uint64_t)chunk_size ? out_cap - gc * (uint64_t)chunk_size : 0);
    // copy a match of `len` bytes at distance `dist` to output position `at`; a distance shorter than the length repeats its
    // pattern (byte-sequential semantics of inffast.c:246-259).  The ring holds the last 32 KiB: a read at the full distance
    // 32768 hits the slot its own lane is about to write.
    auto copy_match = [&](uint32_t at, uint32_t len, uint32_t dist) {
        __builtin_amdgcn_wave_barrier();
        if (len <= 64 && dist >= len) { // the common case
            if (lane < len) { const ring_t v = L.out[(at - dist + lane) & (kOutRing - 1)]; L.out[(at + lane) & (kOutRing - 1)] = v; }
        } else if (dist >= len || dist >= 64) {
            for (uint32_t i0 = 0; i0 < len; i0 += (dist < 64 ? dist : 64)) {
                const uint32_t span = dist < 64 ? dist : 64, i = i0 + lane;
                ring_t v = 0;
                if (lane < span && i < len) v = L.out[(at - dist + i) & (kOutRing - 1)];
                if (lane < span && i < len) L.out[(at + i) & (kOutRing - 1)] = v;
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            const uint32_t recip = 0xFFFFFFFFu / dist + 1;
            for (uint32_t i = lane; i < len; i += 64) {
                const uint32_t qd = dist == 1 ? i : __umulhi(i, recip), r = i - qd * dist; // recip overflows for dist 1
                L.out[(at + i) & (kOutRing - 1)] = L.out[(at - dist + r) & (kOutRing - 1)];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // copy bytes [flushed, upto) of the output to the destination; the range never wraps in the ring
    auto flush_to = [&](uint32_t upto) {
        INF_T(11);
        wave_sync();
        uint32_t nbytes = upto - flushed;
        if (SPEC) { // the next page of the pool (flushed is a multiple of kOutHalf: the page's index within the segment)
            uint32_t page = 0;
            if (lane == 0) page = atomicAdd(sp.page_count, 1u);
            page = uni(page);
            if (page >= sp.page_cap) nofit = true;
            else {
                if (lane == 0) sp.page_owner[page] = (gc << 32) | (flushed / kOutHalf);
                const uint4 *s128 = reinterpret_cast<const uint4 *>(L.out + (flushed & (kOutRing - 1)));
                uint4 *d128 = reinterpret_cast<uint4 *>(sp.mid + (uint64_t)page * kOutHalf);
                for (uint32_t i = lane; i < (nbytes + 7) / 8; i += 64) d128[i] = s128[i]; // (whole vectors: the page and the ring half have the room)
            }
            flushed = upto;
            return;
        }
        if ((uint64_t)flushed + nbytes > dst_room) { nofit = true; nbytes = dst_room > flushed ? (uint32_t)(dst_room - flushed) : 0; }
        const uint8_t *src_r = reinterpret_cast<const uint8_t *>(L.out) + (flushed & (kOutRing - 1)); // (the byte ring; SPEC has returned above)
        uint8_t *d = dst + flushed;
        if ((reinterpret_cast<uintptr_t>(d) & 15) == 0) {
            const uint4 *s128 = reinterpret_cast<const uint4 *>(src_r);
            for (uint32_t i = lane; i < (nbytes >> 4); i += 64) reinterpret_cast<uint4 *>(d)[i] = s128[i];
            for (uint32_t i = (nbytes & ~15u) + lane; i < nbytes; i += 64) d[i] = src_r[i];
        } else {
            for (uint32_t i = lane; i < nbytes; i += 64) d[i] = src_r[i];
        }
        flushed = upto;
        if (FAR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // what has been flushed IS in memory: matches read it back from there
        INF_T(3);
    };
    uint32_t reader_err = kMsgNone;
    INF_T(0);
    for (uint32_t half = 0;; half++) {
        block_sync(); // the reader has completed this half of the ring
        INF_T(14);
        const uint32_t tw = L.tok[(half & 1) * 64 + lane];
        const uint32_t kind = tw & 3u;
        uint64_t cmds = __ballot(kind == 3);
        uint32_t from = 0; // lanes below are done
        bool ended = false;
        for (;;) { // runs of tokens between the commands of this half; after an error only the commands are followed
            const uint32_t upto = cmds ? (uint32_t)__builtin_ctzll(cmds) : 64u;
            if (!err) { // ---- the tokens in lanes [from, upto) ----
                o = uni(o); flushed = uni(flushed);
                const bool inr = lane >= from && lane < upto;
                const uint32_t k2 = inr ? kind : 0u, len = (tw >> 2) & 511u, ol = k2 == 1 ? 1u : k2 == 2 ? len : 0u;
                const uint32_t incl = wave_prefix_sum(ol), offv = o + incl - ol, o_end = o + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                // limits that depend on the output position, in the order inffast.c checks them for one token
                const bool far = k2 == 2 && (tw >> 11) >= offv + reach, over = offv + ol > chunk_size;
                const uint64_t bad = __ballot(far || over);
                if (bad) err = ((__ballot(far) >> (uint32_t)__builtin_ctzll(bad)) & 1) ? kMsgTooFar : kMsgOutput;
                uint64_t todo = bad ? 0ull : __ballot(ol != 0);
                // FAR: matches whose source is no longer in the ring (the ring holds the RING bytes in front of a token when its turn comes)
                uint64_t farm = 0, pfm = 0;
                uint32_t p0, p1, p2, p3, p4, p5, p6, p7; // (written by the loads below when they arrive, read after the wait in front of their first use: nothing else may touch them in between)
                if (FAR) {
                    const bool isfar = k2 == 2 && (tw >> 11) >= kOutRing;
                    farm = todo & __ballot(isfar);
                    if (farm && dst_room >= 32) {
                        // the source ends below flushed - RING / 2 + 516: flushed, and in memory (flush_to waits for its stores).  32 bytes at once when
                        // the match is that short and the 32 bytes lie inside what is flushed (and inside the destination: one that is too small gets no
                        // reads past its end).  Every lane loads -- the others the destination's first bytes -- so that the registers have one writer;
                        // sc0 sc1: from memory, not from a line this CU's cache took in when only a part of it had been flushed.
                        const uint32_t s0 = offv - (tw >> 11) - 1;
                        const bool rdy = isfar && len <= 32 && s0 + 32 <= flushed && (uint64_t)s0 + 32 <= dst_room;
                        pfm = farm & __ballot(rdy);
                        const uint8_t *ps = dst + (rdy ? s0 : 0u);
                        if (pfm) // (uniform; every such run waits for these loads when the first of its lanes' matches is copied)
                        asm volatile("global_load_dword %0, %8, off sc0 sc1\n\tglobal_load_dword %1, %8, off offset:4 sc0 sc1\n\t"
                                     "global_load_dword %2, %8, off offset:8 sc0 sc1\n\tglobal_load_dword %3, %8, off offset:12 sc0 sc1\n\t"
                                     "global_load_dword %4, %8, off offset:16 sc0 sc1\n\tglobal_load_dword %5, %8, off offset:20 sc0 sc1\n\t"
                                     "global_load_dword %6, %8, off offset:24 sc0 sc1\n\tglobal_load_dword %7, %8, off offset:28 sc0 sc1"
                                     : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7) : "v"(ps) : "memory");
                    }
                }
                // Literals are stored by their lanes ahead of the match copies of the same pass.  The ring is exactly as long as
                // the farthest distance, so a store that far ahead may hit what an earlier match still has to read: position
                // q lands on the slot of q - 32768.  A pass therefore ends with the first match whose source would be reached
                // by the pass's last byte (and with the half of the ring that is flushed next).
                while (todo) {
                    const uint32_t lim = flushed + kOutHalf;
                    uint64_t take = todo & __ballot(offv < lim), rest = todo & ~take;
                    uint32_t pend = rest ? (uint32_t)__builtin_amdgcn_readlane((int)offv, (int)__builtin_ctzll(rest)) : o_end;
                    const uint64_t hz = take & ~farm & __ballot(k2 == 2 && offv + (kOutRing - 1) - (tw >> 11) < pend);
                    if (hz) {
                        take &= (2ull << (uint32_t)__builtin_ctzll(hz)) - 1; rest = todo & ~take;
                        pend = rest ? (uint32_t)__builtin_amdgcn_readlane((int)offv, (int)__builtin_ctzll(rest)) : o_end;
                    }
                    if (sel_mask(take, k2, 0u) == 1) L.out[offv & (kOutRing - 1)] = (ring_t)(uint8_t)(tw >> 2);
                    INF_T(10);
                    uint64_t mm = take & __ballot(k2 == 2);
                    if (ZGPU_INF_PARCOPY) {
                        // The matches whose source ends in front of what this pass produces do not depend on each other or on the pass's literals: they are copied together, 64 bytes of
                        // their concatenation a step -- byte b belongs to the first match whose running length exceeds b (a search over the lanes' prefix sums by shuffles) --
                        // instead of one match after the other (most are under ten bytes: six lanes of 64 at work, 215 ns each at ten segments a CU).  The others -- a source inside the
                        // pass's own output, a distance shorter than the length, a source that is read back from the destination -- follow in order as before.
                        const uint32_t dist = (tw >> 11) + 1;
                        const bool ind = ((take >> lane) & 1ull) && k2 == 2 && !(FAR && ((farm >> lane) & 1ull)) && dist >= (offv - o) + len;
                        const uint64_t pm = __ballot(ind);
                        if (__builtin_popcountll(pm) >= 2) {
                            const uint32_t li = ind ? len : 0u, incl = wave_prefix_sum(li), T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            __builtin_amdgcn_wave_barrier();
                            for (uint32_t it = 0; it < T; it += 64) {
                                const uint32_t b = it + lane;
                                uint32_t lo = 0, hi = 63;
#pragma unroll
                                for (int st = 0; st < 6; st++) { const uint32_t mid = (lo + hi) >> 1, v = (uint32_t)__shfl((int)incl, (int)mid); if (v > b) hi = mid; else lo = mid + 1; }
                                const uint32_t im = (uint32_t)__shfl((int)incl, (int)lo), lm = (uint32_t)__shfl((int)li, (int)lo), om = (uint32_t)__shfl((int)offv, (int)lo), dm = (uint32_t)__shfl((int)dist, (int)lo);
                                const uint32_t j = b - (im - lm);
                                if (b < T) { const ring_t v = L.out[(om - dm + j) & (kOutRing - 1)]; L.out[(om + j) & (kOutRing - 1)] = v; }
                            }
                            __builtin_amdgcn_wave_barrier();
                            mm &= ~pm;
                        }
                    }
                    while (mm) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(mm); mm &= mm - 1;
                        const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)tw, (int)l), mo = (uint32_t)__builtin_amdgcn_readlane((int)offv, (int)l);
                        if (FAR && ((farm >> l) & 1ull)) {
                            const uint32_t flen = (t2 >> 2) & 511u, fs = mo - (t2 >> 11) - 1;
                            __builtin_amdgcn_wave_barrier();
                            if ((pfm >> l) & 1ull) { // the 32 bytes lane l asked for: byte i to lane i
                                // (the wait and the reads of the loaded registers in ONE statement: the compiler must not read -- copy -- them before the loads have landed)
                                uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
                                asm volatile("s_waitcnt vmcnt(0)\n\t"
                                             "v_readlane_b32 %0, %8, %16\n\tv_readlane_b32 %1, %9, %16\n\tv_readlane_b32 %2, %10, %16\n\tv_readlane_b32 %3, %11, %16\n\t"
                                             "v_readlane_b32 %4, %12, %16\n\tv_readlane_b32 %5, %13, %16\n\tv_readlane_b32 %6, %14, %16\n\tv_readlane_b32 %7, %15, %16"
                                             : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7)
                                             : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "s"(l) : "memory");
                                const uint32_t ws = (lane >> 2) & 7u;
                                const uint32_t wv = ws == 0 ? d0 : ws == 1 ? d1 : ws == 2 ? d2 : ws == 3 ? d3 : ws == 4 ? d4 : ws == 5 ? d5 : ws == 6 ? d6 : d7;
                                if (lane < flen) L.out[(mo + lane) & (kOutRing - 1)] = (ring_t)(uint8_t)(wv >> (8 * (lane & 3u)));
                            } else { // longer than 32 bytes, or asked for too early: from memory now
                                for (uint32_t i = lane; i < flen; i += 64) {
                                    uint8_t v = 0;
                                    if ((uint64_t)fs + i < dst_room) v = __hip_atomic_load(dst + fs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    L.out[(mo + i) & (kOutRing - 1)] = (ring_t)v;
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                        } else
                        copy_match(mo, (t2 >> 2) & 511u, (t2 >> 11) + 1);
                        INF_N(n_mat);
                    }
                    todo = rest; o = pend;
                    if (o >= lim) flush_to(lim);
                    INF_T(11);
                }
            }
            if (upto == 64) break;
            // ---- the command in lane upto ----
            const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)tw, (int)upto);
            if (((cw >> 2) & 3u) == kCmdEnd) { reader_err = (cw >> 4) & 255u; ended = true; break; }
            if (!err) {
                const uint64_t soff = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)tw, (int)((upto + 2) & 63u)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)tw, (int)((upto + 1) & 63u));
                const uint8_t *src = in + soff;
                const uint32_t slen = (cw >> 4) & 0x1FFFFu;
                if (o + slen > chunk_size) err = kMsgOutput;
                else for (uint32_t done = 0; done < slen;) { // through the ring, half by half
                    const uint32_t room = flushed + kOutHalf - o, part = slen - done < room ? slen - done : room;
                    for (uint32_t i = lane; i < part; i += 64) L.out[(o + i) & (kOutRing - 1)] = src[done + i];
                    o += part; done += part;
                    if (o == flushed + kOutHalf) flush_to(o);
                }
            }
            from = upto + 3;
            cmds &= ~(7ull << upto);
        }
        if (ended) break;
        if (err && lane == 0) L.abort_flag = 1; // void output: the reader stops at its next hand-over and sends its end command
    }
    if (!err) err = reader_err;
    if (!err && !compact && !SPEC && !must_be_final && o != chunk_size) err = kMsgShort; // direct placement assumes full chunks
#ifdef ZGPU_INF_DEBUG2
    if (err && lane == 0) printf("chunk %u err %u o %u\n", c, err, o);
#endif
    // the rest of the chunk (an error leaves what was flushed before it was found; the status says the chunk is void)
    if (!err && (!SPEC || o != flushed)) flush_to(o);
    if (SPEC && !err) // the ring is the last 32 KiB of what this segment knows: slots it never wrote still name the window in front
        for (uint32_t j = lane; j < kOutRing; j += 64) sp.tails[gc * kOutRing + j] = L.out[(o + j) & (kOutRing - 1)];
    const bool fits = !nofit;
    INF_T(4);
    if (lane == 0) {
        status[c].code = err ? ZGPU_DATA_ERROR : (fits ? ZGPU_OK : ZGPU_BUF_ERROR);
        status[c].msg = err ? err : ((L.end_bits & 7u) << 24); status[c].out_bytes = err ? 0 : o; // (a segment that decoded: the bits of its last byte that are its own, for a stream taken up at a bit offset)
        status[c].used = err ? 0u : (((L.end_bits + 7u) >> 3) | (L.end_final << 31));
        if (SPEC) { sp.ends[gc].end_bit = seg_lo * 8 + L.end_bits; sp.ends[gc].out_bytes = err ? 0 : o; sp.ends[gc].flags = err ? (err << 8) : (L.end_final | (nofit ? 2u : 0u)); }
#ifdef ZGPU_INF_TIME
        t_acc[6] = n_mat;
        for (int i_ = 0; i_ < 16; i_++) if (t_acc[i_]) atomicAdd(&inf_time[i_], t_acc[i_]);
#endif
        if (meta) { meta[c].out_bytes = err ? 0 : o; meta[c].ntok = 0; meta[c].adler_a = 1; meta[c].adler_b = 0; meta[c].in_bytes = 0; meta[c].data_type = 2; }
    }
}

// first failing chunk + total bytes (one workgroup; chunk order matters for "first")
// stream_mode: the segments are candidate pieces of ONE stream that ends where its first final block ends.  Segments behind that
// one are not part of it (whatever they decoded to is dropped: their meta.out_bytes is cleared for the scan and the stitcher);
// and when the only failure is the last segment stopping inside a block, the segments before it stand as a partial result.
__global__ void __launch_bounds__(1024) inflate_reduce_kernel(const InfStatus *st, uint32_t nchunks, uint64_t chunk0, uint32_t chunk_size, uint64_t *acc,
                                                              uint32_t stream_mode, uint64_t last_chunk, ChunkMeta *meta, uint32_t trunc_msg)
{
    // acc[0] total bytes, acc[1] first bad chunk (+1, 0 = none), acc[2] its code, acc[3] its msg, acc[4] "short chunk before the end" flag,
    // acc[5] chunk that ends the stream (+1), acc[6] input bytes of that chunk used, acc[7] chunk at which a partial result stops (+1)
    __shared__ unsigned long long bad_min, fin_min;
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) { bad_min = ~0ull; fin_min = ~0ull; total = 0; }
    __syncthreads();
    if (acc[5] != 0 || acc[7] != 0) { // an earlier batch ended the stream (or the partial result): nothing of this batch counts
        if (meta) for (uint32_t i = threadIdx.x; i < nchunks; i += 1024) meta[i].out_bytes = 0;
        return;
    }
    unsigned long long bad = ~0ull, fin = ~0ull;
    for (uint32_t i = threadIdx.x; i < nchunks; i += 1024) {
        if (st[i].code != 0 && bad == ~0ull) bad = chunk0 + i;
        if (stream_mode && st[i].code == 0 && (st[i].used >> 31) && fin == ~0ull) fin = chunk0 + i;
    }
    atomicMin(&bad_min, bad);
    atomicMin(&fin_min, fin);
    __syncthreads();
    unsigned long long upto = chunk0 + nchunks; // chunks [chunk0, upto) count
    if (stream_mode) {
        if (fin_min != ~0ull && fin_min < bad_min) upto = fin_min + 1;                       // the stream ends in chunk fin_min
        else if (bad_min != ~0ull && bad_min == last_chunk && st[bad_min - chunk0].msg == trunc_msg) upto = bad_min; // incomplete tail
    }
    unsigned long long t = 0;
    for (uint32_t i = threadIdx.x; i < nchunks; i += 1024) {
        if (chunk0 + i < upto) t += st[i].out_bytes;
        else if (meta) meta[i].out_bytes = 0;
    }
    atomicAdd(&total, t);
    __syncthreads();
    if (threadIdx.x == 0) {
        acc[0] += total;
        if (stream_mode && upto != chunk0 + nchunks) {
            if (fin_min != ~0ull && fin_min < bad_min) { acc[5] = fin_min + 1; acc[6] = (st[fin_min - chunk0].used & 0x7fffffffu) | ((unsigned long long)((st[fin_min - chunk0].msg >> 24) & 7u) << 56); }
            else acc[7] = bad_min + 1;
        } else if (acc[1] == 0 && bad_min != ~0ull) { acc[1] = bad_min + 1; acc[2] = (uint64_t)(int64_t)st[bad_min - chunk0].code; acc[3] = st[bad_min - chunk0].msg; }
        else if (stream_mode && fin_min != ~0ull) { acc[5] = fin_min + 1; acc[6] = (st[fin_min - chunk0].used & 0x7fffffffu) | ((unsigned long long)((st[fin_min - chunk0].msg >> 24) & 7u) << 56); } // (the final block ends the last chunk)
    }
}

int inflate_run(zgpu_engine *e, const uint8_t *d_in, uint64_t in_bytes, const uint64_t *d_offsets, uint64_t nchunks, uint32_t chunk_size,
                uint8_t *d_out, uint64_t out_cap, zgpu_inflate_result *res, hipStream_t st, uint32_t stream_mode, const uint64_t *h_offsets,
                bool open_end = false, uint8_t *h_dst = nullptr, uint64_t h_cap = 0);

} // namespace zgpu

using namespace zgpu;

// engine internals needed here (defined in zgpu_engine.hip)
namespace zgpu {
void *engine_scratch(zgpu_engine *e, size_t bytes);          // grow-only device scratch
void *engine_scratch2(zgpu_engine *e, size_t bytes);
void *engine_run_state(zgpu_engine *e);
ChunkMeta *engine_meta(zgpu_engine *e, uint32_t batch);
uint64_t *engine_offsets_scratch(zgpu_engine *e, uint64_t n);
int engine_device(zgpu_engine *e);
void engine_collect(zgpu_engine *e);
int engine_fail(zgpu_engine *e, int code, const char *msg);
hipStream_t engine_copy_stream(zgpu_engine *e);
hipEvent_t engine_copy_event(zgpu_engine *e, size_t i);
struct RunStateHostI { uint64_t out_total, in_total, ntokens; uint32_t adler_a, adler_b, data_type, overflow, crc, pad; };

// Adler-32 and CRC-32 of the produced bytes (same kernels as the compress side), over 64 KiB pieces of the output
static int output_checksums(zgpu_engine *e, const uint8_t *d_out, uint64_t nbytes, uint64_t out_cap, zgpu_inflate_result *res, hipStream_t st)
{
    const uint32_t checks = engine_inflate_checks(e); // bit 0: Adler-32, bit 1: CRC-32 (zgpu_inflate_set_checks)
    if (!checks) { engine_collect(e); res->adler32 = 1; res->crc32 = 0; return ZGPU_OK; }
    const uint64_t max_pieces = (out_cap >> 16) + 2;
    const uint32_t cbatch_cap = (uint32_t)(max_pieces < 65536 ? max_pieces : 65536);
    const uint64_t npieces = nbytes ? (nbytes + kChunkMax - 1) / kChunkMax : 1;
    const uint32_t cbatch = (uint32_t)(npieces < cbatch_cap ? npieces : cbatch_cap);
    ChunkMeta *meta = engine_meta(e, cbatch);
    uint64_t *oscr = engine_offsets_scratch(e, npieces + 2);
    if (!meta || !oscr) return engine_fail(e, ZGPU_MEM_ERROR, "checksum scratch");
    RunStateHostI rs{}; rs.adler_a = 1;
    ZGPU_HIP_CHECK(hipMemcpyAsync(engine_run_state(e), &rs, sizeof rs, hipMemcpyHostToDevice, st));
    for (uint64_t c0 = 0; c0 < npieces; c0 += cbatch) {
        const uint32_t nb = (uint32_t)(npieces - c0 < cbatch ? npieces - c0 : cbatch);
        ChunkGeom g{}; g.in = d_out; g.in_bytes = nbytes; g.chunk_size = kChunkMax; g.chunk0 = c0; g.nchunks = nb; g.final_chunk = ~0ull;
        ZGPU_HIP_CHECK(hipMemsetAsync(meta, 0, (size_t)nb * sizeof(ChunkMeta), st));
        if (checks & 1u) launch_adler(g, meta, st); // (not computed: the pieces read a = 0, b = 0; the result is not reported)
        if (checks & 2u) launch_crc(g, meta, st);
        launch_scan(meta, nb, c0, oscr, engine_run_state(e), ~0ull, st, (checks & 2u) != 0);
    }
    ZGPU_HIP_CHECK(hipMemcpyAsync(&rs, engine_run_state(e), sizeof rs, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    engine_collect(e);
    res->adler32 = (checks & 1u) ? rs.adler_a | (rs.adler_b << 16) : 1u;
    res->crc32 = (checks & 2u) ? rs.crc : 0u;
    return ZGPU_OK;
}

// stream_mode (compact or whole-stream calls): see inflate_reduce_kernel; h_offsets = the offsets table on the host (for res->in_used);
// open_end: the last segment ends with a flush marker like the others, no segment has to hold the final block
// h_dst (direct placement only): the caller's host buffer of h_cap bytes; every batch's output is copied there on the engine's copy stream while
// the next batch is being decoded.  out_cap is the DEVICE buffer's room (whole chunks); no copy to the host reaches beyond h_cap, and a result that
// does not fit h_cap is an error before anything that was not served batch by batch is copied
int inflate_run(zgpu_engine *e, const uint8_t *d_in, uint64_t in_bytes, const uint64_t *d_offsets, uint64_t nchunks, uint32_t chunk_size,
                uint8_t *d_out, uint64_t out_cap, zgpu_inflate_result *res, hipStream_t st, uint32_t stream_mode, const uint64_t *h_offsets, bool open_end,
                uint8_t *h_dst, uint64_t h_cap)
{
    const uint64_t host_cap = h_dst ? (h_cap < out_cap ? h_cap : out_cap) : 0;
    const uint64_t last_chunk = open_end ? ~0ull : nchunks - 1;
    if (!e || !res || !d_in || !d_out || !d_offsets || nchunks == 0 || (chunk_size > kChunkMax && !(chunk_size == kWholeStream && nchunks == 1)) ||
        (chunk_size == kWholeStream && in_bytes >= (1ull << 29)))
        return engine_fail(e, ZGPU_STREAM_ERROR, "bad inflate arguments");
    ZGPU_HIP_CHECK(hipSetDevice(engine_device(e)));
    const bool compact = chunk_size == 0; // segments of any size: decode into slots, then concatenate
    // (one batch has nothing to overlap with.  Until round 3 these copies were bounded by the device buffer's size, not by the caller's: a buffer of
    // exactly the decoded length was overrun by up to a chunk -- which is what "threw inside the runtime" on small calls)
    static long tohost_min = -1;
    if (tohost_min < 0) { const char *v = getenv("ZGPU_TOHOST_MIN_CHUNKS"); tohost_min = v ? atol(v) : 4097; }
    const bool to_host = h_dst && chunk_size != 0 && chunk_size <= kChunkMax && (long)nchunks >= tohost_min;
    const uint32_t batch_cap = to_host ? 4096u : 65536u; // (host output: batches small enough for copies and kernels to take turns)
    const uint32_t batch = (uint32_t)(nchunks < batch_cap ? nchunks : batch_cap);
    InfStatus *status = static_cast<InfStatus *>(engine_scratch(e, (size_t)batch * sizeof(InfStatus) + 64));
    if (!status) return engine_fail(e, ZGPU_MEM_ERROR, "inflate scratch");
    uint64_t *acc = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(status) + (((size_t)batch * sizeof(InfStatus) + 15) & ~(size_t)15));
    const uint64_t max_pieces = (out_cap >> 16) + 2;
    const uint32_t cbatch_cap = (uint32_t)(max_pieces < 65536 ? max_pieces : 65536); // the checksum pass works on 64 KiB pieces of the OUTPUT
    ChunkMeta *meta = engine_meta(e, batch > cbatch_cap ? batch : cbatch_cap);
    if (!meta) return engine_fail(e, ZGPU_MEM_ERROR, "inflate meta");
    uint8_t *slots = nullptr;
    if (compact) { slots = static_cast<uint8_t *>(engine_scratch2(e, (size_t)batch * kChunkMax + 256)); if (!slots) return engine_fail(e, ZGPU_MEM_ERROR, "inflate slots"); }
    res->adler32 = 1; res->crc32 = 0; res->first_bad_chunk = -1; res->error_code = 0; res->error_msg = 0; res->out_bytes = 0;
    res->in_used = in_bytes; res->in_used_bits = 0; res->stream_end = 0; res->incomplete = 0;
    ZGPU_HIP_CHECK(hipMemsetAsync(acc, 0, 8 * sizeof(uint64_t), st));
    RunStateHostI rs{}; rs.adler_a = 1;
    ZGPU_HIP_CHECK(hipMemcpyAsync(engine_run_state(e), &rs, sizeof rs, hipMemcpyHostToDevice, st));
    // the ring: 32 KiB (every distance inside it), or -- chunks that go straight to their place in the destination, no dictionary in front -- a smaller
    // one with the far matches read back from the destination (more segments per CU).  ZGPU_INF_RING_KB=8|16|32 picks it.
    int ring_kb = ZGPU_INF_RING_DEFAULT_KB; // (read at every call: tests/test_gpu_inflate.py runs the same streams through all three)
    if (const char *v = getenv("ZGPU_INF_RING_KB")) ring_kb = atoi(v);
    if (ring_kb != 8 && ring_kb != 16) ring_kb = 32;
    const int ring_here = (!compact && chunk_size != kWholeStream && engine_inflate_dict_len(e) == 0) ? ring_kb : 32;
    static bool opt_in = false;
    if (!opt_in) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InflateLds));
        hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel_t<false, 16384>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InflateLdsT<uint8_t, 16384>));
        hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel_t<false, 8192>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InflateLdsT<uint8_t, 8192>));
        opt_in = true;
    }
    hipEvent_t ev{};
    int rc_sum = 0;
    prof_span_begin(e, st, &ev);
    uint64_t *oscr = engine_offsets_scratch(e, nchunks + 1 + (out_cap >> 16) + 2);
    if (!oscr) return engine_fail(e, ZGPU_MEM_ERROR, "inflate offsets");
    for (uint64_t c0 = 0; c0 < nchunks; c0 += batch) {
        const uint32_t nb = (uint32_t)(nchunks - c0 < batch ? nchunks - c0 : batch);
        if (ring_here == 8)
            hipLaunchKernelGGL((inflate_kernel_t<false, 8192>), dim3(nb), dim3(128), sizeof(InflateLdsT<uint8_t, 8192>), st, d_in, in_bytes, d_offsets, c0, nb, last_chunk, chunk_size,
                               d_out, out_cap, status, nullptr, engine_inflate_dict(e), 0u, stream_mode, SpecArgs{});
        else if (ring_here == 16)
            hipLaunchKernelGGL((inflate_kernel_t<false, 16384>), dim3(nb), dim3(128), sizeof(InflateLdsT<uint8_t, 16384>), st, d_in, in_bytes, d_offsets, c0, nb, last_chunk, chunk_size,
                               d_out, out_cap, status, nullptr, engine_inflate_dict(e), 0u, stream_mode, SpecArgs{});
        else
        hipLaunchKernelGGL(inflate_kernel_t<false>, dim3(nb), dim3(128), sizeof(InflateLds), st, d_in, in_bytes, d_offsets, c0, nb, last_chunk, chunk_size,
                           compact ? slots : d_out, out_cap, status, compact ? meta : nullptr, engine_inflate_dict(e), engine_inflate_dict_len(e), stream_mode, SpecArgs{});
        hipLaunchKernelGGL(inflate_reduce_kernel, dim3(1), dim3(1024), 0, st, status, nb, c0, chunk_size, acc, stream_mode, last_chunk, compact ? meta : nullptr, (uint32_t)kMsgTruncated);
        if (compact) {
            launch_scan(meta, nb, c0, oscr, engine_run_state(e), out_cap, st); // out_bytes -> byte offsets, continuing across batches
            launch_stitch(slots, meta, oscr, c0, nb, d_out, out_cap, kChunkMax, st);
        }
        ZGPU_HIP_CHECK(hipGetLastError());
        if (to_host) { // the batch before this one goes to the host while this one is being decoded (a copy to pageable memory holds the host)
            const size_t b = (size_t)(c0 / batch);
            ZGPU_HIP_CHECK(hipEventRecord(engine_copy_event(e, b), st));
            if (b > 0) {
                const uint64_t lo = (c0 - batch) * chunk_size, hi = c0 * (uint64_t)chunk_size < host_cap ? c0 * (uint64_t)chunk_size : host_cap;
                ZGPU_HIP_CHECK(hipStreamWaitEvent(engine_copy_stream(e), engine_copy_event(e, b - 1), 0));
                if (hi > lo) ZGPU_HIP_CHECK(hipMemcpyAsync(h_dst + lo, d_out + lo, hi - lo, hipMemcpyDeviceToHost, engine_copy_stream(e)));
            }
        }
    }
    if (to_host) {
        const size_t b = (size_t)((nchunks - 1) / batch);
        const uint64_t lo = b * (uint64_t)batch * chunk_size, hi = nchunks * (uint64_t)chunk_size < host_cap ? nchunks * (uint64_t)chunk_size : host_cap;
        ZGPU_HIP_CHECK(hipStreamWaitEvent(engine_copy_stream(e), engine_copy_event(e, b), 0));
        if (hi > lo) ZGPU_HIP_CHECK(hipMemcpyAsync(h_dst + lo, d_out + lo, hi - lo, hipMemcpyDeviceToHost, engine_copy_stream(e)));
        ZGPU_HIP_CHECK(hipStreamSynchronize(engine_copy_stream(e)));
    }
    prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev);
    uint64_t h[8];
    ZGPU_HIP_CHECK(hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    if (stream_mode) {
        if (h[5]) { res->stream_end = 1; res->in_used = (h_offsets ? h_offsets[h[5] - 1] : 0) + (h[6] & 0xffffffffffffffull); t_end_bits = (uint32_t)(h[6] >> 56) & 7u; }
        else if (h[7]) { res->incomplete = 1; res->in_used = h_offsets ? h_offsets[h[7] - 1] : 0; }
    }
    res->out_bytes = h[0]; res->first_bad_chunk = h[1] ? (int32_t)(h[1] - 1) : -1; res->error_code = (int32_t)(int64_t)h[2]; res->error_msg = (uint32_t)h[3];
    if (h[1]) { engine_collect(e); return engine_fail(e, res->error_code, kInfMessages[res->error_msg < kMsgCount ? res->error_msg : 0]); }
    if (h[0] > out_cap || (h_dst && h[0] > host_cap)) { engine_collect(e); return engine_fail(e, ZGPU_BUF_ERROR, "output capacity too small"); }
    rc_sum = output_checksums(e, d_out, h[0], out_cap, res, st);
    if (rc_sum) return rc_sum;
    if (h_dst && !to_host && h[0]) { // a host destination that was not served batch by batch
        ZGPU_HIP_CHECK(hipMemcpyAsync(h_dst, d_out, h[0], hipMemcpyDeviceToHost, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    }
    return ZGPU_OK;
}
} // namespace zgpu

namespace zgpu {
// ---- chunk boundaries of a stream that carries no side table: every full-flush marker 00 00 FF FF ends a segment ----
__global__ void __launch_bounds__(256) marker_scan_kernel(const uint8_t *__restrict__ in, uint64_t n, uint64_t *cand, uint32_t cap, uint32_t *count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 4 <= n; i += stride) {
        if (in[i] == 0 && in[i + 1] == 0 && in[i + 2] == 0xFF && in[i + 3] == 0xFF) {
            const uint32_t k = atomicAdd(count, 1u);
            if (k < cap) cand[k] = i + 4;
        }
    }
}
uint8_t *engine_stage_in(zgpu_engine *e);
uint8_t *engine_stage_out(zgpu_engine *e);
int engine_ensure_stage(zgpu_engine *e, uint64_t in_bytes, uint64_t out_bytes);
hipStream_t engine_stream(zgpu_engine *e);
} // namespace zgpu

#include <algorithm>
#include <atomic>
#include <vector>
static std::atomic<uint64_t> g_spec_done{0}, g_whole_done{0};
namespace zgpu {
// ======================================================================================================================================
// A stream that was not produced in chunks (any other zlib's output), decoded in pieces all the same.
//   1. spec_find_kernel: behind every `spacing` bytes of the input, the first bit offset that reads as the header of a dynamic block the
//      decoder would accept (type bits, counts in range, a complete code-length code -- checked by every lane for its own offset --, then
//      the code lengths and the two codes through the decoder's own dynamic_header()).  Such a header at a wrong offset is possible
//      but rare; step 3 finds out.
//   2. inflate_kernel_t<true>: one workgroup per piece, from its start to the first block boundary at or behind the next piece's start,
//      into 16-bit symbols: a byte, or a marker for "byte j of the 32 KiB in front of this piece".
//   3. the host checks the chain: every piece must have ended exactly where the next one started, the last with the final block.  Anything
//      else (a false start, damaged data, input that stops early) and the stream goes to the one-workgroup decoder, whose verdicts stand.
//   4. spec_window_kernel: piece by piece, the last 32 KiB of output with the markers replaced (the only serial step: 32 K look-ups each);
//      spec_resolve_kernel: every page of symbols to its place in the output, markers looked up in the window of the piece in front.
// ======================================================================================================================================
// The block finder's third sieve, one candidate per lane: do the code lengths behind the header at bit `cb` (its three type bits included) describe
// a literal/length and a distance code inflate_table would accept (inftrees.c:106-138), with a code for the end of the block?  Everything a lane
// needs is its own: the bits come from global memory, the code-length code is decoded canonically (counts per length, symbols in
// (length, symbol) order in 19 bytes of LDS), the lengths are summed as they are read.  A yes is confirmed by the decoder's own parse.
__device__ inline bool lane_header_ok(const uint32_t *__restrict__ g32, uint64_t gdwords, uint64_t cb, uint64_t total_bits, uint8_t *sorted)
{
    auto bits33 = [&](uint64_t p) -> uint64_t { // the 33 bits (at least) at absolute bit p
        const uint64_t wi = p >> 5;
        const uint32_t w0 = wi < gdwords ? g32[wi] : 0u, w1 = wi + 1 < gdwords ? g32[wi + 1] : 0u;
        return ((((uint64_t)w1) << 32) | w0) >> (p & 31u);
    };
    uint64_t p = cb + 3;
    if (p + 14 + 57 > total_bits) return false;
    const uint32_t hdr = (uint32_t)bits33(p) & 0x3fffu; p += 14;
    const uint32_t nlen = (hdr & 31u) + 257, ndist = ((hdr >> 5) & 31u) + 1, ncode = (hdr >> 10) + 4;
    if (nlen > 286 || ndist > 30) return false;
    const uint64_t y = (bits33(p) & 0x3fffffffull) | ((bits33(p + 30) & 0x7ffffffull) << 30);
    p += 3 * ncode;
    // lengths by symbol (three bits each), counts by length (a byte each)
    constexpr uint32_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint64_t bysym = 0, counts = 0;
#pragma unroll
    for (uint32_t i = 0; i < 19; i++) {
        const uint32_t l = i < ncode ? (uint32_t)(y >> (3 * i)) & 7u : 0u;
        bysym |= (uint64_t)l << (3 * order[i]);
        counts += l ? 1ull << (8 * l) : 0ull;
    }
    uint32_t k = 0;
    for (uint32_t l = 1; l <= 7; l++)
        for (uint32_t sy = 0; sy < 19; sy++) if (((uint32_t)(bysym >> (3 * sy)) & 7u) == l) sorted[k++] = (uint8_t)sy;
    const uint32_t all = nlen + ndist;
    uint32_t have = 0, prev = 0, kl = 0, kd = 0, eob = 0, big = 0; // big: bit 0 a literal/length code longer than one bit, bit 1 a distance code
    while (have < all) {
        if (p + 14 > total_bits) return false;
        uint32_t w = (uint32_t)bits33(p);
        uint32_t code = 0, first = 0, index = 0, sym = 0xffu, len = 1;
        for (; len <= 7; len++) {
            code |= w & 1u; w >>= 1;
            const uint32_t cnt = (uint32_t)(counts >> (8 * len)) & 255u;
            if (code < first + cnt) { sym = sorted[index + code - first]; break; }
            index += cnt; first = (first + cnt) << 1; code <<= 1;
        }
        if (sym == 0xffu) return false;
        p += len;
        uint32_t rep = 1, val = sym;
        if (sym >= 16) {
            if (sym == 16) { if (have == 0) return false; val = prev; rep = 3 + (w & 3u); p += 2; }
            else if (sym == 17) { val = 0; rep = 3 + (w & 7u); p += 3; }
            else { val = 0; rep = 11 + (w & 127u); p += 7; }
            if (have + rep > all) return false;
        }
        if (val) {
            const uint32_t inl = have >= nlen ? 0u : (have + rep <= nlen ? rep : nlen - have), unit = 32768u >> val;
            kl += inl * unit; kd += (rep - inl) * unit;
            if (kl > 32768u || kd > 32768u) return false;
            if (val > 1) big |= (inl ? 1u : 0u) | (rep > inl ? 2u : 0u);
            if (have <= 256 && have + rep > 256) eob = val;
        }
        prev = val; have += rep;
    }
    const bool lit_ok = kl == 32768u || (kl == 16384u && !(big & 1u));
    const bool dist_ok = kd == 32768u || kd == 0u || (kd == 16384u && !(big & 2u));
    return eob != 0 && lit_ok && dist_ok;
}
#ifdef ZGPU_FIND_TIME // debug build only: clock per phase of the block finder, summed over the finders
__device__ unsigned long long find_time[8];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_find_time(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(find_time), sizeof z);
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(find_time), z, sizeof z);
}
#define FT(i) do { const unsigned long long t_ = wall_clock64(); ft[i] += t_ - ftp; ftp = t_; } while (0)
#else
#define FT(i) do { } while (0)
#endif
__global__ void __launch_bounds__(64) spec_find_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes, uint64_t spacing, uint32_t ntargets, uint64_t *found)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    InflateLdsFind &L = *reinterpret_cast<InflateLdsFind *>(lds_raw);
    const uint32_t t = blockIdx.x + 1, lane = threadIdx.x;
    if (t > ntargets) return;
    const uint64_t total_bits = in_bytes * 8, lo_bit = (uint64_t)t * spacing * 8;
    const uint64_t hi_bit = (uint64_t)(t + 1) * spacing * 8 < total_bits ? (uint64_t)(t + 1) * spacing * 8 : total_bits;
    const uint32_t *g32 = reinterpret_cast<const uint32_t *>(in); // (the input buffer is a device allocation: aligned)
    const uint64_t gdwords = (in_bytes + 3) >> 2;
    uint64_t result = ~0ull;
#ifdef ZGPU_FIND_TIME
    unsigned long long ft[8] = {}, ftp = wall_clock64(), nval = 0;
#endif
    // does a dynamic block the decoder would accept start at bit `cand` (its three type bits are not looked at)?
    auto dynamic_at = [&](uint64_t cand) -> bool {
        BitSrc b;
        b.g32 = g32; b.gdwords = gdwords; b.d0 = cand >> 5; b.filled = 0; b.rd = 0; b.hold = 0; b.bits = 0;
        const uint64_t left = total_bits - (cand & ~31ull);
        b.seg_bits = left > 0xFFFF0000ull ? 0xFFFF0000u : (uint32_t)left;
        wave_sync();
        stage_fill(b, L.stage, lane);
        wave_sync();
        prime(b, L.stage);
        refill(b, L.stage); refill(b, L.stage);
        drop(b, (uint32_t)cand & 31u);
        drop(b, 3);
        CodeRows lr{}, dr{};
        const uint32_t err = dynamic_header<true>(L, b, lane, lr, dr);
        wave_sync();
        // (a block needs its end-of-block code; inflate_table does not ask for it, a block start worth trusting does)
        return !err && uni(L.lens[256]) != 0 && consumed_bits(b) <= b.seg_bits;
    };
    // Stored blocks: data that does not compress (an archive of compressed files) arrives in them, full of block headers that are none of this
    // stream's.  The first byte offset B of the region that reads as LEN, ~LEN behind three zero header bits and zero padding, and whose block is
    // followed by a header that holds as well (stored: LEN, ~LEN again; dynamic: as above) is reported too: the host starts a piece AT its LEN and
    // strikes the dynamic "starts" found inside the block's bytes.
    // the candidates that passed the second sieve wait here (a handful per block) until a lane each can read their code lengths
    uint64_t *wait = reinterpret_cast<uint64_t *>(L.tok); // 64 entries
    uint32_t nwait = 0;
    auto settle = [&]() { // third sieve for up to 64 waiting candidates at once, then the decoder's parse for what is left, in order
        const uint64_t cb = lane < nwait ? wait[lane] : 0ull;
        const bool yes = lane < nwait && lane_header_ok(g32, gdwords, cb, total_bits, reinterpret_cast<uint8_t *>(L.ltab) + lane * 20);
        uint64_t mm = __ballot(yes);
        nwait = 0;
        wave_sync();
        while (mm && result == ~0ull) {
            const uint32_t l = (uint32_t)__builtin_ctzll(mm); mm &= mm - 1;
            const uint64_t cand = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cb >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cb, (int)l);
#ifdef ZGPU_FIND_TIME
            nval++;
#endif
            if (dynamic_at(cand)) result = cand;
        }
    };
    uint64_t stored = ~0ull;
    {
        const uint64_t lo_byte = (uint64_t)t * spacing, hi_byte = hi_bit >> 3;
        for (uint64_t o0 = lo_byte; o0 < hi_byte && stored == ~0ull; o0 += 256) {
            const uint64_t o = o0 + lane * 4, wi = o >> 2; // (lo_byte is a multiple of 4096, t >= 1: wi >= 1)
            const uint32_t dp = wi - 1 < gdwords ? g32[wi - 1] : 0u, d0 = wi < gdwords ? g32[wi] : 0u, d1 = wi + 1 < gdwords ? g32[wi + 1] : 0u;
            uint32_t hit = 0, wsel = 0;
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                const uint32_t w = k == 0 ? d0 : __builtin_amdgcn_alignbyte(d1, d0, k), pb = k == 0 ? dp >> 24 : (d0 >> (8 * (k - 1))) & 255u;
                if (((w ^ (w >> 16)) & 0xFFFFu) == 0xFFFFu && (pb >> 5) == 0 && o + k + 4 <= in_bytes && o + k < hi_byte) { hit |= 1u << k; }
            }
            uint64_t m = __ballot(hit != 0);
            while (m && stored == ~0ull) {
                const uint32_t l = (uint32_t)__builtin_ctzll(m); m &= m - 1;
                uint32_t hk = (uint32_t)__builtin_amdgcn_readlane((int)hit, (int)l);
                const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)d0, (int)l), e1 = (uint32_t)__builtin_amdgcn_readlane((int)d1, (int)l);
                while (hk && stored == ~0ull) {
                    const uint32_t k = (uint32_t)__builtin_ctz(hk); hk &= hk - 1;
                    const uint64_t B = o0 + l * 4 + k;
                    const uint32_t len = (uint32_t)((((uint64_t)e1 << 32) | e0) >> (8 * k)) & 0xFFFFu;
                    const uint64_t N = B + 4 + len; // the header behind the block: it begins a byte
                    if (N + 5 > in_bytes) continue;
                    const uint32_t hb = in[N], type = (hb >> 1) & 3u;
                    bool good = false;
                    if (type == 0) good = ((((uint32_t)in[N + 1] | ((uint32_t)in[N + 2] << 8)) ^ ((uint32_t)in[N + 3] | ((uint32_t)in[N + 4] << 8))) & 0xFFFFu) == 0xFFFFu && (hb >> 3) == 0;
                    else if (type == 2) good = dynamic_at(N * 8);
                    if (good) stored = B;
                }
            }
        }
    }
    FT(0); // the stored sieve
    // the bytes to scan come through LDS, kScanBytes at a time
    uint32_t *scan = reinterpret_cast<uint32_t *>(L.out);
    const uint4 *g128 = reinterpret_cast<const uint4 *>(in);
    const uint64_t gvecs = (in_bytes + 15) >> 4; // (the allocation behind `in` is padded: engine_ensure_stage)
    for (uint64_t blk = lo_bit; blk < hi_bit && result == ~0ull; blk += kScanBytes * 8) {
        wave_sync();
#pragma unroll
        for (uint32_t k = 0; k < kScanBytes / 16 / 64 + 1; k++) {
            const uint32_t v = k * 64 + lane;
            const uint64_t gv = (blk >> 7) + v;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (v <= kScanBytes / 16 && gv < gvecs) q = g128[gv];
            if (v <= kScanBytes / 16) reinterpret_cast<uint4 *>(scan)[v] = q;
        }
        wave_sync();
        const uint64_t blk_hi = blk + kScanBytes * 8 < hi_bit ? blk + kScanBytes * 8 : hi_bit;
        FT(1); // staging
        // Three sieves.  (1) BFINAL 0, BTYPE 2, HLIT <= 29, HDIST <= 29 -- one offset in nine passes -- for 32 offsets per lane at a time, on the 64 bits
        // that start at the lane's first offset: the type bits are ~x & ~(x >> 1) & (x >> 2), a count of 30 or 31 has its upper four bits set; the
        // survivors are listed in LDS in offset order.  (2) whenever 64 are listed (and at the end of the block), one per lane: the code-length
        // code must be complete (inftrees.c:106-138: sum of 2^-len == 1).  (3) what is left, in order, through the decoder's own header parse.
        uint32_t listed = 0;
        uint16_t *list = reinterpret_cast<uint16_t *>(L.out + kScanBytes + 64);
        for (uint64_t base = blk; base < blk_hi + 2048 && result == ~0ull; base += 2048) {
            if (base < blk_hi) {
                const uint32_t rel0 = (uint32_t)(base - blk) + lane * 32, wi = rel0 >> 5;
                const uint64_t x = ((uint64_t)scan[wi + 1] << 32) | scan[wi];
                uint64_t cm = ~x & ~(x >> 1) & (x >> 2) & ~((x >> 4) & (x >> 5) & (x >> 6) & (x >> 7)) & ~((x >> 9) & (x >> 10) & (x >> 11) & (x >> 12)) & 0xFFFFFFFFull;
                const uint64_t left = base + lane * 32 < blk_hi ? blk_hi - (base + lane * 32) : 0; // offsets of this lane inside the block
                if (left < 32) cm &= (1ull << left) - 1;
                const uint32_t cnt = (uint32_t)__builtin_popcountll(cm), incl = wave_prefix_sum(cnt);
                uint32_t at = listed + incl - cnt;
                while (cm) { list[at++] = (uint16_t)(rel0 + (uint32_t)__builtin_ctzll(cm)); cm &= cm - 1; }
                listed += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (listed < 64 && base + 2048 < blk_hi) continue;
            }
            FT(2); // sieve 1
            wave_sync();
            uint32_t lhead = 0; // the list is taken from the front, 64 at a time; what is left (fewer than 64) moves down behind the loop
            while (listed && result == ~0ull && (listed >= 64 || base + 2048 >= blk_hi)) {
                const uint32_t take = listed < 64 ? listed : 64;
                const uint32_t rel = lane < take ? list[lhead + lane] : 0u, wi = rel >> 5, sh = rel & 31u;
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = scan[wi + k];
                const uint32_t b0 = __builtin_amdgcn_alignbit(w[1], w[0], sh), b1 = __builtin_amdgcn_alignbit(w[2], w[1], sh), b2 = __builtin_amdgcn_alignbit(w[3], w[2], sh);
                const uint32_t ncode = ((b0 >> 13) & 15u) + 4;
                uint64_t y = ((((uint64_t)b1 << 32) | b0) >> 17) | ((uint64_t)b2 << 47);
                y &= (1ull << (3 * ncode)) - 1; // lengths that are not sent are 0
                const uint32_t ylo = (uint32_t)y, ymid = (uint32_t)(y >> 30);
                uint32_t kraft = 0;
#pragma unroll
                for (uint32_t i = 0; i < 10; i++) kraft += (128u >> ((ylo >> (3 * i)) & 7u)) & 127u; // a length of 0 counts nothing
#pragma unroll
                for (uint32_t i = 0; i < 9; i++) kraft += (128u >> ((ymid >> (3 * i)) & 7u)) & 127u;
                const bool ok = lane < take && blk + rel + 17 + 3 * ncode < hi_bit && kraft == 128;
                uint64_t m = __ballot(ok);
                lhead += take; listed -= take;
                FT(3); // sieve 2
                if (m) {
                    const uint32_t add = (uint32_t)__builtin_popcountll(m);
                    if (nwait + add > 64) { settle(); FT(4); }
                    if (ok) wait[nwait + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = blk + rel;
                    nwait += add;
                    wave_sync();
                }
            }
            if (nwait >= 32 || (nwait && base + 2048 >= blk_hi && blk + kScanBytes * 8 >= hi_bit)) { settle(); FT(4); } // (half a wave of them, or the region's last)
            if (lhead) { // fewer than 64 are left: to the front
                const uint32_t moved = lane < listed ? list[lhead + lane] : 0u;
                wave_sync();
                if (lane < listed) list[lane] = (uint16_t)moved;
                wave_sync();
            }
        }
    }
    if (lane == 0) { found[t - 1] = result; found[ntargets + t - 1] = stored; }
#ifdef ZGPU_FIND_TIME
    if (lane == 0) { ft[5] = nval; ft[6] = 1; for (int i = 0; i < 8; i++) if (ft[i]) atomicAdd(&find_time[i], ft[i]); }
#endif
}

// The windows: window[i] = the last 32 KiB of the output up to the end of piece i = piece i's tail with its markers looked up in window[i - 1] -- a
// chain as long as the stream has pieces.  Looking up is associative, so the chain is cut into groups:
//   spec_window_rel_kernel  one workgroup per group: piece by piece, the tail with its markers looked up in the previous RELATIVE window, whose own
//                           markers name bytes of the window in front of the group (kept in place of the tail);
//   spec_window_grp_kernel  one workgroup: group by group, the window behind the group's last piece as bytes (the only chain over the whole stream);
//   spec_window_abs_kernel  one workgroup per piece: its relative window with the markers looked up in the window in front of its group.
// A marker that names a byte in front of the stream's first is found out by spec_resolve_kernel (every produced byte passes there).
__device__ inline void window_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } // (loads of the next tail stay in flight)
__device__ inline uint4 lookup8(uint4 q, const uint16_t *prev) // eight symbols; markers replaced by prev[index] (a symbol again)
{
    uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t lo = ws[k] & 0xFFFFu, hi = ws[k] >> 16;
        if (lo & 0x8000u) lo = prev[lo & 0x7FFFu];
        if (hi & 0x8000u) hi = prev[hi & 0x7FFFu];
        ws[k] = lo | (hi << 16);
    }
    return make_uint4(ws[0], ws[1], ws[2], ws[3]);
}
__global__ void __launch_bounds__(1024) spec_window_rel_kernel(uint16_t *tails, uint32_t nseg, uint32_t group)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint16_t *win = reinterpret_cast<uint16_t *>(lds_raw); // two windows of kOutRing symbols
    const uint32_t tid = threadIdx.x, first = blockIdx.x * group, last = first + group < nseg ? first + group : nseg;
    if (first >= nseg) return;
    constexpr uint32_t kPer = kOutRing / 8 / 1024;
    uint4 nxt[kPer];
#pragma unroll
    for (uint32_t r = 0; r < kPer; r++) nxt[r] = reinterpret_cast<const uint4 *>(tails + (uint64_t)first * kOutRing)[r * 1024 + tid];
    for (uint32_t i = first; i < last; i++) {
        const uint16_t *prev = win + ((i + 1) & 1) * kOutRing;
        uint16_t *cur = win + (i & 1) * kOutRing;
        uint4 q4[kPer];
#pragma unroll
        for (uint32_t r = 0; r < kPer; r++) q4[r] = nxt[r];
        if (i + 1 < last) {
#pragma unroll
            for (uint32_t r = 0; r < kPer; r++) nxt[r] = reinterpret_cast<const uint4 *>(tails + (uint64_t)(i + 1) * kOutRing)[r * 1024 + tid];
        }
#pragma unroll
        for (uint32_t r = 0; r < kPer; r++) {
            const uint4 o = i == first ? q4[r] : lookup8(q4[r], prev); // (the group's first piece is relative to the window in front of the group as it is)
            reinterpret_cast<uint4 *>(cur)[r * 1024 + tid] = o;
            if (i != first) reinterpret_cast<uint4 *>(tails + (uint64_t)i * kOutRing)[r * 1024 + tid] = o;
        }
        window_barrier();
    }
}
__global__ void __launch_bounds__(1024) spec_window_grp_kernel(const uint16_t *__restrict__ tails, uint32_t nseg, uint32_t group, uint8_t *__restrict__ entry)
{
    // entry[g] = the window in front of group g, as bytes (group 0: nothing is known, and nothing valid refers to it)
    __shared__ uint16_t win[2][kOutRing]; // bytes, kept as symbols so that lookup8 serves
    const uint32_t tid = threadIdx.x, ngroups = (nseg + group - 1) / group;
    constexpr uint32_t kPer = kOutRing / 8 / 1024;
    for (uint32_t r = 0; r < kPer; r++) reinterpret_cast<uint4 *>(win[1])[r * 1024 + tid] = make_uint4(0, 0, 0, 0);
    for (uint32_t r = tid; r < kOutRing / 16; r += 1024) reinterpret_cast<uint4 *>(entry)[r] = make_uint4(0, 0, 0, 0);
    window_barrier();
    uint4 nxt[kPer];
    auto last_of = [&](uint32_t g) { return (g + 1) * group < nseg ? (g + 1) * group - 1 : nseg - 1; };
#pragma unroll
    for (uint32_t r = 0; r < kPer; r++) nxt[r] = reinterpret_cast<const uint4 *>(tails + (uint64_t)last_of(0) * kOutRing)[r * 1024 + tid];
    for (uint32_t g = 0; g + 1 < ngroups; g++) {
        const uint16_t *prev = win[(g + 1) & 1];
        uint16_t *cur = win[g & 1];
        uint4 q4[kPer];
#pragma unroll
        for (uint32_t r = 0; r < kPer; r++) q4[r] = nxt[r];
        if (g + 2 < ngroups) {
#pragma unroll
            for (uint32_t r = 0; r < kPer; r++) nxt[r] = reinterpret_cast<const uint4 *>(tails + (uint64_t)last_of(g + 1) * kOutRing)[r * 1024 + tid];
        }
#pragma unroll
        for (uint32_t r = 0; r < kPer; r++) {
            uint4 o = lookup8(q4[r], prev);
            o.x &= 0x00FF00FFu; o.y &= 0x00FF00FFu; o.z &= 0x00FF00FFu; o.w &= 0x00FF00FFu; // (what was a marker in group 0's entry is a byte nobody may use)
            reinterpret_cast<uint4 *>(cur)[r * 1024 + tid] = o;
            const uint32_t b0 = (o.x & 255u) | ((o.x >> 8) & 0xFF00u) | ((o.y & 255u) << 16) | ((o.y >> 16) << 24);
            const uint32_t b1 = (o.z & 255u) | ((o.z >> 8) & 0xFF00u) | ((o.w & 255u) << 16) | ((o.w >> 16) << 24);
            reinterpret_cast<uint2 *>(entry + (uint64_t)(g + 1) * kOutRing)[r * 1024 + tid] = make_uint2(b0, b1);
        }
        window_barrier();
    }
}
__global__ void __launch_bounds__(256) spec_window_abs_kernel(const uint16_t *__restrict__ tails, uint32_t nseg, uint32_t group, const uint8_t *__restrict__ entry,
                                                              uint8_t *__restrict__ windows)
{
    const uint32_t i = blockIdx.x;
    if (i >= nseg) return;
    const uint8_t *e = entry + (uint64_t)(i / group) * kOutRing;
    const uint4 *t4 = reinterpret_cast<const uint4 *>(tails + (uint64_t)i * kOutRing);
    for (uint32_t v = threadIdx.x; v < kOutRing / 8; v += 256) {
        const uint4 q = t4[v];
        const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
        uint32_t o8[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t sym = (ws[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            const uint32_t byte = (sym & 0x8000u) ? e[sym & 0x7FFFu] : (sym & 255u);
            o8[k >> 2] |= byte << ((k & 3) * 8);
        }
        reinterpret_cast<uint2 *>(windows + (uint64_t)i * kOutRing)[v] = make_uint2(o8[0], o8[1]);
    }
}

// One workgroup per page of symbols: to its place in the output, markers through the window of the piece in front.
__global__ void __launch_bounds__(256) spec_resolve_kernel(const uint16_t *__restrict__ mid, const uint64_t *__restrict__ page_owner, uint32_t npages, const SpecEnd *__restrict__ ends,
                                                           uint32_t nseg, const uint8_t *__restrict__ windows, const uint64_t *__restrict__ out_start, uint32_t dict_len,
                                                           uint8_t *__restrict__ out, uint64_t out_cap, uint32_t *flag)
{
    const uint32_t pg = blockIdx.x;
    if (pg >= npages) return;
    const uint64_t ow = page_owner[pg];
    const uint32_t seg = (uint32_t)(ow >> 32), k = (uint32_t)ow;
    if (seg >= nseg) return; // a piece behind the end of the stream
    const uint32_t len = ends[seg].out_bytes, from = k * kOutHalf;
    if (from >= len) return;
    const uint32_t n = len - from < kOutHalf ? len - from : kOutHalf;
    const uint64_t at = out_start[seg] + from;
    const uint64_t have_prev = seg == 0 ? 0 : out_start[seg] + dict_len;
    const uint32_t vf_prev = have_prev >= kOutRing ? 0u : kOutRing - (uint32_t)have_prev;
    const uint8_t *win = seg ? windows + (uint64_t)(seg - 1) * kOutRing : windows;
    const uint16_t *src = mid + (uint64_t)pg * kOutHalf;
    uint32_t bad = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t sym = src[i];
        uint32_t byte = sym & 255u;
        if (sym & 0x8000u) {
            const uint32_t idx = sym & 0x7FFFu;
            if (seg != 0 && idx >= vf_prev) byte = win[idx]; else { byte = 0; bad = 1; }
        }
        if (at + i < out_cap) out[at + i] = (uint8_t)byte;
    }
    if (bad) atomicOr(flag, 1u);
}

// 0: decoded (res complete); 1: not this way (the caller uses the one-workgroup decoder); anything else: an error of the engine
// start_bit (0..7): the deflate data begins at that bit of the first byte (a stream taken up again where an earlier call's last whole piece ended)
static int inflate_spec_run(zgpu_engine *e, const uint8_t *d_in, const uint8_t *h_in, uint64_t in_bytes, uint8_t *d_out, uint64_t out_cap, zgpu_inflate_result *res,
                            hipStream_t st, uint32_t stream_mode, uint32_t start_bit = 0, bool force = false)
{
    // force: whatever the size (a stream that goes on at a bit offset has no other decoder: one piece is one workgroup)
    static long min_bytes = -1;
    if (min_bytes < 0) { const char *v = getenv("ZGPU_SPEC_MIN_BYTES"); min_bytes = v ? atol(v) : 128 * 1024; }
    if ((!force && (long)in_bytes < min_bytes) || in_bytes >= (1ull << 40) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return 1;
    res->adler32 = 1; res->crc32 = 0; res->in_used = in_bytes; res->in_used_bits = 0; res->stream_end = 0; res->incomplete = 0;
    uint64_t spacing = (in_bytes / 4096 + 4095) & ~4095ull;
    if (spacing < 32768) spacing = 32768;
    // the finders stand four times as close as the pieces will be (each scans to the next finder at most; of what they find the host keeps starts
    // at least three quarters of `spacing` apart)
    const uint64_t fspacing = spacing / 4 < 16384 ? 16384 : (spacing / 4 + 4095) & ~4095ull;
    const uint32_t ntargets = (uint32_t)((in_bytes - 1) / fspacing);
    if (ntargets < 3 && !force) return 1;
    uint64_t *d_found = static_cast<uint64_t *>(engine_scratch(e, (size_t)ntargets * 16 + 64));
    if (!d_found) return engine_fail(e, ZGPU_MEM_ERROR, "inflate scratch");
    static bool opt_in = false;
    if (!opt_in) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel_t<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InflateLdsSpec));
        hipFuncSetAttribute(reinterpret_cast<const void *>(spec_find_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InflateLdsFind));
        hipFuncSetAttribute(reinterpret_cast<const void *>(spec_window_rel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kOutRing));
        opt_in = true;
    }
    hipEvent_t ev{};
    prof_span_begin(e, st, &ev);
    if (ntargets) hipLaunchKernelGGL(spec_find_kernel, dim3(ntargets), dim3(64), sizeof(InflateLdsFind), st, d_in, in_bytes, fspacing, ntargets, d_found);
    std::vector<uint64_t> found(2 * (size_t)ntargets);
    if (ntargets) ZGPU_HIP_CHECK(hipMemcpyAsync(found.data(), d_found, found.size() * 8, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    // starts: bit positions; kind 1 = the LEN of a stored block (a byte position; the block's header bits lie up to 10 bits in front)
    const uint64_t kStoredFlag = 1ull << 62;
    std::vector<uint64_t> starts;
    std::vector<uint8_t> kind;
    {
        // what lies inside a stored block the finders vouch for is data, whatever it looks like
        std::vector<std::pair<uint64_t, uint64_t>> raw; // [first bit, behind the last bit) of stored data
        for (uint32_t i = 0; i < ntargets; i++) {
            const uint64_t B = found[ntargets + i];
            if (B != ~0ull && B + 4 <= in_bytes) raw.emplace_back(B * 8, (B + 4 + ((uint64_t)h_in[B] | ((uint64_t)h_in[B + 1] << 8))) * 8);
        }
        std::vector<std::pair<uint64_t, uint8_t>> all;
        for (uint32_t i = 0; i < ntargets; i++) {
            const uint64_t d = found[i], B = found[ntargets + i];
            if (B != ~0ull) all.emplace_back(B * 8, (uint8_t)1);
            if (d == ~0ull) continue;
            // (raw ascends with the finders; a stored block is at most 64 KiB and the finders stand 16 KiB apart or more: a few entries can reach d)
            size_t q = (size_t)(std::upper_bound(raw.begin(), raw.end(), std::pair<uint64_t, uint64_t>(d, ~(uint64_t)0)) - raw.begin());
            bool inside = false;
            for (int back = 0; back < 8 && q > 0; back++) { q--; inside = inside || (raw[q].first <= d && d < raw[q].second); }
            if (!inside) all.emplace_back(d, (uint8_t)0);
        }
        std::sort(all.begin(), all.end());
        starts.push_back(start_bit); kind.push_back(0);
        for (const auto &c : all)
            if (c.first >= starts.back() + spacing * 6 && c.first + spacing * 2 < in_bytes * 8) { starts.push_back(c.first); kind.push_back(c.second); }
        starts.push_back(in_bytes * 8); kind.push_back(0);
    }
    uint32_t nseg = (uint32_t)starts.size() - 1;
    // does the piece that ended at bit `eb` hand over to start j?
    auto links = [&](uint64_t eb, size_t j) -> bool {
        if (!kind[j]) return eb == starts[j];
        if (eb + 3 > starts[j] || eb + 10 < starts[j]) return false;
        for (uint64_t q = eb; q < starts[j]; q++) if ((h_in[q >> 3] >> (q & 7)) & 1u) return false; // BFINAL 0, stored, padding of zeros
        return true;
    };
    static const bool dbg = getenv("ZGPU_SPEC_DEBUG") != nullptr;
    if (dbg) {
        fprintf(stderr, "[spec] %llu bytes, spacing %llu, %u targets, %u pieces; first starts:", (unsigned long long)in_bytes, (unsigned long long)spacing, ntargets, nseg);
        for (uint32_t i = 0; i < nseg && i < 6; i++) fprintf(stderr, " %llu", (unsigned long long)starts[i]);
        fprintf(stderr, "\n");
    }
    if (nseg < 3 && !force) { prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev); return 1; }
    // pages: what the output can hold, or -- when the caller's buffer is far larger than this stream can fill -- eight times the input first
    uint64_t guess = in_bytes * 8 + (16u << 20);
    int repairs = 0;
    for (int attempt = 0;;) {
        const uint64_t room = out_cap < guess ? out_cap : guess;
        const uint64_t pages64 = room / kOutHalf + nseg + 2;
        if (pages64 >= 0xFFFFFFFFull) { prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev); return 1; }
        const uint32_t page_cap = (uint32_t)pages64;
        // one allocation: starts | ends | out_start | status | counters | page owners | windows | tails | pages
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
        const size_t o_starts = carve((size_t)(nseg + 1) * 8), o_ends = carve((size_t)nseg * sizeof(SpecEnd)), o_ostart = carve((size_t)(nseg + 1) * 8),
                     o_status = carve((size_t)nseg * sizeof(InfStatus)), o_cnt = carve(64), o_owner = carve((size_t)page_cap * 8),
                     o_win = carve((size_t)nseg * kOutRing), o_entry = carve((size_t)(nseg / 8 + 2) * kOutRing), o_tails = carve((size_t)nseg * kOutRing * 2), o_mid = carve((size_t)page_cap * kOutHalf * 2);
        uint8_t *base = static_cast<uint8_t *>(engine_scratch2(e, off));
        if (!base) { prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev); return 1; } // (no room for the symbols: the slow way needs none)
        uint64_t *d_starts = reinterpret_cast<uint64_t *>(base + o_starts), *d_ostart = reinterpret_cast<uint64_t *>(base + o_ostart);
        SpecEnd *d_ends = reinterpret_cast<SpecEnd *>(base + o_ends);
        uint32_t *d_cnt = reinterpret_cast<uint32_t *>(base + o_cnt);
        SpecArgs sp{};
        sp.mid = reinterpret_cast<uint16_t *>(base + o_mid); sp.page_owner = reinterpret_cast<uint64_t *>(base + o_owner); sp.page_count = d_cnt; sp.page_cap = page_cap;
        sp.tails = reinterpret_cast<uint16_t *>(base + o_tails); sp.ends = d_ends;
        std::vector<uint64_t> up(starts);
        for (size_t i = 0; i < up.size(); i++) if (kind[i]) up[i] |= kStoredFlag;
        ZGPU_HIP_CHECK(hipMemcpy(d_starts, up.data(), (size_t)(nseg + 1) * 8, hipMemcpyHostToDevice));
        ZGPU_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 64, st));
        ZGPU_HIP_CHECK(hipMemsetAsync(d_ends, 0xFF, (size_t)nseg * sizeof(SpecEnd), st));
        hipLaunchKernelGGL(inflate_kernel_t<true>, dim3(nseg), dim3(128), sizeof(InflateLdsSpec), st, d_in, in_bytes, d_starts, 0ull, nseg, ~0ull, kWholeStream,
                           d_out, out_cap, reinterpret_cast<InfStatus *>(base + o_status), nullptr, engine_inflate_dict(e), engine_inflate_dict_len(e), 1u, sp);
        ZGPU_HIP_CHECK(hipGetLastError());
        std::vector<SpecEnd> ends(nseg);
        uint32_t cnt[2] = {0, 0};
        ZGPU_HIP_CHECK(hipMemcpyAsync(ends.data(), d_ends, (size_t)nseg * sizeof(SpecEnd), hipMemcpyDeviceToHost, st));
        ZGPU_HIP_CHECK(hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        // the chain
        uint32_t used_seg = 0; uint64_t total = 0; bool ended = false, dry = false;
        for (uint32_t i = 0; i < nseg; i++) {
            if (ends[i].flags >> 8) break;               // an error (or never written)
            total += ends[i].out_bytes; used_seg = i + 1; dry = dry || (ends[i].flags & 2u);
            if (ends[i].flags & 1u) { ended = true; break; }
            if (i + 1 == nseg || !links(ends[i].end_bit, i + 1)) break;
        }
        // A piece that ran past the next start (or several): those were no block starts -- a deflate stream inside a stored block looks like one, and so
        // does one pattern in 10^9 or so.  Every piece on the chain so far began at a real boundary, so where the last one ended is one too: the false starts go,
        // that boundary becomes a start unless it is one, and the pieces are decoded again (three times at most; then the one-workgroup decoder).
        if (!ended && used_seg >= 1 && used_seg < nseg && !(ends[used_seg - 1].flags >> 8) && !links(ends[used_seg - 1].end_bit, used_seg) && repairs < 3) {
            std::vector<uint64_t> fixed(starts.begin(), starts.begin() + used_seg);
            std::vector<uint8_t> fkind(kind.begin(), kind.begin() + used_seg);
            uint32_t i = used_seg - 1;
            for (;;) { // follow the chain as far as it goes over the starts that are real
                const uint64_t eb = ends[i].end_bit;
                if (eb >= in_bytes * 8) break;
                const uint32_t j = (uint32_t)(std::lower_bound(starts.begin() + i + 1, starts.begin() + nseg, eb) - starts.begin());
                if (j < nseg && links(eb, j) && !(ends[j].flags >> 8) && !(ends[j].flags & 1u)) { fixed.push_back(starts[j]); fkind.push_back(kind[j]); i = j; continue; }
                fixed.push_back(eb); fkind.push_back(0);
                for (uint32_t k = j; k < nseg; k++) if (starts[k] > eb + 10) { fixed.push_back(starts[k]); fkind.push_back(kind[k]); } // (unchecked from here on)
                break;
            }
            if (dbg) fprintf(stderr, "[spec] chain broke behind piece %u (end %llu, next start %llu): %u pieces -> %zu, decoding again\n", used_seg - 1,
                             (unsigned long long)ends[used_seg - 1].end_bit, (unsigned long long)starts[used_seg], nseg, fixed.size());
            fixed.push_back(in_bytes * 8); fkind.push_back(0);
            starts.swap(fixed); kind.swap(fkind);
            nseg = (uint32_t)starts.size() - 1;
            repairs++;
            if (nseg < 2 && !force) { prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev); return 1; }
            continue;
        }
        // stream mode, every piece chained and the last one ran out of input inside a block: the stream is not all there yet, and the pieces in front
        // of the last one are delivered -- they end at a block boundary (a bit position), where the next call takes the stream up again with their last
        // 32 KiB as its window (inflate.c:323-371 updatewindow)
        const bool partial = !ended && stream_mode && used_seg >= 1 && used_seg + 1 == nseg && (ends[nseg - 1].flags >> 8) == kMsgTruncated && links(ends[used_seg - 1].end_bit, used_seg);
        const uint64_t end_bit = used_seg ? ends[used_seg - 1].end_bit : 0;
        const uint64_t end_byte = ended ? (end_bit + 7) >> 3 : partial ? end_bit >> 3 : 0;
        if (dbg) {
            fprintf(stderr, "[spec] chain: %u of %u pieces, ended %d, total %llu, pages %u of %u, dry %d\n", used_seg, nseg, (int)ended, (unsigned long long)total, cnt[0], page_cap, (int)dry);
            for (uint32_t i = used_seg ? used_seg - 1 : 0; i < nseg && i < used_seg + 2; i++)
                fprintf(stderr, "[spec]   piece %u: start %llu end %llu next %llu out %u flags %#x\n", i, (unsigned long long)starts[i], (unsigned long long)ends[i].end_bit,
                        (unsigned long long)starts[i + 1], ends[i].out_bytes, ends[i].flags);
        }
        if (!(ended || partial) || (!stream_mode && end_byte != in_bytes)) { prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev); return 1; }
        if (total > out_cap) {
            prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev);
            res->out_bytes = total; res->first_bad_chunk = -1; res->error_code = ZGPU_BUF_ERROR; res->error_msg = 0;
            engine_collect(e);
            return engine_fail(e, ZGPU_BUF_ERROR, "output capacity too small");
        }
        if (dry) { // the pool was sized from the guess: now the size is known
            if (attempt) { prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev); return 1; }
            guess = total + kOutHalf;
            attempt++;
            continue;
        }
        const uint32_t npages = cnt[0] < page_cap ? cnt[0] : page_cap;
        {
            std::vector<uint64_t> ostart(used_seg + 1);
            uint64_t pos = 0;
            for (uint32_t i = 0; i < used_seg; i++) { ostart[i] = pos; pos += ends[i].out_bytes; }
            ostart[used_seg] = pos;
            ZGPU_HIP_CHECK(hipMemcpyAsync(d_ostart, ostart.data(), (size_t)(used_seg + 1) * 8, hipMemcpyHostToDevice, st));
            ZGPU_HIP_CHECK(hipStreamSynchronize(st)); // (ostart is a local)
            uint32_t group = 8;
            while (group * group < used_seg) group++;
            const uint32_t ngroups = (used_seg + group - 1) / group;
            hipLaunchKernelGGL(spec_window_rel_kernel, dim3(ngroups), dim3(1024), 4 * kOutRing, st, sp.tails, used_seg, group);
            hipLaunchKernelGGL(spec_window_grp_kernel, dim3(1), dim3(1024), 0, st, sp.tails, used_seg, group, base + o_entry);
            hipLaunchKernelGGL(spec_window_abs_kernel, dim3(used_seg), dim3(256), 0, st, sp.tails, used_seg, group, base + o_entry, base + o_win);
        }
        if (npages) hipLaunchKernelGGL(spec_resolve_kernel, dim3(npages), dim3(256), 0, st, sp.mid, sp.page_owner, npages, d_ends, used_seg, base + o_win, d_ostart,
                                       engine_inflate_dict_len(e), d_out, out_cap, d_cnt + 4);
        ZGPU_HIP_CHECK(hipGetLastError());
        uint32_t flag = 0;
        ZGPU_HIP_CHECK(hipMemcpyAsync(&flag, d_cnt + 4, 4, hipMemcpyDeviceToHost, st));
        prof_span_end(e, st, ZGPU_STAGE_INFLATE, ev);
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        if (dbg) fprintf(stderr, "[spec] resolved %u pages, flag %u\n", npages, flag);
        if (flag) return 1;
        res->out_bytes = total; res->first_bad_chunk = -1; res->error_code = 0; res->error_msg = 0;
        res->in_used = end_byte; res->in_used_bits = partial ? (uint32_t)(end_bit & 7u) : 0u; res->stream_end = (stream_mode && ended) ? 1 : 0; res->incomplete = partial ? 1 : 0;
        if (ended) t_end_bits = (uint32_t)(end_bit & 7u);
        g_spec_done++;
        return output_checksums(e, d_out, total, out_cap, res, st);
    }
}
} // namespace zgpu

// Decode a raw deflate body made of full-flush-separated segments without a side table.  Candidate boundaries are the
// marker positions; a candidate that is not a real boundary (the pattern can occur inside stored or coded data) makes its
// segment fail to decode and is merged away.  On success *offsets_out (optional) receives the validated boundaries.
// flags & ZGPU_INF_STREAM: `in` is the rest of a stream, not a delimited body: it ends where its final block ends (res->in_used,
// res->stream_end) whatever follows, and input that stops inside a block yields the segments before it (res->incomplete).
static int inflate_stream_host(zgpu_engine *e, const void *in, uint64_t in_bytes, uint32_t flags, void *out, uint64_t out_cap, zgpu_inflate_result *res,
                               std::vector<uint64_t> *offsets_out, uint32_t start_bit = 0)
{
    if (!e || !in || !res || in_bytes == 0 || start_bit > 7) return engine_fail(e, ZGPU_STREAM_ERROR, "bad inflate arguments");
    ZGPU_HIP_CHECK(hipSetDevice(engine_device(e)));
    hipStream_t st = engine_stream(e);
    const uint32_t stream_mode = (flags & ZGPU_INF_STREAM) ? 1u : 0u;
    if (start_bit) { // the stream goes on inside its first byte (behind the last whole piece of an earlier call): the pieces are the decoder that starts at a bit
        int rc0 = engine_ensure_stage(e, in_bytes + 256, out_cap ? out_cap : 1);
        if (rc0) return rc0;
        uint8_t *d_in0 = engine_stage_in(e);
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_in0, in, in_bytes, hipMemcpyHostToDevice, st));
        // (ZGPU_SPEC_DECLINE_AT_BIT=1, tests: the pieces say "not this way" although the stream is whole)
        const int src = getenv("ZGPU_SPEC_DECLINE_AT_BIT") ? 1 : inflate_spec_run(e, d_in0, static_cast<const uint8_t *>(in), in_bytes, engine_stage_out(e), out_cap, res, st, stream_mode, start_bit, true);
        if (src == ZGPU_OK) { if (out && res->out_bytes) ZGPU_HIP_CHECK(hipMemcpy(out, engine_stage_out(e), res->out_bytes, hipMemcpyDeviceToHost)); return ZGPU_OK; }
        if (src != 1) return src;
        // the pieces do not chain: damage, most likely.  The verdict is the one-workgroup decoder's, on a copy of the stream that starts at bit 0
        std::vector<uint8_t> sh(in_bytes);
        const uint8_t *p0 = static_cast<const uint8_t *>(in);
        for (uint64_t i = 0; i < in_bytes; i++) sh[i] = (uint8_t)((p0[i] >> start_bit) | ((i + 1 < in_bytes ? p0[i + 1] : 0) << (8 - start_bit)));
        zgpu_inflate_result r2;
        t_end_bits = 0;
        const int rc2 = inflate_stream_host(e, sh.data(), in_bytes, flags, out, out_cap, &r2, nullptr, 0);
        if (rc2 != ZGPU_OK) { *res = r2; return rc2; }
        if (r2.incomplete && !r2.out_bytes) { *res = r2; res->in_used = 0; res->in_used_bits = start_bit; return ZGPU_OK; } // (not all there yet: nothing taken)
        // The shifted copy decoded (the pieces had said "not this way" for a harmless reason: no scratch room, too many repairs of the chain -- stored blocks
        // full of what reads as headers --, a flag of the resolve pass): its result stands, with the positions mapped back to the caller's bytes
        *res = r2;
        if (r2.stream_end) { const uint64_t bits = (r2.in_used ? (r2.in_used - 1) * 8 + (t_end_bits ? t_end_bits : 8u) : 0) + start_bit; res->in_used = (bits + 7) >> 3; res->in_used_bits = 0; }
        else { const uint64_t bits = r2.in_used * 8 + r2.in_used_bits + start_bit; res->in_used = bits >> 3; res->in_used_bits = (uint32_t)(bits & 7u); }
        return ZGPU_OK;
    }
    const uint64_t max_cand = in_bytes / 5 + 2;
    int rc = engine_ensure_stage(e, in_bytes + 64 + (max_cand + 2) * 2 * sizeof(uint64_t) + 64, out_cap ? out_cap : 1);
    if (rc) return rc;
    uint8_t *d_in = engine_stage_in(e);
    const uint64_t tab_off = (in_bytes + 127) & ~63ull;
    uint64_t *d_cand = reinterpret_cast<uint64_t *>(d_in + tab_off);          // candidates, later the offsets table
    uint64_t *d_offs = d_cand + max_cand + 2;
    uint32_t *d_count = static_cast<uint32_t *>(engine_scratch(e, 64 * 1024));
    ZGPU_HIP_CHECK(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, st));
    ZGPU_HIP_CHECK(hipMemsetAsync(d_count, 0, 4, st));
    hipLaunchKernelGGL(marker_scan_kernel, dim3(2048), dim3(256), 0, st, d_in, in_bytes, d_cand, (uint32_t)max_cand, d_count);
    uint32_t ncand = 0;
    ZGPU_HIP_CHECK(hipMemcpyAsync(&ncand, d_count, 4, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<uint64_t> b(ncand + 2);
    if (ncand) ZGPU_HIP_CHECK(hipMemcpy(b.data() + 1, d_cand, (size_t)ncand * sizeof(uint64_t), hipMemcpyDeviceToHost));
    b[0] = 0;
    std::sort(b.begin() + 1, b.begin() + 1 + ncand);
    b[ncand + 1] = in_bytes;
    b.erase(std::unique(b.begin(), b.end()), b.end()); // a marker can end exactly at the end of the body
    // A candidate that is not a boundary costs one more pass over the input; streams of other producers (sync-flushed protocols put a
    // marker behind every message and keep the window across it) can hold thousands that are none.  So: a few passes that merge the
    // failing segment into its successor, then -- and at once when a segment fails in the way a kept window looks (a distance that
    // reaches back before the segment, more than 64 KiB of output) -- the stream is decoded from end to end by one workgroup.
    bool whole = false;
    // not one marker in a long body: no chunked stream of this library looks like that -- the pieces are tried at once (the pass below would decode 64 KiB
    // of it, find that the one segment goes on, and come to the same place)
    bool tried_pieces = false;
    if (ncand == 0 && in_bytes >= (1u << 20)) {
        tried_pieces = true;
        const int src = inflate_spec_run(e, d_in, static_cast<const uint8_t *>(in), in_bytes, engine_stage_out(e), out_cap, res, st, stream_mode);
        if (src != 1) {
            if (src != ZGPU_OK) return src;
            if (out && res->out_bytes) ZGPU_HIP_CHECK(hipMemcpy(out, engine_stage_out(e), res->out_bytes, hipMemcpyDeviceToHost));
            if (offsets_out) *offsets_out = b;
            return ZGPU_OK;
        }
    }
    // stream mode: input that ends with a flush marker may simply be all there is so far -- then the last segment is a segment like the
    // others and none of them has to hold the final block
    const uint8_t *hin = static_cast<const uint8_t *>(in);
    bool open_end = stream_mode && in_bytes >= 4 && hin[in_bytes - 4] == 0 && hin[in_bytes - 3] == 0 && hin[in_bytes - 2] == 0xFF && hin[in_bytes - 1] == 0xFF;
    for (int pass = 0;; pass++) {
        const uint64_t nseg = b.size() - 1;
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_offs, b.data(), b.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        rc = inflate_run(e, d_in, in_bytes, d_offs, nseg, 0, engine_stage_out(e), out_cap, res, st, stream_mode, b.data(), open_end);
        if (rc == ZGPU_OK) break;
        if (rc != ZGPU_DATA_ERROR) return rc;
        const bool last_bad = res->first_bad_chunk >= 0 && (uint64_t)res->first_bad_chunk + 1 == nseg;
        if (open_end && last_bad && res->error_msg == kMsgTruncated) { open_end = false; pass--; continue; } // (the marker was data: an incomplete tail after all)
        if (last_bad && res->error_msg == kMsgTruncated) return rc; // the body stops early (strict mode; stream mode reports it as incomplete)
        const bool window_kept = res->error_msg == kMsgTooFar || res->error_msg == kMsgOutput;
        if (!last_bad && !window_kept && pass < 4) { b.erase(b.begin() + res->first_bad_chunk + 1); continue; } // not a boundary after all
        whole = true;
        break;
    }
    if (whole && !tried_pieces) {
        const int src = inflate_spec_run(e, d_in, hin, in_bytes, engine_stage_out(e), out_cap, res, st, stream_mode);
        if (src != 1 && src != ZGPU_OK) return src;
        whole = src == 1;
        if (!whole) b.assign({0, in_bytes});
    }
    if (whole) {
        if (in_bytes >= (1ull << 29)) return engine_fail(e, ZGPU_DATA_ERROR, "stream too long for the one-workgroup decoder");
        b.assign({0, in_bytes});
        g_whole_done++;
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_offs, b.data(), b.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        rc = inflate_run(e, d_in, in_bytes, d_offs, 1, kWholeStream, engine_stage_out(e), out_cap, res, st, stream_mode, b.data());
        if (rc != ZGPU_OK) {
            // stream mode: input that stops inside a block is not an error, nothing of it is taken (one workgroup cannot hand a window on)
            if (stream_mode && rc == ZGPU_DATA_ERROR && res->error_msg == kMsgTruncated) { res->incomplete = 1; res->in_used = 0; res->out_bytes = 0; res->stream_end = 0; return ZGPU_OK; }
            return rc;
        }
    }
    if (out && res->out_bytes) ZGPU_HIP_CHECK(hipMemcpy(out, engine_stage_out(e), res->out_bytes, hipMemcpyDeviceToHost));
    if (offsets_out) *offsets_out = b;
    return ZGPU_OK;
}

extern "C" {
#pragma GCC visibility push(default)
const char *zgpu_inflate_message(uint32_t index) { return index < kMsgCount ? kInfMessages[index] : ""; }
uint64_t zgpu_inflate_spec_count(int which) { return which == 0 ? g_spec_done.load() : which == 1 ? g_whole_done.load() : 0; }
int zgpu_inflate_find_chunks_host(zgpu_engine *e, const void *in, uint64_t in_bytes, uint32_t chunk_size, uint64_t *offsets, uint64_t max_chunks,
                                  uint64_t *nchunks)
{
    (void)chunk_size;
    if (!offsets || !nchunks) return ZGPU_STREAM_ERROR;
    std::vector<uint64_t> b;
    zgpu_inflate_result res{};
    // the staging output is sized from the data (four times the input, more when the decode asks for it), not from the capacity of the
    // caller's table: max_chunks * 64 KiB is 15 GB for the default table of a 1 MiB body
    uint64_t cap = in_bytes * 4 + 65536;
    const uint64_t cap_max = max_chunks * (uint64_t)kChunkMax;
    if (cap > cap_max) cap = cap_max;
    int rc;
    for (;;) {
        rc = inflate_stream_host(e, in, in_bytes, 0, nullptr, cap, &res, &b);
        if (rc != ZGPU_BUF_ERROR || cap >= cap_max) break;
        cap = res.out_bytes > cap ? res.out_bytes : cap * 4;
        if (cap > cap_max) cap = cap_max;
    }
    if (rc) return rc;
    if (b.size() - 1 > max_chunks) return engine_fail(e, ZGPU_BUF_ERROR, "offset table too small");
    for (size_t i = 0; i < b.size(); i++) offsets[i] = b[i];
    *nchunks = b.size() - 1;
    return ZGPU_OK;
}
int zgpu_inflate_stream_host(zgpu_engine *e, const void *in, uint64_t in_bytes, void *out, uint64_t out_cap, zgpu_inflate_result *res)
{
    if (!out) return ZGPU_STREAM_ERROR;
    return inflate_stream_host(e, in, in_bytes, 0, out, out_cap, res, nullptr);
}
int zgpu_inflate_stream_host2(zgpu_engine *e, const void *in, uint64_t in_bytes, uint32_t flags, void *out, uint64_t out_cap, zgpu_inflate_result *res)
{
    if (!out) return ZGPU_STREAM_ERROR;
    return inflate_stream_host(e, in, in_bytes, flags, out, out_cap, res, nullptr);
}
int zgpu_inflate_stream_host3(zgpu_engine *e, const void *in, uint64_t in_bytes, uint32_t start_bit, uint32_t flags, void *out, uint64_t out_cap, zgpu_inflate_result *res)
{
    if (!out) return ZGPU_STREAM_ERROR;
    return inflate_stream_host(e, in, in_bytes, flags, out, out_cap, res, nullptr, start_bit);
}
#pragma GCC visibility pop
}
