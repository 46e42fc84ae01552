// zgpu_stitch.hip -- everything between per-chunk segments and one stream: Adler-32 per chunk
// (/root/reference/qcsrc/adler32.c:57-125 as a parallel reduction), ordered combination of the chunk checksums
// (adler32.c:128-149 as an associative monoid), exclusive scan of segment sizes, and the copy of every
// segment to its final byte offset ("stitching").  Segments end byte-aligned (flush marker / bi_windup), so
// stitching is a byte copy, never a bit shift.  Also hosts the corpus generator kernel used by bench.py.
#include "zgpu_common.h"
#include "corpus.h"

namespace zgpu {

constexpr uint32_t kAdlerBase = 65521;

// running totals carried across batches of one call (device memory)
struct RunState {
    uint64_t out_total; // bytes placed so far (starts at 2 when a zlib header is prepended)
    uint64_t in_total;
    uint64_t ntokens;
    uint32_t adler_a, adler_b; // Adler-32 halves of all input so far (a starts at 1, b at 0)
    uint32_t data_type;
    uint32_t overflow;         // set when the output capacity was exceeded
    uint32_t crc, pad;         // CRC-32 of all input so far (meaningful when the chunks' crc fields were filled)
};

// ---- Adler-32 of each chunk: A = 1 + sum b_i, B = n + sum (n - i) b_i (mod 65521) ----
__global__ void __launch_bounds__(256) adler_kernel(ChunkGeom g, ChunkMeta *meta)
{
    __shared__ uint64_t red2[4];
    __shared__ uint32_t red1[4];
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    if (c >= g.nchunks) return;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint32_t skip = chunk_skip(g, c); // a preset dictionary in front of the data is not part of the checksum
    const uint8_t *src = g.in + lo + skip;
    n -= skip;
    uint32_t s1 = 0; uint64_t s2 = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *v = reinterpret_cast<const uint4 *>(src);
        const uint32_t nvec = n >> 4;
        for (uint32_t i = tid; i < nvec; i += 256) {
            uint4 q = v[i];
            uint32_t w[4] = {q.x, q.y, q.z, q.w}, o = i << 4, t1 = 0, t2 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) { uint32_t b = (w[k] >> (8 * j)) & 255; t1 += b; t2 += b * (uint32_t)(k * 4 + j); }
            }
            s1 += t1; s2 += (uint64_t)(n - o) * t1 - t2;
        }
        for (uint32_t i = (nvec << 4) + tid; i < n; i += 256) { uint32_t b = src[i]; s1 += b; s2 += (uint64_t)(n - i) * b; }
    } else {
        for (uint32_t i = tid; i < n; i += 256) { uint32_t b = src[i]; s1 += b; s2 += (uint64_t)(n - i) * b; }
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
    if ((tid & 63) == 0) { red1[tid >> 6] = s1; red2[tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        uint64_t a = 1 + (uint64_t)red1[0] + red1[1] + red1[2] + red1[3];
        uint64_t b = n + red2[0] % kAdlerBase + red2[1] % kAdlerBase + red2[2] % kAdlerBase + red2[3] % kAdlerBase;
        meta[c].adler_a = (uint32_t)(a % kAdlerBase); meta[c].adler_b = (uint32_t)(b % kAdlerBase); meta[c].in_bytes = n;
    }
}


// ---- CRC-32 (the reference's crc32.c:219-335 and crc32_combine :370-423, as polynomial arithmetic) ----
// Reflected CRC-32 polynomial; a 32-bit word holds a polynomial over GF(2) with x^0 in bit 31.
constexpr uint32_t kCrcPoly = 0xedb88320u;
__host__ __device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b) // a(x) * b(x) mod P(x)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
__host__ __device__ inline uint32_t crc_xpow8n(uint64_t n) // x^(8n) mod P: the operator "append n zero bytes"
{
    uint32_t r = 0x80000000u, sq = 0x00800000u; // x^0, x^8
    while (n) {
        if (n & 1) r = crc_mulmod(sq, r);
        sq = crc_mulmod(sq, sq);
        n >>= 1;
    }
    return r;
}
// CRC of X||Y from the finished CRCs of X and Y and |Y| (what crc32_combine computes with its 32x32 bit matrices)
__host__ __device__ inline uint32_t crc_join(uint32_t cx, uint32_t cy, uint32_t op_leny) { return crc_mulmod(op_leny, cx) ^ cy; }

// One 256-lane workgroup per chunk.  Lane t takes a slice of L = ceil(n/256) bytes, slices aligned to the END of the chunk
// (so every right-hand operand of the tree below has a full power-of-two number of slices and one operator per level
// serves all lanes; the short or empty slices are on the left, where length does not matter), table-driven bytewise
// (crc32.c:242-266 without the 4-byte variant: the table lives in LDS), then an 8-level tree of joins.
__global__ void __launch_bounds__(256) crc_kernel(ChunkGeom g, ChunkMeta *meta)
{
    __shared__ uint32_t table[4][256]; // table[k][b]: the CRC register after byte b and k more zero bytes (slicing by four, crc32.c:268-290)
    __shared__ uint32_t part[256];
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    if (c >= g.nchunks) return;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    const uint32_t skip = chunk_skip(g, c);
    const uint8_t *src = g.in + lo + skip;
    n -= skip;
    {
        uint32_t r = tid;
#pragma unroll
        for (int k = 0; k < 8; k++) r = (r & 1u) ? (r >> 1) ^ kCrcPoly : r >> 1;
        table[0][tid] = r;
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const uint32_t r = table[k - 1][tid];
        table[k][tid] = (r >> 8) ^ table[0][r & 255u];
        __syncthreads();
    }
    const uint32_t L = (n + 255) / 256;
    const int64_t hi = (int64_t)n - (int64_t)(255 - tid) * L, lo_b = hi - L; // [lo_b, hi) clipped to [0, n)
    uint32_t crc = 0;
    if (n == 65536 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // a full, aligned chunk: the lane's 256 bytes come in as sixteen 16-byte loads issued together, four bytes per step
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src + (size_t)tid * 256);
        uint4 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = s4[k];
        crc = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t x = crc ^ w[j];
                crc = table[3][x & 255u] ^ table[2][(x >> 8) & 255u] ^ table[1][(x >> 16) & 255u] ^ table[0][x >> 24];
            }
        }
        crc ^= 0xffffffffu;
    } else if (hi > 0) {
        crc = 0xffffffffu;
        for (int64_t i = lo_b > 0 ? lo_b : 0; i < hi; i++) crc = table[0][(crc ^ src[i]) & 255u] ^ (crc >> 8);
        crc ^= 0xffffffffu;
    }
    part[tid] = crc;
    uint32_t op = crc_xpow8n(L); // append L bytes
    __syncthreads();
    for (uint32_t s = 1; s < 256; s <<= 1) {
        uint32_t v = 0;
        const bool mine = (tid & (2 * s - 1)) == 0;
        if (mine) v = crc_join(part[tid], part[tid + s], op); // the right operand covers exactly s slices: s*L bytes
        __syncthreads();
        if (mine) part[tid] = v;
        op = crc_mulmod(op, op);
        __syncthreads();
    }
    if (tid == 0) { meta[c].crc = part[0]; meta[c].in_bytes = n; }
}

// Adler of X||Y from Adler(X) = (ax, bx), Adler(Y) = (ay, by), |Y| = ny:  a = ax + ay - 1,  b = bx + by + ny (ax - 1)
__device__ inline void adler_join(uint32_t &ax, uint32_t &bx, uint32_t ay, uint32_t by, uint64_t ny)
{
    uint64_t rem = ny % kAdlerBase;
    uint64_t a = ((uint64_t)ax + ay + kAdlerBase - 1) % kAdlerBase;
    uint64_t b = ((uint64_t)bx + by + rem * ((ax + kAdlerBase - 1) % kAdlerBase)) % kAdlerBase;
    ax = (uint32_t)a; bx = (uint32_t)b;
}

// One workgroup: exclusive scan of out_bytes over the batch (continuing RunState), ordered Adler combination.
__global__ void __launch_bounds__(1024) scan_kernel(const ChunkMeta *__restrict__ meta, uint32_t nchunks, uint64_t chunk0, uint64_t *offsets,
                                                    RunState *run, uint64_t out_cap, uint32_t with_crc)
{
    __shared__ uint64_t part[1024];
    __shared__ uint32_t pa[1024], pb[1024];
    __shared__ uint64_t plen[1024], ptok[1024];
    __shared__ uint32_t pcrc[1024];
    const uint32_t tid = threadIdx.x, per = (nchunks + 1023) / 1024;
    const uint32_t a = tid * per < nchunks ? tid * per : nchunks, z = (tid + 1) * per < nchunks ? (tid + 1) * per : nchunks;
    uint64_t sum = 0, len = 0, ntok = 0; uint32_t xa = 1, xb = 0, xc = 0;
    uint32_t op_len = ~0u, op = 0; // the append operator of the last chunk length seen (chunks are the same size but for the last)
    for (uint32_t i = a; i < z; i++) {
        sum += meta[i].out_bytes; ntok += meta[i].ntok;
        adler_join(xa, xb, meta[i].adler_a, meta[i].adler_b, meta[i].in_bytes); len += meta[i].in_bytes;
        if (with_crc) {
            if (meta[i].in_bytes != op_len) { op_len = meta[i].in_bytes; op = crc_xpow8n(op_len); }
            xc = crc_join(xc, meta[i].crc, op);
        }
    }
    part[tid] = sum; pa[tid] = xa; pb[tid] = xb; plen[tid] = len; ptok[tid] = ntok; pcrc[tid] = xc;
    __syncthreads();
    // The 1024 partial results are put together by all lanes (one lane walking them -- two 64-bit divisions and a CRC product each -- was 0.39 ms of
    // every call, a third of a small call): an inclusive scan of the sizes, and for the checksums, which are associative but not commutative, a
    // tree of joins of NEIGHBOURING ranges (lane t takes in the range that starts at t + d).
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    const uint64_t incl = part[tid];
    // (the operator "append the right-hand range" costs ~50 products: ranges of the usual length -- 2^k times lane 0's -- share one, squared per level)
    const uint64_t len0 = plen[0];
    __syncthreads(); // (the tree below overwrites plen[0])
    uint32_t op_level = with_crc ? crc_xpow8n(len0) : 0;
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        if ((tid & (2 * d - 1)) == 0) {
            const uint64_t rl = plen[tid + d];
            uint32_t a2 = pa[tid], b2 = pb[tid];
            adler_join(a2, b2, pa[tid + d], pb[tid + d], rl);
            pa[tid] = a2; pb[tid] = b2;
            if (with_crc) pcrc[tid] = crc_join(pcrc[tid], pcrc[tid + d], rl == len0 * d ? op_level : crc_xpow8n(rl));
            plen[tid] += rl; ptok[tid] += ptok[tid + d];
        }
        if (with_crc) op_level = crc_mulmod(op_level, op_level);
        __syncthreads();
    }
    const uint64_t base = run->out_total;
    __syncthreads(); // (every lane has read the total before lane 0 moves it on)
    if (tid == 0) {
        uint32_t ra = run->adler_a, rb = run->adler_b, rc = run->crc;
        adler_join(ra, rb, pa[0], pb[0], plen[0]);
        if (with_crc) rc = crc_join(rc, pcrc[0], crc_xpow8n(plen[0]));
        const uint64_t acc = base + part[1023];
        run->out_total = acc; run->adler_a = ra; run->adler_b = rb; run->crc = rc; run->in_total += plen[0]; run->ntokens += ptok[0];
        if (chunk0 == 0 && nchunks > 0) run->data_type = meta[0].data_type;
        if (acc > out_cap) run->overflow = 1;
        offsets[chunk0 + nchunks] = acc;
    }
    uint64_t o = base + incl - sum;
    for (uint32_t i = a; i < z; i++) { offsets[chunk0 + i] = o; o += meta[i].out_bytes; }
}

// copy slot c to out + offsets[chunk0 + c]; 4-byte destination-aligned stores
__global__ void __launch_bounds__(256) stitch_kernel(const uint8_t *__restrict__ slots, const ChunkMeta *__restrict__ meta,
                                                     const uint64_t *__restrict__ offsets, uint64_t chunk0, uint32_t nchunks, uint8_t *out,
                                                     uint64_t out_cap, uint32_t slot_stride)
{
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    if (c >= nchunks) return;
    const uint32_t n = meta[c].out_bytes;
    const uint64_t off = offsets[chunk0 + c];
    if (off + n > out_cap) return; // reported through RunState.overflow
    const uint8_t *src = slots + (size_t)c * slot_stride;
    uint8_t *dst = out + off;
    uint32_t head = (uint32_t)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);
    if (head > n) head = n;
    if (tid < head) dst[tid] = src[tid];
    const uint32_t nwords = (n - head) >> 2;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + head);
    const uint32_t sh = head & 3, w0 = head >> 2;
    for (uint32_t j = tid; j < nwords; j += 256) {
        uint32_t lo = s32[w0 + j], hi = s32[w0 + j + 1]; // slot has slack, reading one word past is in bounds
        d32[j] = __builtin_amdgcn_alignbyte(hi, lo, sh);
    }
    const uint32_t done = head + (nwords << 2);
    if (tid < n - done) dst[done + tid] = src[done + tid];
}

__global__ void __launch_bounds__(64) corpus_kernel(uint32_t kind, uint64_t seed, uint64_t first_chunk, uint64_t nchunks, uint8_t *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) zc_fill_chunk(kind, seed, first_chunk + i, out + i * ZC_CHUNK);
}

void launch_crc(const ChunkGeom &g, ChunkMeta *meta, hipStream_t st)
{
    hipLaunchKernelGGL(crc_kernel, dim3(g.nchunks), dim3(256), 0, st, g, meta);
}

// the batch's chunks the lane-per-chunk loop has handed on (ChunkMeta::ntok == kHandedOn), as a list for a launch of the wave-per-chunk kernel (any order)
__global__ void __launch_bounds__(256) collect_handed_on_kernel(const ChunkMeta *__restrict__ meta, uint32_t n, uint32_t *__restrict__ list, uint32_t *__restrict__ count)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c < n && meta[c].ntok == kHandedOn) list[atomicAdd(count, 1u)] = c;
}
void launch_collect_handed_on(const ChunkMeta *meta, uint32_t n, uint32_t *list, uint32_t *count, hipStream_t st)
{
    hipMemsetAsync(count, 0, 4, st);
    hipLaunchKernelGGL(collect_handed_on_kernel, dim3((n + 255) / 256), dim3(256), 0, st, meta, n, list, count);
}

void launch_adler(const ChunkGeom &g, ChunkMeta *meta, hipStream_t st)
{
    hipLaunchKernelGGL(adler_kernel, dim3(g.nchunks), dim3(256), 0, st, g, meta);
}
void launch_scan(const ChunkMeta *meta, uint32_t nchunks, uint64_t chunk0, uint64_t *offsets, void *run, uint64_t out_cap, hipStream_t st, bool with_crc)
{
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, meta, nchunks, chunk0, offsets, static_cast<RunState *>(run), out_cap, with_crc ? 1u : 0u);
}
void launch_stitch(const uint8_t *slots, const ChunkMeta *meta, const uint64_t *offsets, uint64_t chunk0, uint32_t nchunks, uint8_t *out,
                   uint64_t out_cap, uint32_t slot_stride, hipStream_t st)
{
    hipLaunchKernelGGL(stitch_kernel, dim3(nchunks), dim3(256), 0, st, slots, meta, offsets, chunk0, nchunks, out, out_cap, slot_stride);
}
void launch_corpus(uint32_t kind, uint64_t seed, uint64_t first_chunk, uint64_t nchunks, uint8_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(corpus_kernel, dim3((uint32_t)((nchunks + 63) / 64)), dim3(64), 0, st, kind, seed, first_chunk, nchunks, out);
}

} // namespace zgpu
