// zgpu_cont.hip -- ONE continuous deflate stream of any size on the device: what plain compress2() / deflate(Z_NO_FLUSH ... Z_FINISH) of the reference emits
// (round 4).  The reference slides a 32 KiB window through the whole input (qcsrc/deflate.c:1266-1358), its matches cross every 64 KiB
// boundary (deflate.c:1027-1168), its lazy parse is one serial chain (deflate.c:1554-1674), blocks are cut every 16383 tokens counted from the start of the
// stream (h/deflate.h:313), a block may be stored only while its first byte is still in the window (deflate.c:1364-1367) and blocks follow each other bit
// by bit (trees.c:217-229).  None of that splits into independent 64 KiB chunks; all of it splits into TILES plus four small serial chains:
//
//   tile i   = the 64 KiB of input that start at 32512 * i: it parses its local positions [h0, h1) = [32512, 65024) with the 32512 bytes in front as
//              history (MAX_DIST = 32506) and 512 bytes behind for the lazy game and the lookahead of its last positions (tests/tools/cont_tile_model.c
//              is this decomposition on the CPU, byte-identical with the reference's loop).  sort3_kernel and walk_kernel<2> (zgpu_lz_sorted.hip) work
//              on a tile as they work on a chunk; every quantity the reference keeps in window coordinates is a function of the absolute position here.
//   chain 1  where the parse ENTERS each tile: a tile's walkers start from every possible entry (513 of them), tile_exits() turns their games into the
//              tile's exit as a function of its entry, and the entries are the composition of those functions along the tiles: chain_*_kernel, groups of
//              256 tiles composed side by side, one short serial chain over the groups.
//   chain 2  which tokens make a BLOCK: a prefix sum of the tiles' token counts; the tokens are copied into one array (cont_compact_kernel), block b is
//              tokens [16383 b, 16383 (b + 1)) of it, and the tile that holds a block's last token says where the block ends in the input.
//   chain 3  where each block STARTS in the output, a bit position: a block's size depends on the phase it starts at only when it is stored (the padding
//              in front of LEN), so the scan runs over functions of the phase (cont_bitscan_kernel).  huffman_kernel<true> codes every block from bit 0 of
//              a slot of its own; cont_stitch_kernel shifts it into place (stored blocks: straight from the input).
//   chain 4  from one batch of tiles to the next, and from one feed of a stream to the next: ContState + the tokens of the block that is still filling.
#include "zgpu_common.h"
#include "../../include/zamd_gpu.h"

namespace zgpu {

// ------------------------------------------------------------------------------------------------- chain 1: entries
constexpr uint32_t kChainGroup = 256;

__global__ void __launch_bounds__(576) chain_compose_kernel(const uint16_t *__restrict__ exits, uint32_t ntiles, uint16_t *__restrict__ comp)
{
    const uint32_t g = blockIdx.x, k = threadIdx.x, t0 = g * kChainGroup, t1 = t0 + kChainGroup < ntiles ? t0 + kChainGroup : ntiles;
    uint32_t v = k < kTileEntries ? k : 0;
    for (uint32_t t = t0; t < t1; t++) v = exits[(size_t)t * kTileExitStride + v];
    if (k < kTileExitStride) comp[(size_t)g * kTileExitStride + k] = (uint16_t)v;
}
__global__ void __launch_bounds__(64) chain_serial_kernel(const uint16_t *__restrict__ comp, uint32_t ngroups, uint16_t *__restrict__ gentry, const uint16_t *__restrict__ first)
{
    if (threadIdx.x != 0) return;
    uint32_t v = *first;
    for (uint32_t g = 0; g < ngroups; g++) { gentry[g] = (uint16_t)v; if (g + 1 < ngroups) v = comp[(size_t)g * kTileExitStride + v]; }
}
// entry[t] for the tiles of the launch; entry[ntiles] = the entry of the tile behind them (the next batch's first, or where the feed's parse ends)
__global__ void __launch_bounds__(64) chain_apply_kernel(const uint16_t *__restrict__ exits, uint32_t ntiles, const uint16_t *__restrict__ gentry, uint16_t *__restrict__ entry)
{
    if (threadIdx.x != 0) return;
    const uint32_t g = blockIdx.x, t0 = g * kChainGroup, t1 = t0 + kChainGroup < ntiles ? t0 + kChainGroup : ntiles;
    uint32_t v = gentry[g];
    for (uint32_t t = t0; t < t1; t++) { entry[t] = (uint16_t)v; v = exits[(size_t)t * kTileExitStride + v]; }
    if (t1 == ntiles) entry[ntiles] = (uint16_t)v;
}

// ------------------------------------------------------------------------------------------------- chain 2: tokens -> blocks
// exclusive scan of the tiles' token counts behind the carried tokens (which become the front of the compact array)
__global__ void __launch_bounds__(1024) cont_tokscan_kernel(const ChunkMeta *__restrict__ tmeta, uint32_t ntiles, ContState *st, uint32_t *__restrict__ tokoff,
                                                            const uint32_t *__restrict__ carry, uint32_t *__restrict__ T)
{
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, per = (ntiles + 1023) / 1024;
    const uint32_t a = tid * per < ntiles ? tid * per : ntiles, z = (tid + 1) * per < ntiles ? (tid + 1) * per : ntiles;
    uint32_t sum = 0;
    for (uint32_t i = a; i < z; i++) sum += tmeta[i].ntok;
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    const uint32_t cn = st->carry_n;
    uint32_t o = cn + part[tid] - sum;
    for (uint32_t i = a; i < z; i++) { tokoff[i] = o; o += tmeta[i].ntok; }
    for (uint32_t j = tid; j < cn; j += 1024) T[j] = carry[j];
    if (tid == 1023) { tokoff[ntiles] = cn + part[1023]; st->total = cn + part[1023]; }
}

// a tile's tokens to their place in the compact array; for every block whose LAST token lies in this tile: where the block ends, how long that token is
constexpr uint32_t kMaxCuts = 6; // ceil(65536 / 16383) + 1
__global__ void __launch_bounds__(256) cont_compact_kernel(ChunkGeom g, TileGeom tg, const uint32_t *__restrict__ tokens, const ChunkMeta *__restrict__ tmeta,
                                                           const uint32_t *__restrict__ tokoff, uint32_t *__restrict__ T, ContBlk *__restrict__ blk)
{
    __shared__ unsigned long long sums[kMaxCuts];
    __shared__ uint32_t lastlen[kMaxCuts];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, n = tmeta[c].ntok, off = tokoff[c];
    const uint32_t *tok = tokens + (size_t)c * kChunkMax;
    uint32_t ncut = 0, istar[kMaxCuts];
    for (uint32_t B = (off / kBlockTokens + 1) * kBlockTokens; B <= off + n && ncut < kMaxCuts; B += kBlockTokens) istar[ncut++] = B - 1 - off;
    if (tid < kMaxCuts) { sums[tid] = 0; lastlen[tid] = 0; }
    __syncthreads();
    unsigned long long s[kMaxCuts] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j = tid; j < n; j += 256) {
        const uint32_t t = tok[j], len = (t >> 8) ? (t & 255u) + kMinMatch : 1u;
        T[off + j] = t;
        for (uint32_t q = 0; q < ncut; q++) { if (j <= istar[q]) s[q] += len; if (j == istar[q]) lastlen[q] = len; }
    }
    if (ncut == 0) return; // (uniform)
    for (uint32_t q = 0; q < ncut; q++) {
        unsigned long long v = s[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((tid & 63) == 0 && v) atomicAdd(&sums[q], v);
    }
    __syncthreads();
    if (tid < ncut) {
        uint64_t wb; uint32_t nloc, h0, h1, ne;
        tile_span(g, tg, c, wb, nloc, h0, h1, ne);
        const uint64_t e_abs = tg.abs0 + wb + h0 + tg.entry[g.chunk0 + c];
        ContBlk &b = blk[(off + istar[tid] + 1) / kBlockTokens - 1];
        b.end_pos = e_abs + sums[tid]; b.last_len = lastlen[tid];
    }
}

// slides of the reference's window that have happened when its loop stands at stream position ptop (deflate.c:1293: strstart >= wsize + MAX_DIST inside
// fill_window, which is called while lookahead < MIN_LOOKAHEAD): slide k at the first loop top >= 32768 (k - 1) + 65275 -- or one position earlier when
// that position, sp, lies in the last 261 bytes in front of a segment's end (less than MIN_LOOKAHEAD left: fill_window is called at every loop top there).
__host__ __device__ inline uint64_t cont_slides(uint64_t ptop, uint64_t sp)
{
    return (ptop >= 65275u ? (ptop - 65275u) / 32768u + 1u : 0u) + (ptop == sp ? 1u : 0u);
}
// sp for a segment that ends at stream position n: the position 32768 k + 65274 in [n - 261, n], ~0 if there is none
__host__ __device__ inline uint64_t cont_special(uint64_t n)
{
    if (n < 65274u) return ~0ull;
    const uint64_t sp = (n - 65274u) / 32768u * 32768u + 65274u;
    return sp + 261u >= n ? sp : ~0ull;
}

// the batch's blocks.  seg_end: the stream position the segment ends at when this batch reaches it (the block that is filling is closed there), ~0: more follows
__global__ void __launch_bounds__(256) cont_table_kernel(ContBlk *__restrict__ blk, ContState *st, const uint32_t *__restrict__ T, uint64_t seg_end, uint32_t final_block,
                                                         uint64_t sp, uint32_t nblk_cap, uint32_t slow)
{
    const uint32_t total = st->total;
    uint32_t nblk = total / kBlockTokens;
    if (seg_end != ~0ull) {
        // deflate_slow tallies the last byte's pending literal BEHIND its loop without looking at "buffer full" (deflate.c:1660-1665): when that literal fills the
        // block, the full block is the segment's last; any other token that fills a block is followed by one more (possibly empty) block at the flush
        // (deflate_fast has no such literal: its last token is tallied inside the loop, deflate.c:1524-1541)
        const bool full_last = slow && total != 0 && total % kBlockTokens == 0 && (T[total - 1] >> 8) == 0;
        if (!full_last) nblk++;
    }
    if (nblk > nblk_cap) nblk = nblk_cap; // (cannot happen: the capacity is computed from the positions of the batch)
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b == 0) st->nblk = nblk;
    if (b >= nblk) return;
    ContBlk &k = blk[b];
    k.tok0 = b * kBlockTokens;
    const bool complete = (uint64_t)(b + 1) * kBlockTokens <= total;
    k.nt = complete ? kBlockTokens : total - k.tok0;
    if (!complete) { k.end_pos = seg_end; k.last_len = 1; }          // the flush at the segment's end: the loop stands at the end itself
    else if (seg_end != ~0ull && b + 1 == nblk) k.last_len = 1;       // (the full last block: flushed behind the loop as well; its end_pos IS seg_end)
    k.start_pos = b == 0 ? st->block_start : blk[b - 1].end_pos;
    k.eof = final_block && b + 1 == nblk;
    k.first = st->first_block && b == 0;
    // where the loop stood when the block was flushed: deflate_slow emits a token one iteration late (the match of position m while it stands at m + 1,
    // deflate.c:1617-1640; a literal likewise, :1644-1656), deflate_fast in the iteration of the token's own position (:1499-1541); at a segment's end both
    // flush behind the loop, which stands at the end then (last_len is 1 there and end_pos the end: the slow formula)
    const uint64_t ptop = k.end_pos - k.last_len + ((slow || !complete) ? 1 : 0);
    k.nostore = k.start_pos < 32768u * cont_slides(ptop, sp);
    k.nbits = 0; k.btype = 0; k.stored_len = 0; k.eob_len = 8;
}

// ------------------------------------------------------------------------------------------------- chain 3: bit positions
__device__ inline uint64_t blk_advance(uint64_t pos, const ContBlk &k)
{
    return k.btype ? pos + k.nbits : ((pos + 3 + 7) & ~7ull) + 32 + 8ull * k.stored_len; // trees.c:867-879: 3 header bits, bi_windup, LEN, NLEN, the bytes
}
__global__ void __launch_bounds__(1024) cont_bitscan_kernel(const ContBlk *__restrict__ blk, ContState *st, uint64_t *__restrict__ pos, uint64_t out_cap_bits)
{
    __shared__ uint32_t adv[1024][8];
    __shared__ uint64_t segpos[1025];
    const uint32_t tid = threadIdx.x, nblk = st->nblk, per = (nblk + 1023) / 1024;
    const uint32_t a = tid * per < nblk ? tid * per : nblk, z = (tid + 1) * per < nblk ? (tid + 1) * per : nblk;
    for (uint32_t ph = 0; ph < 8; ph++) {
        uint64_t p = ph;
        for (uint32_t i = a; i < z; i++) p = blk_advance(p, blk[i]);
        adv[tid][ph] = (uint32_t)(p - ph);
    }
    __syncthreads();
    if (tid == 0) {
        uint64_t p = st->out_bits;
        for (uint32_t s = 0; s < 1024; s++) { segpos[s] = p; p += adv[s][p & 7]; }
        segpos[1024] = p;
        st->out_bits = p;
        pos[nblk] = p;
        if (p > out_cap_bits) st->overflow = 1;
    }
    __syncthreads();
    uint64_t p = segpos[tid];
    for (uint32_t i = a; i < z; i++) { pos[i] = p; p = blk_advance(p, blk[i]); }
}

// A word of the output belongs to one block alone when all its 32 bits do: such words are stored.  A word that holds a block boundary is ORed into by
// every block that has bits in it, and is cleared here first -- except the word the batch starts in when earlier bits (the batch before, the stream's
// header, the bits a flush left over) are in it already.
__device__ inline bool word_is_inner(uint64_t w, uint64_t D, uint64_t E) { return 32 * w >= D && 32 * w + 32 <= E; }
__global__ void __launch_bounds__(256) cont_edges_kernel(const ContState *st, const uint64_t *__restrict__ pos, uint32_t *__restrict__ out, uint64_t out_words)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x, nblk = st->nblk;
    if (b >= nblk) return;
    const uint64_t D = pos[b], E = pos[b + 1];
    if (E == D) return;
    const uint64_t wf = D >> 5, wl = (E - 1) >> 5, keep = (pos[0] & 31) ? pos[0] >> 5 : ~0ull; // the word the batch starts in, when it is not empty
    if (!word_is_inner(wf, D, E) && wf != keep && wf < out_words) out[wf] = 0;
    if (wl != wf && !word_is_inner(wl, D, E) && wl < out_words) out[wl] = 0;
}
__global__ void __launch_bounds__(256) cont_stitch_kernel(const ContBlk *__restrict__ blk, const ContState *st, const uint64_t *__restrict__ pos, const uint8_t *__restrict__ slots,
                                                          uint32_t slot_stride, const uint8_t *__restrict__ in, uint64_t abs0, uint32_t *__restrict__ out, uint64_t out_words)
{
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (b >= st->nblk) return;
    const ContBlk &k = blk[b];
    const uint64_t D = pos[b], E = pos[b + 1];
    if (E == D) return;
    const uint64_t wf = D >> 5, wl = (E - 1) >> 5;
    if (k.btype) { // a coded block: its bits, from bit 0 of the slot, shifted to D
        const uint32_t *S = reinterpret_cast<const uint32_t *>(slots + (size_t)b * slot_stride);
        const uint32_t nsw = (k.nbits + 31) >> 5;
        for (uint64_t w = wf + tid; w <= wl; w += 256) {
            if (w >= out_words) break;
            uint32_t v;
            if (32 * w < D) v = S[0] << (uint32_t)(D - 32 * w);
            else {
                const uint64_t o = 32 * w - D;
                const uint32_t sw = (uint32_t)(o >> 5), sh = (uint32_t)(o & 31), lo = S[sw], hi = sw + 1 < nsw ? S[sw + 1] : 0u;
                v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
            }
            if (32 * w + 32 > E) v &= (1u << (uint32_t)(E - 32 * w)) - 1u;
            if (word_is_inner(w, D, E)) out[w] = v; else atomicOr(&out[w], v);
        }
    } else { // a stored block (trees.c:867-879, 1197-1219): three header bits at D, zeros up to the byte boundary, LEN, ~LEN, the bytes of the input
        const uint32_t ph = (uint32_t)(D & 7), hb = (ph + 3 + 7) >> 3, len = k.stored_len, hdr = k.eof << ph;
        const uint64_t db0 = D >> 3, nb = hb + 4 + (uint64_t)len;
        const uint8_t *src = in + (k.start_pos - abs0);
        for (uint64_t w = wf + tid; w <= wl; w += 256) {
            if (w >= out_words) break;
            uint32_t v = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint64_t a = 4 * w + j;
                if (a < db0 || a - db0 >= nb) continue;
                const uint64_t i = a - db0;
                uint32_t byte;
                if (i < hb) byte = (hdr >> (8 * (uint32_t)i)) & 255u;
                else if (i < hb + 4) { const uint32_t f = (uint32_t)(i - hb), l16 = f < 2 ? len : ~len; byte = (l16 >> (8 * (f & 1))) & 255u; }
                else byte = src[i - hb - 4];
                v |= byte << (8 * j);
            }
            if (word_is_inner(w, D, E)) out[w] = v; else atomicOr(&out[w], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------- chain 4: to the next batch
// the tokens behind the last emitted block wait in the carry buffer; the state moves on.  seg_end as in cont_table_kernel.
__global__ void __launch_bounds__(1024) cont_carry_kernel(const ContBlk *__restrict__ blk, ContState *st, const uint32_t *__restrict__ T, uint32_t *__restrict__ carry, uint64_t seg_end)
{
    const uint32_t tid = threadIdx.x, nblk = st->nblk, total = st->total, old_carry = st->carry_n;
    const uint32_t used = seg_end != ~0ull ? total : nblk * kBlockTokens, rem = total - used;
    for (uint32_t j = tid; j < rem; j += 1024) carry[j] = T[used + j];
    __syncthreads();
    if (tid == 0) {
        st->ntokens += total - old_carry;
        st->carry_n = rem;
        if (nblk) { st->block_start = blk[nblk - 1].end_pos; st->last_eob = blk[nblk - 1].eob_len; st->first_block = 0; }
        if (seg_end != ~0ull) st->block_start = seg_end;
    }
}

// ------------------------------------------------------------------------------------------------- launches
void launch_chain(const uint16_t *exits, uint32_t ntiles, uint16_t *comp, uint16_t *gentry, uint16_t *entry, hipStream_t st)
{
    const uint32_t ngroups = (ntiles + kChainGroup - 1) / kChainGroup;
    if (ngroups > 1) hipLaunchKernelGGL(chain_compose_kernel, dim3(ngroups), dim3(576), 0, st, exits, ntiles, comp);
    hipLaunchKernelGGL(chain_serial_kernel, dim3(1), dim3(64), 0, st, comp, ngroups, gentry, entry);
    hipLaunchKernelGGL(chain_apply_kernel, dim3(ngroups), dim3(64), 0, st, exits, ntiles, gentry, entry);
}
uint32_t chain_groups(uint32_t ntiles) { return (ntiles + kChainGroup - 1) / kChainGroup; }

void launch_cont_tokens(const ChunkGeom &g, const TileGeom &tg, const uint32_t *tokens, const ChunkMeta *tmeta, ContState *st, uint32_t *tokoff, const uint32_t *carry, uint32_t *T,
                        ContBlk *blk, uint64_t seg_end, bool final_block, uint64_t sp, uint32_t nblk_cap, bool slow, hipStream_t s)
{
    hipLaunchKernelGGL(cont_tokscan_kernel, dim3(1), dim3(1024), 0, s, tmeta, g.nchunks, st, tokoff, carry, T);
    if (g.nchunks) hipLaunchKernelGGL(cont_compact_kernel, dim3(g.nchunks), dim3(256), 0, s, g, tg, tokens, tmeta, tokoff, T, blk);
    hipLaunchKernelGGL(cont_table_kernel, dim3((nblk_cap + 255) / 256), dim3(256), 0, s, blk, st, T, seg_end, final_block ? 1u : 0u, sp, nblk_cap, slow ? 1u : 0u);
}
void launch_cont_stitch(const ContBlk *blk, ContState *st, uint64_t *pos, const uint8_t *slots, uint32_t slot_stride, const uint8_t *in, uint64_t abs0, uint8_t *out, uint64_t out_cap,
                        uint32_t nblk_cap, const uint32_t *T, uint32_t *carry, uint64_t seg_end, hipStream_t s)
{
    const uint64_t out_words = out_cap >> 2;
    hipLaunchKernelGGL(cont_bitscan_kernel, dim3(1), dim3(1024), 0, s, blk, st, pos, out_words * 32);
    hipLaunchKernelGGL(cont_edges_kernel, dim3((nblk_cap + 255) / 256), dim3(256), 0, s, st, pos, reinterpret_cast<uint32_t *>(out), out_words);
    hipLaunchKernelGGL(cont_stitch_kernel, dim3(nblk_cap), dim3(256), 0, s, blk, st, pos, slots, slot_stride, in, abs0, reinterpret_cast<uint32_t *>(out), out_words);
    hipLaunchKernelGGL(cont_carry_kernel, dim3(1), dim3(1024), 0, s, blk, st, T, carry, seg_end);
}
uint64_t cont_special_pos(uint64_t n) { return cont_special(n); }

} // namespace zgpu
