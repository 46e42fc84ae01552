/* deflate_oracle.c -- CPU restatement of the zlib-1.2.3 compress path for ONE fresh stream per chunk.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity: PINNED against oracle/_ref/libzref.so and
 * tests/golden/ by tests/test_oracle_vs_reference.py.
 *
 * What is restated (file:line under /root/reference):
 *   level table                      qcsrc/deflate.c:137-149
 *   rolling hash / INSERT_STRING     qcsrc/deflate.c:170,189-192
 *   fill_window slide                qcsrc/deflate.c:1293-1326
 *   longest_match                    qcsrc/deflate.c:1027-1168
 *   deflate_stored/_fast/_slow       qcsrc/deflate.c:1390-1439, 1448-1546, 1554-1674
 *   tally + block cut                h/deflate.h:308-324
 *   build_tree & friends             qcsrc/trees.c:434-478, 490-567, 577-609, 619-701
 *   scan_tree/send_tree/bl tree      qcsrc/trees.c:707-862
 *   _tr_flush_block / stored blocks  qcsrc/trees.c:867-879, 921-1016, 1178-1219
 *   compress_block                   qcsrc/trees.c:1072-1118
 *   static tables                    qcsrc/trees.c:238-316 (regenerated here from RFC 1951)
 *
 * Coordinates: `p` is an offset into the chunk; the reference's window index is w = p + base - off,
 * where base is 0 for a fresh stream (3 when the chunk is made "position-0 matchable" by a 3-byte
 * preset dictionary) and off is 0 before / 32768 after the one window slide a
 * 64 KiB chunk can trigger.  Hash-table entries hold window indices so that NIL (0) and the slide
 * behave exactly as in the reference.
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>

enum {
    WSIZE = 32768, WMASK = WSIZE - 1, HSIZE = 32768, HMASK = HSIZE - 1, HSHIFT = 5,
    MINM = 3, MAXM = 258, MIN_LOOK = MAXM + MINM + 1, MAXDIST = WSIZE - MIN_LOOK, FAR_LIMIT = 4096,
    LITBUF = 16384, NLIT = 286, NDIST = 30, NBL = 19, NHEAP = 2 * NLIT + 1, MAXBITS = 15,
    MAXBLBITS = 7, EOB = 256
};

typedef struct { uint16_t good, lazy, nice, chain; int mode; } level_cfg; /* mode 0 store 1 fast 2 slow */
static const level_cfg LEVELS[10] = {
    {0, 0, 0, 0, 0},       {4, 4, 8, 4, 1},        {4, 5, 16, 8, 1},      {4, 6, 32, 32, 1},
    {4, 4, 16, 16, 2},     {8, 16, 32, 32, 2},     {8, 16, 128, 128, 2},  {8, 32, 128, 256, 2},
    {32, 128, 258, 1024, 2}, {32, 258, 258, 4096, 2}};

static const uint8_t XL[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint8_t XD[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint8_t XBL[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
static const uint8_t BLORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

/* ---- static tables, regenerated (trees.c:238-316) ---- */
static uint8_t  T_lencode[256];   /* match length-3 -> length code 0..28 */
static uint8_t  T_distcode[512];  /* see d_code(), h/deflate.h:290-291 */
static uint16_t T_baselen[29], T_basedist[30];
static uint16_t T_slcode[288]; static uint8_t T_sllen[288];
static uint16_t T_sdcode[30];
static int tables_ready;

static unsigned bitrev(unsigned v, int len) { unsigned r = 0; while (len-- > 0) { r = (r << 1) | (v & 1); v >>= 1; } return r; }

static void make_tables(void)
{
    if (tables_ready) return;
    int code, n, length = 0, dist = 0;
    for (code = 0; code < 28; code++) {
        T_baselen[code] = (uint16_t)length;
        for (n = 0; n < (1 << XL[code]); n++) T_lencode[length++] = (uint8_t)code;
    }
    T_lencode[255] = 28; T_baselen[28] = 0; /* length 258 gets its own code (trees.c:268-272) */
    for (code = 0; code < 16; code++) {
        T_basedist[code] = (uint16_t)dist;
        for (n = 0; n < (1 << XD[code]); n++) T_distcode[dist++] = (uint8_t)code;
    }
    dist >>= 7;
    for (; code < NDIST; code++) {
        T_basedist[code] = (uint16_t)(dist << 7);
        for (n = 0; n < (1 << (XD[code] - 7)); n++) T_distcode[256 + dist++] = (uint8_t)code;
    }
    unsigned blc[16] = {0}, next[16], c = 0;
    for (n = 0; n < 288; n++) { T_sllen[n] = (uint8_t)(n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8); blc[T_sllen[n]]++; }
    for (n = 1; n <= 15; n++) { c = (c + blc[n - 1]) << 1; next[n] = c; }
    for (n = 0; n < 288; n++) T_slcode[n] = (uint16_t)bitrev(next[T_sllen[n]]++, T_sllen[n]);
    for (n = 0; n < NDIST; n++) T_sdcode[n] = (uint16_t)bitrev((unsigned)n, 5);
    tables_ready = 1;
}

static unsigned dcode_of(unsigned dm1) { return dm1 < 256 ? T_distcode[dm1] : T_distcode[256 + (dm1 >> 7)]; }

/* ---- bit sink: LSB-first, byte stream identical to send_bits/put_short (trees.c:217-229) ---- */
typedef struct { uint8_t *out; size_t cap, len; uint64_t acc; int nacc; int overflow; } bitsink;

static void sink_byte(bitsink *b, unsigned v) { if (b->len < b->cap) b->out[b->len++] = (uint8_t)v; else b->overflow = 1; }
static void put_bits(bitsink *b, unsigned value, int nbits)
{
    b->acc |= (uint64_t)value << b->nacc; b->nacc += nbits;
    while (b->nacc >= 8) { sink_byte(b, (unsigned)(b->acc & 0xff)); b->acc >>= 8; b->nacc -= 8; }
}
static void byte_align(bitsink *b) { if (b->nacc > 0) { sink_byte(b, (unsigned)(b->acc & 0xff)); } b->acc = 0; b->nacc = 0; } /* bi_windup trees.c:1178 */

/* ---- one Huffman tree under construction ---- */
typedef struct {
    uint32_t freq[NHEAP]; uint16_t dad[NHEAP]; uint16_t len[NHEAP]; uint16_t code[NHEAP];
    int max_code;
} htree;

typedef struct {
    const uint8_t *in; uint32_t n, base, off, start; int level, strategy; const level_cfg *cfg; /* start: bytes of preset dictionary in front of the data */
    uint16_t head[HSIZE], prev[WSIZE];
    uint32_t ntok; uint16_t dbuf[LITBUF]; uint8_t lbuf[LITBUF];
    htree lt, dt, bt;
    int heap[NHEAP], heap_len, heap_max; uint8_t depth[NHEAP]; uint16_t bl_count[MAXBITS + 1];
    uint64_t opt_len, static_len;
    uint32_t block_start; /* chunk coords */
    bitsink bs;
    ora_token *tok_out; uint32_t tok_total; ora_chunk_info *info; uint32_t nblocks; int data_type;
    int last_eob; /* last_eob_len, trees.c:1117 (compress_block), :1206 (copy_block: 8), :400 (_tr_init: 8) -- what _tr_align looks at */
} enc;

static void new_block(enc *e) /* init_block trees.c:411-424 */
{
    memset(e->lt.freq, 0, sizeof(uint32_t) * NLIT); memset(e->dt.freq, 0, sizeof(uint32_t) * NDIST);
    memset(e->bt.freq, 0, sizeof(uint32_t) * NBL);
    e->lt.freq[EOB] = 1; e->opt_len = e->static_len = 0; e->ntok = 0;
}

/* ---- heap Huffman (trees.c:434-478) ---- */
static int lighter(const enc *e, const htree *t, int a, int b) { return t->freq[a] < t->freq[b] || (t->freq[a] == t->freq[b] && e->depth[a] <= e->depth[b]); }
static void sift_down(enc *e, const htree *t, int k)
{
    int v = e->heap[k], j = k << 1;
    while (j <= e->heap_len) {
        if (j < e->heap_len && lighter(e, t, e->heap[j + 1], e->heap[j])) j++;
        if (lighter(e, t, v, e->heap[j])) break;
        e->heap[k] = e->heap[j]; k = j; j <<= 1;
    }
    e->heap[k] = v;
}

/* gen_bitlen trees.c:490-567 */
static void assign_lengths(enc *e, htree *t, int elems_max_code, const uint8_t *sllen, const uint8_t *extra, int xbase, int max_length)
{
    int h, n, m, bits, xbits, overflow = 0; (void)elems_max_code;
    for (bits = 0; bits <= MAXBITS; bits++) e->bl_count[bits] = 0;
    t->len[e->heap[e->heap_max]] = 0;
    for (h = e->heap_max + 1; h < NHEAP; h++) {
        n = e->heap[h]; bits = t->len[t->dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        t->len[n] = (uint16_t)bits;
        if (n > t->max_code) continue;
        e->bl_count[bits]++; xbits = 0;
        if (n >= xbase) xbits = extra[n - xbase];
        e->opt_len += (uint64_t)t->freq[n] * (unsigned)(bits + xbits);
        if (sllen) e->static_len += (uint64_t)t->freq[n] * (unsigned)(sllen[n] + xbits);
    }
    if (overflow == 0) return;
    do {
        bits = max_length - 1;
        while (e->bl_count[bits] == 0) bits--;
        e->bl_count[bits]--; e->bl_count[bits + 1] += 2; e->bl_count[max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    for (bits = max_length; bits != 0; bits--) {
        n = e->bl_count[bits];
        while (n != 0) {
            m = e->heap[--h];
            if (m > t->max_code) continue;
            if (t->len[m] != (unsigned)bits) {
                e->opt_len += ((uint64_t)bits - (uint64_t)t->len[m]) * (uint64_t)t->freq[m];
                t->len[m] = (uint16_t)bits;
            }
            n--;
        }
    }
}

/* build_tree trees.c:619-701 (+ gen_codes :577-609) */
static void grow_tree(enc *e, htree *t, int elems, const uint8_t *sllen, const uint8_t *extra, int xbase, int max_length)
{
    int n, m, node, max_code = -1;
    e->heap_len = 0; e->heap_max = NHEAP;
    for (n = 0; n < elems; n++) {
        if (t->freq[n] != 0) { e->heap[++e->heap_len] = max_code = n; e->depth[n] = 0; }
        else t->len[n] = 0;
    }
    while (e->heap_len < 2) {
        node = e->heap[++e->heap_len] = (max_code < 2 ? ++max_code : 0);
        t->freq[node] = 1; e->depth[node] = 0; e->opt_len--;
        if (sllen) e->static_len -= sllen[node];
    }
    t->max_code = max_code;
    for (n = e->heap_len / 2; n >= 1; n--) sift_down(e, t, n);
    node = elems;
    do {
        n = e->heap[1]; e->heap[1] = e->heap[e->heap_len--]; sift_down(e, t, 1);
        m = e->heap[1];
        e->heap[--e->heap_max] = n; e->heap[--e->heap_max] = m;
        t->freq[node] = t->freq[n] + t->freq[m];
        e->depth[node] = (uint8_t)((e->depth[n] >= e->depth[m] ? e->depth[n] : e->depth[m]) + 1);
        t->dad[n] = t->dad[m] = (uint16_t)node;
        e->heap[1] = node++; sift_down(e, t, 1);
    } while (e->heap_len >= 2);
    e->heap[--e->heap_max] = e->heap[1];
    assign_lengths(e, t, max_code, sllen, extra, xbase, max_length);
    unsigned next[MAXBITS + 1], c = 0; int bits;
    for (bits = 1; bits <= MAXBITS; bits++) { c = (c + e->bl_count[bits - 1]) << 1; next[bits] = c; }
    for (n = 0; n <= max_code; n++) { int l = t->len[n]; if (l) t->code[n] = (uint16_t)bitrev(next[l]++, l); }
}

/* scan_tree trees.c:707-746 / send_tree :752-797 share this run-length walk. emit==0: count into bt.freq */
static void walk_lengths(enc *e, const htree *t, int max_code, int emit)
{
    int n, prevlen = -1, curlen, nextlen = t->len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = (n == max_code) ? 0xffff : t->len[n + 1]; /* guard, trees.c:721 */
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) {
            if (emit) { do { put_bits(&e->bs, e->bt.code[curlen], e->bt.len[curlen]); } while (--count != 0); }
            else e->bt.freq[curlen] += (uint32_t)count;
        } else if (curlen != 0) {
            if (curlen != prevlen) { if (emit) { put_bits(&e->bs, e->bt.code[curlen], e->bt.len[curlen]); count--; } else e->bt.freq[curlen]++; }
            if (emit) { put_bits(&e->bs, e->bt.code[16], e->bt.len[16]); put_bits(&e->bs, (unsigned)(count - 3), 2); } else e->bt.freq[16]++;
        } else if (count <= 10) {
            if (emit) { put_bits(&e->bs, e->bt.code[17], e->bt.len[17]); put_bits(&e->bs, (unsigned)(count - 3), 3); } else e->bt.freq[17]++;
        } else {
            if (emit) { put_bits(&e->bs, e->bt.code[18], e->bt.len[18]); put_bits(&e->bs, (unsigned)(count - 11), 7); } else e->bt.freq[18]++;
        }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

/* compress_block trees.c:1072-1118 */
static void emit_tokens(enc *e, const uint16_t *lcode, const uint16_t *llen_, const uint8_t *llen8, const uint16_t *dcode, const uint16_t *dlen_, int dlen_fixed)
{
    for (uint32_t i = 0; i < e->ntok; i++) {
        unsigned dist = e->dbuf[i], lc = e->lbuf[i];
        if (dist == 0) { put_bits(&e->bs, lcode[lc], llen_ ? llen_[lc] : llen8[lc]); continue; }
        unsigned c = T_lencode[lc], s = c + 257;
        put_bits(&e->bs, lcode[s], llen_ ? llen_[s] : llen8[s]);
        if (XL[c]) put_bits(&e->bs, lc - T_baselen[c], XL[c]);
        dist--; c = dcode_of(dist);
        put_bits(&e->bs, dcode[c], dlen_ ? dlen_[c] : dlen_fixed);
        if (XD[c]) put_bits(&e->bs, dist - T_basedist[c], XD[c]);
    }
    put_bits(&e->bs, lcode[EOB], llen_ ? llen_[EOB] : llen8[EOB]);
}

static void stored_block(enc *e, const uint8_t *buf, uint32_t len, int eof) /* trees.c:867-879,1197-1219 */
{
    put_bits(&e->bs, (unsigned)eof, 3); byte_align(&e->bs);
    sink_byte(&e->bs, len & 0xff); sink_byte(&e->bs, (len >> 8) & 0xff);
    sink_byte(&e->bs, ~len & 0xff); sink_byte(&e->bs, (~len >> 8) & 0xff);
    for (uint32_t i = 0; i < len; i++) sink_byte(&e->bs, buf[i]);
}

/* _tr_flush_block trees.c:921-1016; p_end = strstart in chunk coords */
static void close_block(enc *e, uint32_t p_end, int eof)
{
    uint32_t stored_len = p_end - e->block_start;
    long bs_w = (long)e->block_start + (long)e->base - (long)e->off; /* <0 -> buf==NULL, deflate.c:1365-1367 */
    uint64_t opt_lenb, static_lenb; int max_blindex = 0, btype;
    if (e->tok_out) for (uint32_t i = 0; i < e->ntok; i++) { ora_token t = {e->dbuf[i], e->lbuf[i], 0}; e->tok_out[e->tok_total + i] = t; }
    e->tok_total += e->ntok;
    if (e->level > 0) {
        if (stored_len > 0 && e->data_type == 2) { /* set_data_type trees.c:1126-1139 */
            int n; for (n = 0; n < 9; n++) if (e->lt.freq[n]) break;
            if (n == 9) for (n = 14; n < 32; n++) if (e->lt.freq[n]) break;
            e->data_type = (n == 32) ? 1 : 0;
        }
        grow_tree(e, &e->lt, NLIT, T_sllen, XL, 257, MAXBITS);
        static const uint8_t sdlen[NDIST] = {5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5};
        grow_tree(e, &e->dt, NDIST, sdlen, XD, 0, MAXBITS);
        walk_lengths(e, &e->lt, e->lt.max_code, 0); walk_lengths(e, &e->dt, e->dt.max_code, 0);
        grow_tree(e, &e->bt, NBL, NULL, XBL, 0, MAXBLBITS);
        for (max_blindex = NBL - 1; max_blindex >= 3; max_blindex--) if (e->bt.len[BLORDER[max_blindex]] != 0) break;
        e->opt_len += 3 * (uint64_t)(max_blindex + 1) + 5 + 5 + 4;
        opt_lenb = (e->opt_len + 3 + 7) >> 3; static_lenb = (e->static_len + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    } else opt_lenb = static_lenb = (uint64_t)stored_len + 5;

    if ((uint64_t)stored_len + 4 <= opt_lenb && bs_w >= 0) { stored_block(e, e->in + e->block_start, stored_len, eof); btype = 0; e->last_eob = 8; }
    else if (e->strategy == ORA_FIXED || static_lenb == opt_lenb) { /* trees.c:986 */
        put_bits(&e->bs, (1u << 1) + (unsigned)eof, 3);
        emit_tokens(e, T_slcode, NULL, T_sllen, T_sdcode, NULL, 5); btype = 1; e->last_eob = 7;
    } else {
        put_bits(&e->bs, (2u << 1) + (unsigned)eof, 3);
        put_bits(&e->bs, (unsigned)(e->lt.max_code + 1 - 257), 5); put_bits(&e->bs, (unsigned)(e->dt.max_code + 1 - 1), 5);
        put_bits(&e->bs, (unsigned)(max_blindex + 1 - 4), 4);
        for (int r = 0; r <= max_blindex; r++) put_bits(&e->bs, e->bt.len[BLORDER[r]], 3);
        walk_lengths(e, &e->lt, e->lt.max_code, 1); walk_lengths(e, &e->dt, e->dt.max_code, 1);
        emit_tokens(e, e->lt.code, e->lt.len, NULL, e->dt.code, e->dt.len, 0); btype = 2; e->last_eob = e->lt.len[EOB];
    }
    if (e->info && e->nblocks < 8) e->info->btype[e->nblocks] = (uint32_t)btype;
    e->nblocks++;
    new_block(e);
    if (eof) byte_align(&e->bs);
    e->block_start = p_end;
}

/* tally: h/deflate.h:308-324; returns the "flush now" flag */
static int note_literal(enc *e, unsigned c) { e->dbuf[e->ntok] = 0; e->lbuf[e->ntok++] = (uint8_t)c; e->lt.freq[c]++; return e->ntok == LITBUF - 1; }
static int note_match(enc *e, unsigned dist, unsigned lenm3)
{
    e->dbuf[e->ntok] = (uint16_t)dist; e->lbuf[e->ntok++] = (uint8_t)lenm3;
    e->lt.freq[T_lencode[lenm3] + 257]++; e->dt.freq[dcode_of(dist - 1)]++;
    return e->ntok == LITBUF - 1;
}

static uint32_t widx(const enc *e, uint32_t p) { return p + e->base - e->off; }

/* INSERT_STRING deflate.c:189-192 with the pure 3-byte hash the rolling update converges to */
static uint32_t insert_at(enc *e, uint32_t p)
{
    const uint8_t *s = e->in + p;
    uint32_t h = ((((uint32_t)s[0] << HSHIFT ^ s[1]) << HSHIFT) ^ s[2]) & HMASK, w = widx(e, p), old = e->head[h];
    e->prev[w & WMASK] = (uint16_t)old; e->head[h] = (uint16_t)w;
    return old;
}

/* the slide inside fill_window, deflate.c:1293-1326; called whenever the reference would call
 * fill_window (lookahead < MIN_LOOKAHEAD) */
static void refill(enc *e, uint32_t p)
{
    if (widx(e, p) >= (uint32_t)(WSIZE + MAXDIST)) {
        for (int i = 0; i < HSIZE; i++) e->head[i] = (uint16_t)(e->head[i] >= WSIZE ? e->head[i] - WSIZE : 0);
        for (int i = 0; i < WSIZE; i++) e->prev[i] = (uint16_t)(e->prev[i] >= WSIZE ? e->prev[i] - WSIZE : 0);
        e->off += WSIZE;
    }
}

/* longest_match deflate.c:1027-1168.  cur is a window index.  Byte compares are clipped at the end
 * of the chunk; the reference reads past it but clips the result to lookahead (deflate.c:1166-1167),
 * which yields the same return value and the same match_start whenever the result is used. */
static uint32_t best_match(enc *e, uint32_t p, uint32_t cur, uint32_t prev_length, uint32_t *match_start)
{
    uint32_t chain = e->cfg->chain, look = e->n - p, nice = e->cfg->nice, best = prev_length;
    uint32_t w = widx(e, p), limit = w > (uint32_t)MAXDIST ? w - MAXDIST : 0, cap = look < MAXM ? look : MAXM;
    if (prev_length >= e->cfg->good) chain >>= 2;
    if (nice > look) nice = look;
    do {
        uint32_t q = cur - e->base + e->off, l = 0;
        while (l < cap && e->in[q + l] == e->in[p + l]) l++;
        /* (the quick check, deflate.c:1121-1124, looks at the bytes best_len - 1 and best_len of the candidate: implied by l > best -- except with a seed of 0,
         *  which deflate_fast can inherit from deflate_slow through deflateParams: then the byte IN FRONT of the candidate must equal the one in front of p) */
        if (l < 2) l = 0; /* (the first two bytes are compared outright, deflate.c:1123-1124: a shorter prefix is no candidate even for a seed below 2) */
        if (best == 0 && (q == 0 || p == 0 || e->in[q - 1] != e->in[p - 1])) l = 0;
        if (l > best) { *match_start = q; best = l; if (l >= nice) break; }
    } while ((cur = e->prev[cur & WMASK]) > limit && --chain != 0);
    return best <= look ? best : look;
}

/* longest_match_fast deflate.c:1173-1228 (only reached with strategy Z_RLE in this build: FASTEST is not defined).  No chain,
 * no seed: the common prefix with the one candidate, MIN_MATCH-1 when shorter than MIN_MATCH, clipped to the lookahead. */
static uint32_t fast_match(enc *e, uint32_t p, uint32_t cur, uint32_t *match_start)
{
    uint32_t q = cur - e->base + e->off, look = e->n - p, cap = look < MAXM ? look : MAXM, l = 0;
    if (cap < 2 || e->in[q] != e->in[p] || e->in[q + 1] != e->in[p + 1]) return MINM - 1; /* (look >= MIN_MATCH here) */
    while (l < cap && e->in[q + l] == e->in[p + l]) l++;
    if (l < MINM) return MINM - 1;
    *match_start = q;
    return l;
}

/* how many bytes the very first fill_window can take: window_size - strstart (deflate.c:1275,1342) */
static uint32_t first_fill(const enc *e) { uint32_t room = 2 * WSIZE - e->base; return e->n < room ? e->n : room; }

static void run_slow(enc *e, int eof) /* deflate_slow deflate.c:1554-1674 */
{
    uint32_t p = e->start, n = e->n, buffered = first_fill(e), match_len = MINM - 1, prev_len, match_start = 0, prev_match, hash_head = 0;
    int pending = 0, cut;
    for (;;) {
        if (buffered - p < MIN_LOOK) { refill(e, p); buffered = n; if (n - p == 0) break; }
        uint32_t look = n - p;
        if (look >= MINM) hash_head = insert_at(e, p);
        prev_len = match_len; prev_match = match_start; match_len = MINM - 1;
        if (hash_head != 0 && prev_len < e->cfg->lazy && widx(e, p) - hash_head <= (uint32_t)MAXDIST) {
            if (e->strategy != ORA_HUFFMAN_ONLY && e->strategy != ORA_RLE) match_len = best_match(e, p, hash_head, prev_len, &match_start);
            else if (e->strategy == ORA_RLE && widx(e, p) - hash_head == 1) match_len = fast_match(e, p, hash_head, &match_start);
            if (match_len <= 5 && (e->strategy == ORA_FILTERED || (match_len == MINM && p - match_start > FAR_LIMIT))) match_len = MINM - 1; /* deflate.c:1601-1611 */
        }
        if (prev_len >= MINM && match_len <= prev_len) {
            uint32_t max_insert = p + look - MINM, k = prev_len - 2;
            cut = note_match(e, p - 1 - prev_match, prev_len - MINM);
            do { if (++p <= max_insert) hash_head = insert_at(e, p); } while (--k != 0);
            pending = 0; match_len = MINM - 1; p++;
            if (cut) close_block(e, p, 0);
        } else if (pending) {
            cut = note_literal(e, e->in[p - 1]);
            if (cut) close_block(e, p, 0);
            p++;
        } else { pending = 1; p++; }
    }
    if (pending) note_literal(e, e->in[p - 1]);
    close_block(e, p, eof);
}

static void run_fast(enc *e, int eof) /* deflate_fast deflate.c:1448-1546 */
{
    uint32_t p = e->start, n = e->n, buffered = first_fill(e), match_len = MINM - 1, match_start = 0, hash_head = 0;
    int cut;
    for (;;) {
        if (buffered - p < MIN_LOOK) { refill(e, p); buffered = n; if (n - p == 0) break; }
        uint32_t look = n - p;
        if (look >= MINM) hash_head = insert_at(e, p);
        if (hash_head != 0 && widx(e, p) - hash_head <= (uint32_t)MAXDIST) { /* deflate.c:1478-1497 */
            if (e->strategy != ORA_HUFFMAN_ONLY && e->strategy != ORA_RLE) match_len = best_match(e, p, hash_head, MINM - 1, &match_start);
            else if (e->strategy == ORA_RLE && widx(e, p) - hash_head == 1) match_len = fast_match(e, p, hash_head, &match_start);
        }
        if (match_len >= MINM) {
            cut = note_match(e, p - match_start, match_len - MINM);
            look -= match_len;
            if (match_len <= e->cfg->lazy && look >= MINM) {
                match_len--;
                do { p++; hash_head = insert_at(e, p); } while (--match_len != 0);
                p++;
            } else { p += match_len; match_len = 0; }
        } else { cut = note_literal(e, e->in[p]); p++; }
        if (cut) close_block(e, p, 0);
    }
    close_block(e, p, eof);
}

static void run_stored(enc *e, int eof) /* deflate_stored deflate.c:1390-1439, fill_window :1266-1358 */
{
    const uint32_t max_block = 65536 - 5 < 0xffff ? 65536 - 5 : 0xffff; /* pending_buf_size-5, deflate.c:1397-1402 */
    uint32_t p = e->start, look = 0, taken = e->start; /* p: strstart (chunk coords); taken: bytes read so far (a preset dictionary is already in the window) */
    for (;;) {
        if (look <= 1) {
            do { /* fill_window */
                uint32_t more = 2 * WSIZE - look - widx(e, p);
                if (widx(e, p) >= (uint32_t)(WSIZE + MAXDIST)) { e->off += WSIZE; more += WSIZE; }
                if (taken == e->n) break;
                uint32_t got = e->n - taken < more ? e->n - taken : more;
                taken += got; look += got;
            } while (look < MIN_LOOK && taken != e->n);
            if (look == 0) break;
        }
        p += look; look = 0;
        uint32_t max_start = e->block_start + max_block;
        if (p >= max_start) { look = p - max_start; p = max_start; close_block(e, p, 0); }
        if (p - e->block_start >= (uint32_t)MAXDIST) close_block(e, p, 0);
    }
    close_block(e, p, eof);
}

size_t ora_deflate_chunk(const uint8_t *in, size_t n, int level, int pos0_matchable, int is_last,
                         uint8_t *out, size_t cap, ora_token *tokens, ora_chunk_info *info)
{
    return ora_deflate_chunk_s(in, n, level, ORA_DEFAULT_STRATEGY, pos0_matchable, is_last, out, cap, tokens, info);
}

size_t ora_deflate_chunk_s(const uint8_t *in, size_t n, int level, int strategy, int pos0_matchable, int is_last,
                           uint8_t *out, size_t cap, ora_token *tokens, ora_chunk_info *info)
{
    return ora_deflate_chunk_d(in, n, 0, level, strategy, pos0_matchable, is_last, out, cap, tokens, info);
}

/* The same with a preset dictionary (deflateSetDictionary, deflate.c:315-354): `in` holds the dictionary bytes the window
 * receives (at most MAX_DIST of them, at least MIN_MATCH) followed by the data, n counts both; every dictionary position
 * but the last two is in the hash chains before the first byte of data is looked at, and strstart = block_start = dict_len. */
size_t ora_deflate_chunk_d(const uint8_t *in, size_t n, size_t dict_len, int level, int strategy, int pos0_matchable, int is_last,
                           uint8_t *out, size_t cap, ora_token *tokens, ora_chunk_info *info)
{
    if (dict_len > n || (dict_len != 0 && (dict_len < MINM || dict_len > MAXDIST || pos0_matchable))) return 0;
    if (n > ORA_CHUNK_MAX || level < 0 || level > 9 || strategy < 0 || strategy > ORA_FIXED) return 0;
    make_tables();
    enc *e = (enc *)calloc(1, sizeof(enc));
    if (!e) return 0;
    e->in = in; e->n = (uint32_t)n; e->base = pos0_matchable ? 3u : 0u; e->level = level; e->strategy = strategy; e->cfg = &LEVELS[level];
    e->bs.out = out; e->bs.cap = cap; e->tok_out = tokens; e->info = info; e->data_type = 2;
    if (info) memset(info, 0, sizeof(*info));
    new_block(e);
    e->start = (uint32_t)dict_len; e->block_start = e->start;
    for (uint32_t q = 0; q + MINM <= e->start; q++) insert_at(e, q); /* deflate.c:345-351 */
    if (e->cfg->mode == 2) run_slow(e, is_last); else if (e->cfg->mode == 1) run_fast(e, is_last); else run_stored(e, is_last);
    if (!is_last) { /* Z_FULL_FLUSH marker: _tr_stored_block(s,0,0,0), deflate.c:811-812 */
        put_bits(&e->bs, 0, 3); byte_align(&e->bs);
        sink_byte(&e->bs, 0); sink_byte(&e->bs, 0); sink_byte(&e->bs, 0xff); sink_byte(&e->bs, 0xff);
    }
    if (info) { info->ntokens = e->tok_total; info->nblocks = e->nblocks; info->data_type = (uint32_t)e->data_type; }
    size_t len = e->bs.overflow ? 0 : e->bs.len;
    free(e);
    return len;
}

size_t ora_deflate_bound(size_t n, size_t chunk_size)
{
    size_t nchunks = n ? (n + chunk_size - 1) / chunk_size : 1;
    return n + nchunks * 64 + 16;
}

size_t ora_deflate_stream(const uint8_t *in, size_t n, int level, size_t chunk_size, uint8_t *out, size_t cap)
{
    return ora_deflate_stream_s(in, n, level, ORA_DEFAULT_STRATEGY, chunk_size, out, cap);
}

size_t ora_deflate_stream_s(const uint8_t *in, size_t n, int level, int strategy, size_t chunk_size, uint8_t *out, size_t cap)
{
    if (chunk_size == 0 || chunk_size > ORA_CHUNK_MAX || cap < 6) return 0;
    /* zlib header, deflate.c:625-641 */
    unsigned hdr = (8u + (7u << 4)) << 8, lf = (strategy >= ORA_HUFFMAN_ONLY || level < 2) ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
    hdr |= lf << 6; hdr += 31 - hdr % 31;
    size_t o = 0; out[o++] = (uint8_t)(hdr >> 8); out[o++] = (uint8_t)hdr;
    size_t nchunks = n ? (n + chunk_size - 1) / chunk_size : 1;
    for (size_t k = 0; k < nchunks; k++) {
        size_t lo = k * chunk_size, len = n - lo < chunk_size ? n - lo : chunk_size;
        size_t got = ora_deflate_chunk_s(in + lo, len, level, strategy, 0, k == nchunks - 1, out + o, cap - o, NULL, NULL);
        if (got == 0) return 0;
        o += got;
    }
    if (cap - o < 4) return 0;
    uint32_t a = ora_adler32(1, in, n);
    out[o++] = (uint8_t)(a >> 24); out[o++] = (uint8_t)(a >> 16); out[o++] = (uint8_t)(a >> 8); out[o++] = (uint8_t)a;
    return o;
}

/* =====================================================================================================================
 * The CONTINUOUS stream: one deflate stream over any number of bytes, driven call by call as an application drives
 * deflate() (round 4).  Restates, on top of the per-chunk pieces above:
 *   fill_window with every slide a long stream makes        qcsrc/deflate.c:1266-1358
 *   deflate_stored / deflate_fast / deflate_slow as they RETURN need_more and are called again   :1390-1439, 1448-1546, 1554-1674
 *   the flush arm of deflate(): _tr_align, the empty stored block, CLEAR_HASH                     :808-825, trees.c:892-915
 *   deflateSetDictionary in front of the first byte          :315-354
 * Coordinates: p is the offset into the whole input (uint32: streams below 4 GiB), the reference's window index is
 * w = p - off where off grows by 32768 with every slide.  e->n is the end of what the window holds ("strstart + lookahead").
 * What is NOT modelled: a caller that runs out of output space in the middle of a call (avail_out == 0) -- the blocks are the same,
 * only the calls they leave in differ.
 * ===================================================================================================================== */
typedef struct {
    uint32_t p;        /* strstart */
    uint32_t filled;   /* strstart + lookahead: end of the bytes read into the window */
    uint32_t avail;    /* end of the bytes the caller has handed over so far (next_in + avail_in) */
    uint32_t match_len, match_start; int pending; /* deflate_slow's match_length, match_start, match_available (they live in the state, deflate.h:152-160) */
    uint32_t prev_length; /* s->prev_length: deflate_slow's; deflate_fast never sets it but longest_match seeds its search with it (deflate.c:1035), 2 in a stream that has always been fast */
} cstate;

static void cont_fill(enc *e, cstate *c) /* fill_window */
{
    do {
        refill(e, c->p);                                              /* the slide, when strstart has reached wsize + MAX_DIST */
        const uint32_t more = 2 * WSIZE - (c->filled - c->p) - widx(e, c->p);
        if (c->filled == c->avail) return;                            /* avail_in == 0 */
        const uint32_t got = c->avail - c->filled < more ? c->avail - c->filled : more; /* read_buf */
        c->filled += got;
    } while (c->filled - c->p < MIN_LOOK && c->filled != c->avail);
}

/* returns 1 when the function ran to its flush (block_done / finish_done), 0 for need_more */
static int cont_slow(enc *e, cstate *c, int flush)
{
    uint32_t prev_len, prev_match, hash_head = 0; /* (a local of deflate_slow: NIL again in every call, deflate.c:1558) */
    int cut;
    for (;;) {
        if (c->filled - c->p < MIN_LOOK) {
            cont_fill(e, c);
            if (c->filled - c->p < MIN_LOOK && flush == 0) return 0;
            if (c->filled == c->p) break;
        }
        const uint32_t look = c->filled - c->p, p = c->p;
        e->n = c->filled;
        if (look >= MINM) hash_head = insert_at(e, p);
        prev_len = c->match_len; prev_match = c->match_start; c->match_len = MINM - 1; c->prev_length = prev_len;
        if (hash_head != 0 && prev_len < e->cfg->lazy && widx(e, p) - hash_head <= (uint32_t)MAXDIST) {
            if (e->strategy != ORA_HUFFMAN_ONLY && e->strategy != ORA_RLE) c->match_len = best_match(e, p, hash_head, prev_len, &c->match_start);
            else if (e->strategy == ORA_RLE && widx(e, p) - hash_head == 1) c->match_len = fast_match(e, p, hash_head, &c->match_start);
            if (c->match_len <= 5 && (e->strategy == ORA_FILTERED || (c->match_len == MINM && p - c->match_start > FAR_LIMIT))) c->match_len = MINM - 1;
        }
        if (prev_len >= MINM && c->match_len <= prev_len) {
            uint32_t max_insert = p + look - MINM, k = prev_len - 2, q = p;
            cut = note_match(e, p - 1 - prev_match, prev_len - MINM);
            do { if (++q <= max_insert) hash_head = insert_at(e, q); } while (--k != 0);
            c->pending = 0; c->match_len = MINM - 1; c->p = q + 1; c->prev_length = 0; /* (the insertion loop counts s->prev_length down to 0, deflate.c:1630-1636) */
            if (cut) close_block(e, c->p, 0);
        } else if (c->pending) {
            cut = note_literal(e, e->in[p - 1]);
            if (cut) close_block(e, p, 0);
            c->p = p + 1;
        } else { c->pending = 1; c->p = p + 1; }
    }
    if (c->pending) { note_literal(e, e->in[c->p - 1]); c->pending = 0; }
    close_block(e, c->p, flush == 4);
    return 1;
}

static int cont_fast(enc *e, cstate *c, int flush)
{
    uint32_t hash_head = 0;
    int cut;
    for (;;) {
        if (c->filled - c->p < MIN_LOOK) {
            cont_fill(e, c);
            if (c->filled - c->p < MIN_LOOK && flush == 0) return 0;
            if (c->filled == c->p) break;
        }
        uint32_t look = c->filled - c->p, p = c->p;
        e->n = c->filled;
        if (look >= MINM) hash_head = insert_at(e, p);
        if (hash_head != 0 && widx(e, p) - hash_head <= (uint32_t)MAXDIST) {
            if (e->strategy != ORA_HUFFMAN_ONLY && e->strategy != ORA_RLE) c->match_len = best_match(e, p, hash_head, c->prev_length, &c->match_start);
            else if (e->strategy == ORA_RLE && widx(e, p) - hash_head == 1) c->match_len = fast_match(e, p, hash_head, &c->match_start);
        }
        if (c->match_len >= MINM) {
            cut = note_match(e, p - c->match_start, c->match_len - MINM);
            look -= c->match_len;
            if (c->match_len <= e->cfg->lazy && look >= MINM) {
                c->match_len--;
                do { p++; hash_head = insert_at(e, p); } while (--c->match_len != 0);
                p++;
            } else { p += c->match_len; c->match_len = 0; }
        } else { cut = note_literal(e, e->in[p]); p++; }
        c->p = p;
        if (cut) close_block(e, p, 0);
    }
    close_block(e, c->p, flush == 4);
    return 1;
}

static int cont_stored(enc *e, cstate *c, int flush)
{
    const uint32_t max_block = 65536 - 5 < 0xffff ? 65536 - 5 : 0xffff;
    for (;;) {
        if (c->filled - c->p <= 1) {
            cont_fill(e, c);
            if (c->filled == c->p && flush == 0) return 0;
            if (c->filled == c->p) break;
        }
        uint32_t look;
        c->p = c->filled; look = 0;
        const uint32_t max_start = e->block_start + max_block;
        if (c->p >= max_start) { look = c->p - max_start; c->p = max_start; close_block(e, c->p, 0); }
        (void)look; /* (the bytes behind max_start stay in the window as lookahead: filled does not move) */
        if (c->p - e->block_start >= (uint32_t)MAXDIST) close_block(e, c->p, 0);
    }
    close_block(e, c->p, flush == 4);
    return 1;
}

static void cont_align(enc *e) /* _tr_align, trees.c:892-915; the bit sink holds fewer than 8 bits between calls, which is bi_valid after bi_flush */
{
    put_bits(&e->bs, 1u << 1, 3); put_bits(&e->bs, T_slcode[EOB], T_sllen[EOB]);
    if (1 + e->last_eob + 10 - e->bs.nacc < 9) { put_bits(&e->bs, 1u << 1, 3); put_bits(&e->bs, T_slcode[EOB], T_sllen[EOB]); }
    e->last_eob = 7;
}

/* in[0 .. dict_len) is the part of a preset dictionary the window receives (the caller cuts it to its last MAX_DIST bytes; 0: none), the data
 * follow; n counts both.  Call k hands deflate() the bytes up to offset cuts[k] (offsets into the DATA, ascending) with flush kinds[k]
 * (0 Z_NO_FLUSH, 1 Z_PARTIAL_FLUSH, 2 Z_SYNC_FLUSH, 3 Z_FULL_FLUSH); one more call hands over the rest with Z_FINISH.
 * Output: the raw deflate stream.  Returns its length, 0 when cap is too small or an argument is off. */
size_t ora_deflate_cont(const uint8_t *in, size_t n, size_t dict_len, int level, int strategy, const uint32_t *cuts, const int32_t *kinds, size_t ncuts,
                        uint8_t *out, size_t cap)
{
    return ora_deflate_cont_p(in, n, dict_len, level, strategy, cuts, kinds, NULL, NULL, ncuts, out, cap);
}

/* The same with deflateParams() (deflate.c:416-451) in front of some calls: plevel[k] / pstrategy[k] >= 0 are set before call k (k == ncuts: before the
 * Z_FINISH call; the arrays hold ncuts + 1 entries; -1: no change).  A change of the compress function flushes what has been read with Z_PARTIAL_FLUSH
 * first, any other change takes effect where the loop stands. */
size_t ora_deflate_cont_p(const uint8_t *in, size_t n, size_t dict_len, int level, int strategy, const uint32_t *cuts, const int32_t *kinds,
                          const int32_t *plevel, const int32_t *pstrategy, size_t ncuts, uint8_t *out, size_t cap)
{
    if (n >= 0xfff00000u || dict_len > n || (dict_len != 0 && (dict_len < MINM || dict_len > MAXDIST))) return 0;
    if (level < 0 || level > 9 || strategy < 0 || strategy > ORA_FIXED) return 0;
    make_tables();
    enc *e = (enc *)calloc(1, sizeof(enc));
    if (!e) return 0;
    e->in = in; e->n = (uint32_t)dict_len; e->level = level; e->strategy = strategy; e->cfg = &LEVELS[level];
    e->bs.out = out; e->bs.cap = cap; e->data_type = 2; e->last_eob = 8;
    new_block(e);
    e->start = (uint32_t)dict_len; e->block_start = e->start;
    for (uint32_t q = 0; q + MINM <= e->start; q++) insert_at(e, q);
    cstate c; memset(&c, 0, sizeof c);
    c.p = c.filled = c.avail = (uint32_t)dict_len; c.match_len = MINM - 1; c.prev_length = MINM - 1; /* lm_init, deflate.c:1003-1005 */
    int prev_flush = 0;
    for (size_t k = 0; k <= ncuts; k++) {
        if (plevel && pstrategy && (plevel[k] >= 0 || pstrategy[k] >= 0)) { /* deflateParams */
            const int nl = plevel[k] >= 0 ? plevel[k] : e->level, ns = pstrategy[k] >= 0 ? pstrategy[k] : e->strategy;
            if (nl > 9 || ns > ORA_FIXED) { free(e); return 0; }
            if (LEVELS[nl].mode != e->cfg->mode && c.avail != (uint32_t)dict_len && prev_flush < 1) { /* func changes and total_in != 0: deflate(strm, Z_PARTIAL_FLUSH) -- which answers Z_BUF_ERROR and does nothing right behind another flush (deflate.c:774-777; no input is pending here) */
                const int done = e->cfg->mode == 2 ? cont_slow(e, &c, 1) : e->cfg->mode == 1 ? cont_fast(e, &c, 1) : cont_stored(e, &c, 1);
                if (done) cont_align(e);
                prev_flush = 1;
            }
            e->level = nl; e->strategy = ns; e->cfg = &LEVELS[nl];
        }
        const int flush = k == ncuts ? 4 : kinds[k];
        const uint32_t upto = k == ncuts ? (uint32_t)n : (uint32_t)dict_len + cuts[k];
        if (upto < c.avail || upto > n || flush < 0 || flush > 4 || (k < ncuts && flush == 4)) { free(e); return 0; }
        if (upto == c.avail && flush <= prev_flush && flush != 4) { free(e); return 0; } /* the reference answers Z_BUF_ERROR, deflate.c:774-777 */
        const int had_in = upto != c.avail;
        c.avail = upto; prev_flush = flush;
        if (!(had_in || c.filled != c.p || flush != 0)) continue; /* deflate.c:786-787 */
        const int done = e->cfg->mode == 2 ? cont_slow(e, &c, flush) : e->cfg->mode == 1 ? cont_fast(e, &c, flush) : cont_stored(e, &c, flush);
        if (done && flush != 4) { /* block_done, deflate.c:808-819 */
            if (flush == 1) cont_align(e);
            else {
                put_bits(&e->bs, 0, 3); byte_align(&e->bs);
                sink_byte(&e->bs, 0); sink_byte(&e->bs, 0); sink_byte(&e->bs, 0xff); sink_byte(&e->bs, 0xff);
                e->last_eob = 8;
                if (flush == 3) memset(e->head, 0, sizeof e->head); /* CLEAR_HASH */
            }
        }
    }
    const size_t len = e->bs.overflow ? 0 : e->bs.len;
    free(e);
    return len;
}
