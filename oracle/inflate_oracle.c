/* inflate_oracle.c -- CPU restatement of the zlib-1.2.3 decode path for a complete in-memory stream.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity: PINNED against oracle/_ref/libzref.so
 * (tests/test_oracle_vs_reference.py: identical bytes, return codes and error strings).
 *
 * Restated (file:line under /root/reference):
 *   inflate() block/header state machine   qcsrc/inflate.c:773-949
 *   symbol decode loop                      qcsrc/inffast.c:67-302 and qcsrc/inflate.c:950-1076
 *   code-length validation                  qcsrc/inftrees.c:106-138
 *   length/distance bases and extra bits    qcsrc/inftrees.c:60-73
 *   zlib wrapper: header + Adler trailer    qcsrc/inflate.c:589-632, 1077-1098
 *
 * The reference decodes through 2-level lookup tables; only the decoded bytes, the return code and the
 * error text are observable, so this restatement uses a canonical count/offset decoder and reproduces
 * the *acceptance rules* of inflate_table instead of its table layout.
 */
#include "oracle.h"
#include <string.h>

enum { KIND_CODES = 0, KIND_LENS = 1, KIND_DISTS = 2 };

typedef struct { const uint8_t *in; size_t n, pos; uint64_t acc; int nacc; } bitsrc;

/* returns 0 on success, -1 when the input is exhausted */
static int need(bitsrc *b, int k)
{
    while (b->nacc < k) {
        if (b->pos >= b->n) return -1;
        b->acc |= (uint64_t)b->in[b->pos++] << b->nacc; b->nacc += 8;
    }
    return 0;
}
static unsigned take(bitsrc *b, int k) { unsigned v = (unsigned)(b->acc & ((1ull << k) - 1)); b->acc >>= k; b->nacc -= k; return v; }

typedef struct { uint16_t count[16]; uint16_t sym[320]; int maxlen; int empty; } canon;

/* Acceptance rules of inflate_table (inftrees.c:106-138): returns 0 ok, -1 rejected. */
static int canon_build(canon *c, const uint16_t *lens, int n, int kind)
{
    int len, s, left = 1; uint16_t offs[16];
    memset(c->count, 0, sizeof c->count);
    for (s = 0; s < n; s++) c->count[lens[s]]++;
    for (c->maxlen = 15; c->maxlen >= 1; c->maxlen--) if (c->count[c->maxlen]) break;
    c->empty = (c->maxlen == 0);
    if (c->empty) return 0; /* decoding any symbol then fails (inftrees.c:117-125) */
    for (len = 1; len <= 15; len++) { left <<= 1; left -= c->count[len]; if (left < 0) return -1; }
    if (left > 0 && (kind == KIND_CODES || c->maxlen != 1)) return -1;
    offs[1] = 0;
    for (len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + c->count[len]);
    for (s = 0; s < n; s++) if (lens[s]) c->sym[offs[lens[s]]++] = (uint16_t)s;
    return 0;
}

/* decode one symbol; -1 input exhausted, -2 bit pattern not assigned (incomplete/empty code) */
static int canon_decode(bitsrc *b, const canon *c)
{
    int code = 0, first = 0, index = 0, len;
    if (c->empty) { if (need(b, 1)) return -1; return -2; }
    for (len = 1; len <= c->maxlen; len++) {
        if (need(b, 1)) return -1;
        code |= (int)take(b, 1);
        int count = c->count[len];
        if (code - count < first) return c->sym[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -2;
}

static const uint16_t LBASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t  LEXT[29]  = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DBASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t  DEXT[30]  = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint8_t  ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

#define FAIL(text) do { *msg = (text); rc = ORA_DATA_ERROR; goto done; } while (0)
#define STARVED() do { rc = ORA_BUF_ERROR; goto done; } while (0)

int ora_inflate_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *used, size_t *produced, const char **msg)
{
    bitsrc b = {in, n, 0, 0, 0};
    size_t o = 0; int rc = ORA_STREAM_END, last = 0; const char *dummy; if (!msg) msg = &dummy; *msg = NULL;
    canon lc, dc, cc; uint16_t lens[320];
    while (!last) {
        if (need(&b, 3)) STARVED();
        last = (int)take(&b, 1);
        unsigned type = take(&b, 2);
        if (type == 3) FAIL("invalid block type");
        if (type == 0) {
            take(&b, b.nacc & 7);
            if (need(&b, 32)) STARVED();
            unsigned len = take(&b, 16), nlen = take(&b, 16);
            if (len != (nlen ^ 0xffffu)) FAIL("invalid stored block lengths");
            /* bytes still in the accumulator first, then straight from the input */
            while (len) {
                if (b.nacc == 0 && b.pos >= b.n) STARVED();
                if (o >= cap) STARVED();
                if (need(&b, 8)) STARVED();
                out[o++] = (uint8_t)take(&b, 8); len--;
            }
            continue;
        }
        if (type == 1) {
            int s; for (s = 0; s < 144; s++) lens[s] = 8; for (; s < 256; s++) lens[s] = 9; for (; s < 280; s++) lens[s] = 7; for (; s < 288; s++) lens[s] = 8;
            canon_build(&lc, lens, 288, KIND_LENS);
            for (s = 0; s < 32; s++) lens[s] = 5;
            canon_build(&dc, lens, 32, KIND_DISTS);
        } else {
            if (need(&b, 14)) STARVED();
            unsigned nlen = take(&b, 5) + 257, ndist = take(&b, 5) + 1, ncode = take(&b, 4) + 4, have = 0;
            if (nlen > 286 || ndist > 30) FAIL("too many length or distance symbols");
            memset(lens, 0, sizeof lens);
            for (unsigned i = 0; i < ncode; i++) { if (need(&b, 3)) STARVED(); lens[ORDER[i]] = (uint16_t)take(&b, 3); }
            if (canon_build(&cc, lens, 19, KIND_CODES)) FAIL("invalid code lengths set");
            uint16_t ll[320]; memset(ll, 0, sizeof ll);
            while (have < nlen + ndist) {
                int s = canon_decode(&b, &cc);
                if (s == -1) STARVED();
                if (s < 0) FAIL("invalid code lengths set"); /* unreachable: complete code enforced */
                if (s < 16) { ll[have++] = (uint16_t)s; continue; }
                unsigned rep, val = 0;
                if (s == 16) { if (need(&b, 2)) STARVED(); if (have == 0) FAIL("invalid bit length repeat"); val = ll[have - 1]; rep = 3 + take(&b, 2); }
                else if (s == 17) { if (need(&b, 3)) STARVED(); rep = 3 + take(&b, 3); }
                else { if (need(&b, 7)) STARVED(); rep = 11 + take(&b, 7); }
                if (have + rep > nlen + ndist) FAIL("invalid bit length repeat");
                while (rep--) ll[have++] = (uint16_t)val;
            }
            if (canon_build(&lc, ll, (int)nlen, KIND_LENS)) FAIL("invalid literal/lengths set");
            if (canon_build(&dc, ll + nlen, (int)ndist, KIND_DISTS)) FAIL("invalid distances set");
        }
        for (;;) {
            int s = canon_decode(&b, &lc);
            if (s == -1) STARVED();
            if (s == -2 || s > 285) FAIL("invalid literal/length code");
            if (s < 256) { if (o >= cap) STARVED(); out[o++] = (uint8_t)s; continue; }
            if (s == 256) break;
            s -= 257;
            if (need(&b, LEXT[s])) STARVED();
            unsigned len = LBASE[s] + take(&b, LEXT[s]);
            int d = canon_decode(&b, &dc);
            if (d == -1) STARVED();
            if (d == -2 || d > 29) FAIL("invalid distance code");
            if (need(&b, DEXT[d])) STARVED();
            unsigned dist = DBASE[d] + take(&b, DEXT[d]);
            if (dist > o) FAIL("invalid distance too far back");
            if (o + len > cap) { /* copy what fits, then report the shortage like the reference's partial progress */
                while (o < cap) { out[o] = out[o - dist]; o++; }
                STARVED();
            }
            for (unsigned i = 0; i < len; i++) { out[o] = out[o - dist]; o++; }
        }
    }
done:
    /* whole bytes left in the bit accumulator were not consumed (inffast.c:288-291) */
    if (used) *used = b.pos - (size_t)(b.nacc >> 3);
    if (produced) *produced = o;
    return rc;
}

int ora_inflate_zlib(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *used, size_t *produced, const char **msg)
{
    const char *dummy; if (!msg) msg = &dummy; *msg = NULL;
    if (used) *used = 0;
    if (produced) *produced = 0;
    if (n < 2) return ORA_BUF_ERROR;
    unsigned hdr = ((unsigned)in[0] << 8) | in[1];
    if (hdr % 31) { *msg = "incorrect header check"; return ORA_DATA_ERROR; }
    if ((in[0] & 15) != 8) { *msg = "unknown compression method"; return ORA_DATA_ERROR; }
    if ((unsigned)(in[0] >> 4) + 8 > 15) { *msg = "invalid window size"; return ORA_DATA_ERROR; }
    if (in[1] & 0x20) return ORA_NEED_DICT;
    size_t u = 0, p = 0;
    int rc = ora_inflate_raw(in + 2, n - 2, out, cap, &u, &p, msg);
    if (used) *used = 2 + u;
    if (produced) *produced = p;
    if (rc != ORA_STREAM_END) return rc;
    if (n - 2 - u < 4) return ORA_BUF_ERROR;
    const uint8_t *t = in + 2 + u;
    uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    if (used) *used = 2 + u + 4;
    if (want != ora_adler32(1, out, p)) { *msg = "incorrect data check"; return ORA_DATA_ERROR; }
    return ORA_STREAM_END;
}
