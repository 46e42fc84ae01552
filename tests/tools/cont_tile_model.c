/* cont_tile_model.c -- CPU model of how the device cuts ONE continuous deflate stream (levels 4-9) into tiles (round 4; TEST INFRASTRUCTURE).
 *
 * The reference slides a 32 KiB window through the whole input (qcsrc/deflate.c:1266-1358) and its lazy parse (deflate.c:1554-1674) is one serial
 * chain.  The device works on TILES: tile i sees the 64 KiB of input that start at 32512 * i, parses the positions [h0, h1) = [32512, 65024) of
 * it (tile 0: from 0) and uses the 32512 bytes in front as history -- MAX_DIST is 32506 -- and the 512 bytes behind h1 as room for the
 * lazy-evaluation game and the lookahead of the last positions.  A tile starts at the neutral position ("nothing in hand") its predecessor's
 * parse ended in.  Everything the reference keeps in window coordinates is restated as a function of the ABSOLUTE position here:
 *   - the hash chain of p: the earlier positions with p's 3-byte hash, nearest first, built from the tile's own 64 KiB only;
 *   - in reach: the first candidate at most MAX_DIST back (deflate.c:1588), the others strictly less (deflate.c:1163), never position 0 (NIL);
 *   - slide k happens at the first loop top at or above thr_k = max(32768 (k-1) + 65274, min(N, 32768 (k-1) + 65536) - 261) (fill_window is
 *     called when lookahead < MIN_LOOKAHEAD and slides when strstart >= wsize + MAX_DIST);
 *   - a block may be stored only if its first byte is still in the window when it is flushed (deflate.c:1364-1367): block_start >= 32768 * slides;
 *   - the one position whose first candidate, exactly MAX_DIST back, has become NIL by a slide at that very loop top: 32768 k + 65274 when it lies
 *     in the last 261 bytes of the input.
 * This program runs that model and compares its bytes with ora_deflate_cont (the restatement of the reference's own loop, checked against the
 * compiled reference by tests/test_oracle_vs_reference.py).  Usage: cont_tile_model <level> <file> [strategy]; exit code 0 = identical.
 */
#include "../../oracle/deflate_oracle.c"
#include <stdio.h>

enum { T_STRIDE = 32512, T_H1 = 65024, T_WIN = 65536 };

static uint32_t g_link[T_WIN];  /* local position + 1 of the previous position with the same hash, 0: none */
static uint32_t g_head[HSIZE];

typedef struct { const uint8_t *in; uint32_t n; uint32_t wb, nloc; const level_cfg *cfg; int strategy; uint32_t nil_pos; } tile;

static void tile_chains(tile *t)
{
    memset(g_head, 0, sizeof g_head);
    for (uint32_t x = 0; x + MINM <= t->nloc; x++) {
        const uint8_t *s = t->in + t->wb + x;
        const uint32_t h = ((((uint32_t)s[0] << HSHIFT ^ s[1]) << HSHIFT) ^ s[2]) & HMASK;
        g_link[x] = g_head[h]; g_head[h] = x + 1;
    }
}

/* longest_match at local x with the match in hand of length prev_len (deflate.c:1027-1168); 2: nothing better */
static uint32_t tile_search(const tile *t, uint32_t x, uint32_t prev_len, uint32_t *mstart)
{
    const uint32_t P = t->wb + x, look = t->wb + t->nloc - P;
    if (look < MINM) return MINM - 1;                      /* no INSERT_STRING, a stale hash_head gives nothing that counts */
    if (look < MIN_LOOK && t->wb + t->nloc != t->n) { fprintf(stderr, "model: position %u of a tile has only %u bytes of lookahead\n", x, look); exit(3); }
    uint32_t cur = g_link[x];
    if (cur == 0 || prev_len >= t->cfg->lazy) return MINM - 1;
    uint32_t q = cur - 1;
    if (t->wb + q == 0 || x - q > (uint32_t)MAXDIST) return MINM - 1;       /* hash_head NIL or out of reach (deflate.c:1588) */
    if (P == t->nil_pos && x - q == (uint32_t)MAXDIST) return MINM - 1;      /* ... or NIL since the slide at this very loop top */
    uint32_t chain = t->cfg->chain, nice = t->cfg->nice, best = prev_len, cap = look < MAXM ? look : MAXM;
    if (prev_len >= t->cfg->good) chain >>= 2;
    if (nice > look) nice = look;
    const uint8_t *b = t->in + t->wb;
    for (;;) {
        uint32_t l = 0;
        while (l < cap && b[q + l] == b[x + l]) l++;
        if (l > best) { *mstart = q; best = l; if (l >= nice) break; }
        cur = g_link[q];
        if (cur == 0) break;
        q = cur - 1;
        if (t->wb + q == 0 || x - q >= (uint32_t)MAXDIST) break;             /* deflate.c:1163: strictly inside MAX_DIST, not NIL */
        if (--chain == 0) break;
    }
    return best <= look ? best : look;
}

static uint32_t slides_at(uint32_t ptop, uint32_t n) /* slides that have happened when the loop stands at ptop */
{
    uint32_t k = 0;
    for (;;) {
        const uint64_t base = 32768ull * k, full = base + 65536;
        const uint64_t a = base + 65274, b = (n < full ? n : full) - 261;
        const uint64_t thr = (n < 262 ? a : (a > b ? a : b));
        if (ptop < thr) return k;
        k++;
    }
}

static size_t tile_deflate(const uint8_t *in, size_t n, int level, int strategy, uint8_t *out, size_t cap, uint32_t *ntiles)
{
    make_tables();
    enc *e = (enc *)calloc(1, sizeof(enc));
    e->in = in; e->n = (uint32_t)n; e->level = level; e->strategy = strategy; e->cfg = &LEVELS[level];
    e->bs.out = out; e->bs.cap = cap; e->data_type = 2; e->last_eob = 8;
    new_block(e);
    tile t; memset(&t, 0, sizeof t);
    t.in = in; t.n = (uint32_t)n; t.cfg = e->cfg; t.strategy = strategy; t.nil_pos = 0xffffffffu;
    for (uint64_t k = 0;; k++) { const uint64_t ps = 32768 * k + 65274; if (ps >= n) break; if (ps + 261 >= n) t.nil_pos = (uint32_t)ps; }
    uint32_t P = 0;           /* neutral position the next tile starts at */
    int pending = 0;          /* the byte at P - 1 is a literal that has not been tallied yet (match_available) */
    *ntiles = 0;
    while (P < n) {
        const uint32_t i = P < T_H1 ? 0 : P / T_STRIDE - 1;   /* the tile whose range [h0, h1) holds P: tile i >= 1 parses [32512 (i + 1), 32512 (i + 2)) */
        t.wb = i * (uint32_t)T_STRIDE;
        t.nloc = n - t.wb < T_WIN ? (uint32_t)(n - t.wb) : T_WIN;
        const uint32_t h1 = t.nloc < T_H1 ? t.nloc : T_H1;
        tile_chains(&t);
        (*ntiles)++;
        uint32_t x = P - t.wb, hand_len = MINM - 1, hand_start = 0;
        for (;;) { /* the loop of deflate_slow, local coordinates */
            if (x >= t.nloc) break;                    /* the end of the input (only the last tile's window ends before its h1 + 512) */
            if (hand_len < MINM && x >= h1) break;     /* neutral at or behind h1: the next tile's */
            uint32_t ms = hand_start, ml;
            const uint32_t prev_len = hand_len, prev_start = hand_start;
            ml = tile_search(&t, x, prev_len, &ms);
            if (ml <= prev_len) ml = MINM - 1;                               /* (longest_match returns the seed when nothing is longer) */
            if (ml <= 5 && ml >= MINM && (strategy == ORA_FILTERED || (ml == MINM && x - ms > FAR_LIMIT))) ml = MINM - 1;
            int cut;
            if (prev_len >= MINM && ml <= prev_len) {                        /* the match in hand stands: emitted while the loop is at x = its start + 1 */
                cut = note_match(e, (x - 1) - prev_start, prev_len - MINM);
                const uint32_t ptop = t.wb + x;
                x = (x - 1) + prev_len; hand_len = MINM - 1; pending = 0;
                if (cut) { e->off = 32768u * slides_at(ptop, (uint32_t)n); close_block(e, t.wb + x, 0); }
            } else {
                if (pending) {
                    cut = note_literal(e, in[t.wb + x - 1]);
                    if (cut) { e->off = 32768u * slides_at(t.wb + x, (uint32_t)n); close_block(e, t.wb + x, 0); }
                }
                pending = 1; hand_len = ml; hand_start = ms; x++;
            }
        }
        P = t.wb + x;
    }
    if (pending) note_literal(e, in[n - 1]);
    e->off = 32768u * slides_at((uint32_t)n, (uint32_t)n);
    close_block(e, (uint32_t)n, 1);
    const size_t len = e->bs.overflow ? 0 : e->bs.len;
    free(e);
    return len;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s level file [strategy]\n", argv[0]); return 2; }
    const int level = atoi(argv[1]), strategy = argc > 3 ? atoi(argv[3]) : 0;
    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *in = (uint8_t *)malloc((size_t)n + 1);
    if (fread(in, 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    const size_t cap = (size_t)n + ((size_t)n >> 3) + 4096;
    uint8_t *a = (uint8_t *)malloc(cap), *b = (uint8_t *)malloc(cap);
    uint32_t ntiles = 0;
    const size_t la = ora_deflate_cont(in, (size_t)n, 0, level, strategy, NULL, NULL, 0, a, cap);
    const size_t lb = tile_deflate(in, (size_t)n, level, strategy, b, cap, &ntiles);
    size_t d = 0;
    while (d < la && d < lb && a[d] == b[d]) d++;
    const int same = la == lb && d == la;
    printf("level %d strategy %d: %ld bytes, %u tiles, reference loop %zu bytes, tile model %zu bytes: %s", level, strategy, n, ntiles, la, lb, same ? "identical\n" : "DIFFERENT");
    if (!same) printf(" (first difference at byte %zu)\n", d);
    return same ? 0 : 1;
}
