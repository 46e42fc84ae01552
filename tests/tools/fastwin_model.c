/* Model of the window form of deflate_fast (levels 1-3, deflate.c:1448-1546) that fastwin_kernel (zlib_amd/csrc/zgpu_lz_fastwin.hip) runs --
 * NOT part of the product; a tool to check the algorithm's exactness on the CPU and to count what it costs.
 *
 * One wave per chunk, 64 positions per window.  The chains of deflate_fast depend on the parse (positions inside a match longer than
 * max_insert_length are never inserted), so they are kept as: the positions counting-sorted by hash (S, idx, rank: what sort3_kernel writes) plus
 * ONE BIT per S index, "in the chains".  The predecessors of position p in its bucket are S[idx-1], S[idx-2], ...; the chain of p is those whose
 * bit is set, so the bits of the DEPTH nearest predecessors are DEPTH consecutive bits of the bitmap: one read, a few ctz.
 * Per window: all lanes evaluate longest_match at their own position under the bits as they stand -- final below the window, a guess (set)
 * inside it; a scalar walk follows the token starts; a long match clears the bits of its inside; a later token start whose search had
 * examined a cleared position is stale: everything from there on is evaluated again.  A search that needs more than DEPTH predecessors is
 * done the slow way (all of the bucket) when the walk reaches it.
 *   gcc -O2 -o fastwin_model fastwin_model.c && ./fastwin_model [first nchunks level kind depth n base] */
#include "../../zlib_amd/csrc/corpus.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define NMAX 65536
#define MAX_DIST 32506
static const int cfg[4][4] = {{0,0,0,0},{4,4,8,4},{4,5,16,8},{4,6,32,32}}; /* good, max_insert, nice, chain */
static uint8_t b[NMAX + 600];
static int S[NMAX], idx[NMAX], rnk[NMAX], bstart[32769];
static uint8_t F[NMAX]; /* by S index */
static int n, base, maxins, nice, chain, DEPTH;
static unsigned hash3(const uint8_t *p) { return (((unsigned)(p[0] & 31) << 10) ^ ((unsigned)p[1] << 5) ^ p[2]) & 0x7fff; }
static int lcp(int q, int p, int cap) { int l = 0; while (l < cap && b[q + l] == b[p + l]) l++; return l; }

/* the plain loop, with a table of flags by position: what the reference does */
static uint32_t ref_tok[NMAX]; static int ref_ntok;
static uint8_t rf[NMAX];
static void reference_parse(void)
{
    memset(rf, 0, sizeof rf); ref_ntok = 0;
    const int npos = n >= 3 ? n - 2 : 0;
    for (int p = 0; p < n;) {
        int len = 2, mstart = 0;
        const int look = n - p;
        if (p < npos) {
            int first = 1, best = 2, ch = chain, ni = nice < look ? nice : look;
            const int w = p + base, limit = w > MAX_DIST ? w - MAX_DIST : 0, cap = look < 258 ? look : 258;
            for (int k = 0; k < rnk[p]; k++) {
                const int q = S[idx[p] - 1 - k];
                if (!rf[q]) continue;
                const int wq = q + base;
                if (first) { if (wq <= 0 || w - wq > MAX_DIST) break; first = 0; } else if (wq <= limit) break;
                const int l = lcp(q, p, cap);
                if (l > best) { best = l; mstart = q; if (l >= ni) break; }
                if (--ch == 0) break;
            }
            rf[p] = 1;
            len = first ? 2 : (best < look ? best : look);
        }
        if (len >= 3) {
            ref_tok[ref_ntok++] = (uint32_t)(len - 3) | ((uint32_t)(p - mstart) << 8);
            if (len <= maxins && look - len >= 3) for (int k = 1; k < len; k++) rf[p + k] = 1;
            p += len;
        } else { ref_tok[ref_ntok++] = b[p]; p++; }
    }
}

/* ---- the window form ---- */
enum { R_LIT, R_MATCH, R_INCOMPLETE };
typedef struct { int kind, len, term, mstart; uint64_t dep; } lane_res;
static long st_windows, st_evals, st_slow, st_tokens, st_ext, st_stale, st_rounds, st_lanes_deep;

static void eval_lane(int p, int w0, lane_res *o)
{
    const int i = idx[p], r = rnk[p], look = n - p;
    const int w = p + base, limit = w > MAX_DIST ? w - MAX_DIST : 0, cap = look < 258 ? look : 258, ni = nice < look ? nice : look;
    int best = 2, first = 1, stopped = 0, nsel = 0;
    o->kind = R_LIT; o->len = 1; o->term = 0; o->mstart = 0; o->dep = 0;
    const int dmax = r < DEPTH ? r : DEPTH;
    for (int k = 0; k < dmax && nsel < chain; k++) {
        if (!F[i - 1 - k]) continue;
        const int q = S[i - 1 - k], wq = q + base;
        if (q >= w0) o->dep |= 1ull << (q - w0);
        if (first) { if (wq <= 0 || w - wq > MAX_DIST) { stopped = 1; break; } first = 0; } else if (wq <= limit) { stopped = 1; break; }
        nsel++;
        int l = lcp(q, p, cap < nice ? cap : nice); /* the lanes compare `nice` bytes at most */
        if (l > best) { best = l; o->mstart = q; if (l >= ni) { stopped = 1; o->term = l < cap; break; } }
    }
    if (!stopped && nsel < chain && r > DEPTH) { o->kind = R_INCOMPLETE; return; }
    if (!first && best >= 3) { o->kind = R_MATCH; o->len = best; }
}
/* the whole bucket, when the walk stands on p (all bits below p are final) */
static void slow_lane(int p, lane_res *o)
{
    const int i = idx[p], r = rnk[p], look = n - p;
    const int w = p + base, limit = w > MAX_DIST ? w - MAX_DIST : 0, cap = look < 258 ? look : 258, ni = nice < look ? nice : look;
    int best = 2, first = 1, ch = chain;
    o->kind = R_LIT; o->len = 1; o->term = 0; o->mstart = 0;
    for (int k = 0; k < r; k++) {
        if (!F[i - 1 - k]) continue;
        const int q = S[i - 1 - k], wq = q + base;
        if (first) { if (wq <= 0 || w - wq > MAX_DIST) break; first = 0; } else if (wq <= limit) break;
        const int l = lcp(q, p, cap);
        if (l > best) { best = l; o->mstart = q; if (l >= ni) break; }
        if (--ch == 0) break;
    }
    if (!first && best >= 3) { o->kind = R_MATCH; o->len = best; }
}

static uint32_t tok[NMAX]; static int ntok;
static void window_parse(void)
{
    const int npos = n >= 3 ? n - 2 : 0;
    memset(F, 0, sizeof F); ntok = 0;
    int pos = 0, cross_short = 0;
    for (int w0 = 0; w0 < n; w0 += 64) {
        if (pos >= w0 + 64) { /* the whole window lies inside a match (a short one is at most 6 long: its inside would have been set below) */ continue; }
        st_windows++;
        const int entry = pos - w0;
        for (int L = 0; L < 64; L++) { const int p = w0 + L; if (p < npos && (L >= entry || cross_short)) F[idx[p]] = 1; }
        lane_res res[64];
        int start = entry;
        for (;;) { /* rounds */
            st_rounds++;
            for (int L = start; L < 64; L++) { const int p = w0 + L; if (p < npos) eval_lane(p, w0, &res[L]); else { res[L].kind = R_LIT; res[L].len = 1; res[L].dep = 0; res[L].term = 0; } }
            st_evals++;
            /* the walk */
            uint64_t cleared = 0; int L = start, stop = -1, slow = 0;
            while (L < 64 && w0 + L < n) {
                const int p = w0 + L;
                if (res[L].dep & cleared) { stop = L; st_stale++; break; }
                if (res[L].kind == R_INCOMPLETE) { stop = L; slow = 1; break; }
                st_tokens++;
                if (res[L].kind == R_MATCH) {
                    int len = res[L].len;
                    if (res[L].term) { const int look = n - p; len = lcp(res[L].mstart, p, look < 258 ? look : 258); st_ext++; }
                    tok[ntok++] = (uint32_t)(len - 3) | ((uint32_t)(p - res[L].mstart) << 8);
                    const int sh = len <= maxins && n - p - len >= 3;
                    if (!sh) for (int k = 1; k < len && L + k < 64; k++) { cleared |= 1ull << (L + k); if (p + k < npos) F[idx[p + k]] = 0; }
                    if (L + len >= 64) cross_short = sh;
                    L += len;
                } else { tok[ntok++] = b[p]; L++; }
            }
            if (stop < 0) { pos = w0 + L; break; }
            if (slow) { /* this one token the slow way, then on with the lanes behind it */
                const int p = w0 + stop;
                lane_res o; slow_lane(p, &o); st_slow++; st_tokens++;
                if (o.kind == R_MATCH) {
                    const int len = o.len;
                    tok[ntok++] = (uint32_t)(len - 3) | ((uint32_t)(p - o.mstart) << 8);
                    const int sh = len <= maxins && n - p - len >= 3;
                    if (!sh) for (int k = 1; k < len && stop + k < 64; k++) if (p + k < npos) F[idx[p + k]] = 0;
                    if (stop + len >= 64) cross_short = sh;
                    start = stop + len;
                } else { tok[ntok++] = b[p]; start = stop + 1; }
                if (start >= 64 || w0 + start >= n) { pos = w0 + start; break; }
            } else start = stop;
        }
    }
}

int main(int argc, char **argv)
{
    const long first = argc > 1 ? atol(argv[1]) : 0, nch = argc > 2 ? atol(argv[2]) : 16;
    const int level = argc > 3 ? atoi(argv[3]) : 1, kind = argc > 4 ? atoi(argv[4]) : 0;
    DEPTH = argc > 5 ? atoi(argv[5]) : 32;
    n = argc > 6 ? atoi(argv[6]) : NMAX; base = argc > 7 ? atoi(argv[7]) : 0;
    maxins = cfg[level][1]; nice = cfg[level][2]; chain = cfg[level][3];
    long bad = 0;
    for (long c = first; c < first + nch; c++) {
        if (kind < 2) zc_fill_chunk(kind, kind ? 0x10C7E47ull : 0x5EED5117ull, (uint64_t)c, b);
        else { /* hostile: short periods, runs, few symbols */
            uint32_t x = (uint32_t)c * 2654435761u + 12345u;
            for (int i = 0; i < NMAX; i++) { x = x * 1664525u + 1013904223u; const uint32_t r = x >> 8;
                b[i] = kind == 2 ? (uint8_t)("ab"[ (r >> 3) & 1 ]) : kind == 3 ? (uint8_t)((i % ((int)(c % 7) + 1)) + 'a') : (uint8_t)((r % 5 == 0) ? 'x' : 'a' + (r >> 5) % 3); }
        }
        memset(b + n, 0, 600);
        static int cnt[32769], fill[32768];
        memset(cnt, 0, sizeof cnt);
        const int npos = n >= 3 ? n - 2 : 0;
        for (int p = 0; p < npos; p++) cnt[hash3(b + p) + 1]++;
        bstart[0] = 0; for (int h = 0; h < 32768; h++) bstart[h + 1] = bstart[h] + cnt[h + 1];
        memcpy(fill, bstart, sizeof fill);
        for (int p = 0; p < npos; p++) { const unsigned h = hash3(b + p); idx[p] = fill[h]; rnk[p] = fill[h] - bstart[h]; S[fill[h]++] = p; }
        reference_parse();
        window_parse();
        if (ntok != ref_ntok || memcmp(tok, ref_tok, sizeof(uint32_t) * (size_t)ntok)) {
            bad++; int k = 0; while (k < ntok && k < ref_ntok && tok[k] == ref_tok[k]) k++;
            printf("chunk %ld: MISMATCH at token %d (ntok %d / %d)\n", c, k, ntok, ref_ntok);
        }
    }
    printf("level %d kind %d depth %d n %d base %d: %s; per chunk: %.0f windows, %.2f evaluations a window, %.1f tokens, %.1f extensions, %.1f slow searches, %.1f stale stops\n",
           level, kind, DEPTH, n, base, bad ? "MISMATCH" : "all chunks exact", st_windows / (double)nch, (double)st_evals / st_windows, st_tokens / (double)nch,
           st_ext / (double)nch, st_slow / (double)nch, st_stale / (double)nch);
    return bad != 0;
}
