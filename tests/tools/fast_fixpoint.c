/* Experiment (levels 1-3, deflate_fast, deflate.c:1448-1546) -- NOT part of the product or of the parity tests.
 * deflate_fast does not insert the positions inside a match longer than max_insert_length into the hash chains, so its chains depend
 * on its parse, and the parse is sequential.  Could the pair (parse, inserted set) be found as a fixed point instead?  Start with
 * "every position is inserted", parse with the chains that set implies (every search independent of the others: parallel work),
 * take the inserted set the parse produces, repeat until it stops changing.  The fixed point is unique and is the reference's parse
 * (a decision at s depends on the inserted flags below s only).  This program counts the iterations.
 *   gcc -O2 -o fast_fixpoint fast_fixpoint.c && ./fast_fixpoint [first nchunks level kind] */
#include "../../zlib_amd/csrc/corpus.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define N 65536
#define MAX_DIST 32506
static const int cfg[4][4] = {{0,0,0,0},{4,4,8,4},{4,5,16,8},{4,6,32,32}}; /* good, max_insert, nice, chain */
static uint8_t b[N + 300], flag[N], nflag[N];
static int S[N], idx[N], bstart[32769];
static unsigned hash3(const uint8_t *p) { return (((unsigned)(p[0] & 31) << 10) ^ ((unsigned)p[1] << 5) ^ p[2]) & 0x7fff; }
static int lm(int s, int chain, int nice, int *mstart)
{ /* candidates: earlier positions with the same hash that are flagged, nearest first */
    int best = 2, look = N - s, i = idx[s] - 1;
    const unsigned h = hash3(b + s);
    if (nice > look) nice = look;
    const int limit = s > MAX_DIST ? s - MAX_DIST : 0;
    int first = 1;
    for (; i >= bstart[h]; i--) {
        const int q = S[i];
        if (!flag[q]) continue;
        if (first) { if (q == 0 || s - q > MAX_DIST) break; first = 0; } else if (q <= limit) break;
        int len = 0; const int maxl = look < 258 ? look : 258;
        while (len < maxl && b[q + len] == b[s + len]) len++;
        if (len > best) { best = len; *mstart = q; if (len >= nice) break; }
        if (--chain == 0) break;
    }
    return best < look ? best : look;
}
int main(int argc, char **argv)
{
    const long first = argc > 1 ? atol(argv[1]) : 0, nch = argc > 2 ? atol(argv[2]) : 16;
    const int level = argc > 3 ? atoi(argv[3]) : 1, kind = argc > 4 ? atoi(argv[4]) : 0;
    const int maxins = cfg[level][1], nice = cfg[level][2], chain = cfg[level][3];
    long tot_iter = 0, max_iter = 0;
    for (long c = first; c < first + nch; c++) {
        zc_fill_chunk(kind, kind ? 0x10C7E47ull : 0x5EED5117ull, (uint64_t)c, b);
        memset(b + N, 0, 300);
        static int cnt[32769];
        memset(cnt, 0, sizeof cnt);
        for (int p = 0; p + 3 <= N; p++) cnt[hash3(b + p) + 1]++;
        bstart[0] = 0; for (int h = 0; h < 32768; h++) bstart[h + 1] = bstart[h] + cnt[h + 1];
        static int fill[32768]; memcpy(fill, bstart, sizeof fill);
        for (int p = 0; p + 3 <= N; p++) { const unsigned h = hash3(b + p); idx[p] = fill[h]; S[fill[h]++] = p; }
        memset(flag, 1, N);
        int it = 0, prefix_ok = 0;
        for (;;) {
            memset(nflag, 0, N);
            for (int s = 0; s < N;) { /* deflate_fast with the chains `flag` implies */
                int len = 2, ms = 0;
                if (s + 3 <= N) { nflag[s] = 1; len = lm(s, chain, nice, &ms); }
                if (len >= 3) {
                    if (len <= maxins && s + len + 2 < N + 3) for (int k = 1; k < len; k++) if (s + k + 3 <= N) nflag[s + k] = 1;
                    s += len;
                } else s++;
            }
            it++;
            int same = 1, firstdiff = N;
            for (int p = 0; p < N; p++) if (flag[p] != nflag[p]) { same = 0; firstdiff = p; break; }
            if (it <= 12 && c == first) printf("  chunk %ld iteration %d: flags agree up to position %d\n", c, it, firstdiff);
            prefix_ok = firstdiff;
            if (same) break;
            memcpy(flag, nflag, N);
            if (it > 4000) break;
        }
        (void)prefix_ok;
        tot_iter += it; if (it > max_iter) max_iter = it;
    }
    printf("level %d kind %d: %.1f iterations per chunk on average, %ld at most\n", level, kind, (double)tot_iter / nch, max_iter);
    return 0;
}
