/* Experiment (levels 1-3, deflate_fast, deflate.c:1448-1546) -- NOT part of the product or of the parity tests.
 * Counts for a window-speculative form of deflate_fast: the chunk in windows of W positions, every lane prepares the search of its own
 * position from data that does not depend on the parse (the earlier positions with the same hash, nearest first, and the match length with each),
 * a scalar walk over the real token starts then applies what does depend on it (which of them are in the chains).
 * Per token start of the reference's parse: how many bucket predecessors the search looks at (inserted or not) before it ends, how often one of
 * them lies in the same window, tokens per chunk.
 *   gcc -O2 -o fast_window_model fast_window_model.c && ./fast_window_model [first nchunks level kind W] */
#include "../../zlib_amd/csrc/corpus.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define N 65536
#define MAX_DIST 32506
static const int cfg[4][4] = {{0,0,0,0},{4,4,8,4},{4,5,16,8},{4,6,32,32}}; /* good, max_insert, nice, chain */
static uint8_t b[N + 300], flag[N];
static int S[N], idx[N], bstart[32769];
static long h_exam[66], n_search, n_inwin, n_tok, n_lit, sum_exam, sum_cmp, n_ins;
static unsigned hash3(const uint8_t *p) { return (((unsigned)(p[0] & 31) << 10) ^ ((unsigned)p[1] << 5) ^ p[2]) & 0x7fff; }
static int lm(int s, int chain, int nice, int *mstart, int wbase)
{
    int best = 2, look = N - s, i = idx[s] - 1, exam = 0, inwin = 0;
    const unsigned h = hash3(b + s);
    if (nice > look) nice = look;
    const int limit = s > MAX_DIST ? s - MAX_DIST : 0;
    int first = 1;
    for (; i >= bstart[h]; i--) {
        const int q = S[i];
        exam++;
        if (q >= wbase) inwin = 1;
        if (!flag[q]) continue;
        if (first) { if (q == 0 || s - q > MAX_DIST) break; first = 0; } else if (q <= limit) break;
        int len = 0; const int maxl = look < 258 ? look : 258;
        while (len < maxl && b[q + len] == b[s + len]) len++;
        sum_cmp++;
        if (len > best) { best = len; *mstart = q; if (len >= nice) break; }
        if (--chain == 0) break;
    }
    n_search++; sum_exam += exam; h_exam[exam > 65 ? 65 : exam]++; n_inwin += inwin;
    return best < look ? best : look;
}
int main(int argc, char **argv)
{
    const long first = argc > 1 ? atol(argv[1]) : 0, nch = argc > 2 ? atol(argv[2]) : 16;
    const int level = argc > 3 ? atoi(argv[3]) : 1, kind = argc > 4 ? atoi(argv[4]) : 0, W = argc > 5 ? atoi(argv[5]) : 64;
    const int maxins = cfg[level][1], nice = cfg[level][2], chain = cfg[level][3];
    for (long c = first; c < first + nch; c++) {
        zc_fill_chunk(kind, kind ? 0x10C7E47ull : 0x5EED5117ull, (uint64_t)c, b);
        memset(b + N, 0, 300);
        static int cnt[32769];
        memset(cnt, 0, sizeof cnt);
        for (int p = 0; p + 3 <= N; p++) cnt[hash3(b + p) + 1]++;
        bstart[0] = 0; for (int h = 0; h < 32768; h++) bstart[h + 1] = bstart[h] + cnt[h + 1];
        static int fill[32768]; memcpy(fill, bstart, sizeof fill);
        for (int p = 0; p + 3 <= N; p++) { const unsigned h = hash3(b + p); idx[p] = fill[h]; S[fill[h]++] = p; }
        memset(flag, 0, N);
        for (int s = 0; s < N;) { /* the reference's loop; flags are final below s */
            int len = 2, ms = 0;
            if (s + 3 <= N) { len = lm(s, chain, nice, &ms, s / W * W); flag[s] = 1; n_ins++; }
            n_tok++;
            if (len >= 3) {
                if (len <= maxins) for (int k = 1; k < len; k++) if (s + k + 3 <= N) { flag[s + k] = 1; n_ins++; }
                s += len;
            } else { s++; n_lit++; }
        }
    }
    const double P = (double)nch * N;
    printf("level %d kind %d W %d: %.0f tokens per chunk (%.0f literals), %.3f searches per byte, inserted %.1f%% of positions\n", level, kind, W, n_tok / (double)nch, n_lit / (double)nch, n_search / P, 100.0 * n_ins / P);
    printf("bucket predecessors looked at per search: mean %.2f (string compares %.2f); a predecessor in the same window: %.1f%% of searches\n", (double)sum_exam / n_search, (double)sum_cmp / n_search, 100.0 * n_inwin / n_search);
    long acc = 0;
    printf("searches that end within M predecessors:");
    for (int m = 0; m <= 65; m++) { acc += h_exam[m]; if (m == 0 || m == 1 || m == 2 || m == 4 || m == 8 || m == 16 || m == 32 || m == 64) printf("  M=%d %.2f%%", m, 100.0 * acc / n_search); }
    printf("\n");
    return 0;
}
