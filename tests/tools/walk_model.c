/* Experiment ("demand-driven search") -- NOT part of the product or of the parity tests.
 *
 * How much search work does a parse-driven level 4-9 path need if a chunk is cut into blocks of B positions and one
 * walker per block runs deflate_slow's loop (deflate.c:1554-1674) from the block's first position in the neutral state
 * (no match in hand), searching on demand, until it reaches a position another walker has already been at in the same
 * state?  Counts, per chunk: searches, chain steps, candidates that pass the quick check, for the exact parse and for
 * the block-walk scheme; and, for waves of 64 walkers stepping one candidate per iteration, the lane utilisation.
 * The window slide and the NIL corner are left out (they move a handful of positions; this program only counts).
 *
 *   gcc -O2 -o walk_model walk_model.c && ./walk_model [first_chunk nchunks level block]
 */
#include "../../zlib_amd/csrc/corpus.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_DIST 32506
#define TOO_FAR 4096
#define N 65536

static const int cfg[10][4] = {{0,0,0,0},{4,4,8,4},{4,5,16,8},{4,6,32,32},{4,4,16,16},{8,16,32,32},{8,16,128,128},{8,32,128,256},{32,128,258,1024},{32,258,258,4096}};
static int good, lazy, nice_, chain_;
static uint8_t b[N + 300];
static int linkp[N + 1];
static long c_search, c_steps, c_pass, c_cmpbytes;

/* longest_match (deflate.c:1027-1168) on static chains; counts */
static int lm(int n, int p, int prev_length, int *mstart)
{
    int best = prev_length, chain_length = chain_, look = n - p, nice = nice_;
    if (prev_length >= good) chain_length >>= 2;
    if (nice > look) nice = look;
    const int limit = p > MAX_DIST ? p - MAX_DIST : 0;
    int cur = linkp[p];
    c_search++;
    do {
        c_steps++;
        if (b[cur + best] != b[p + best] || b[cur + best - 1] != b[p + best - 1] || b[cur] != b[p] || b[cur + 1] != b[p + 1]) continue;
        c_pass++;
        int len = 2;
        const int maxl = look < 258 ? look : 258;
        while (len < maxl && b[cur + len] == b[p + len]) len++;
        c_cmpbytes += len;
        if (len > best) { best = len; *mstart = cur; if (len >= nice) break; }
    } while ((cur = linkp[cur]) > limit && --chain_length != 0);
    return best < look ? best : look;
}

/* One step of the loop from a neutral position s: returns the next neutral position; *searches_here counts. */
static int game(int n, int s)
{
    int match_length = 2, match_start = 0, match_available = 0;
    for (;;) {
        if (s >= n) return n;
        const int hash_head = (s + 3 <= n) ? linkp[s] : 0;
        const int prev_length = match_length;
        match_length = 2;
        if (hash_head != 0 && prev_length < lazy && s - hash_head <= MAX_DIST) {
            match_length = lm(n, s, prev_length, &match_start);
            if (match_length == 3 && s - match_start > TOO_FAR) match_length = 2;
        }
        if (prev_length >= 3 && match_length <= prev_length) return s - 1 + prev_length; /* match emitted: neutral behind it */
        if (match_available) { s++; if (match_length < 3) return s; } /* literal emitted; nothing in hand: neutral again */
        else { match_available = 1; s++; if (match_length < 3) return s; }
    }
}

int main(int argc, char **argv)
{
    const long first = argc > 1 ? atol(argv[1]) : 0, nch = argc > 2 ? atol(argv[2]) : 32;
    const int level = argc > 3 ? atoi(argv[3]) : 6, B = argc > 4 ? atoi(argv[4]) : 64;
    good = cfg[level][0]; lazy = cfg[level][1]; nice_ = cfg[level][2]; chain_ = cfg[level][3];
    static int head[32768];
    static uint8_t neu[N + 1], neu2[N + 1];
    long T_search = 0, T_steps = 0, T_pass = 0, W_search = 0, W_steps = 0, W_pass = 0, W_ext_search = 0, W_maxlane = 0, W_cmp = 0;
    long wave_iter1 = 0, lane_iter1 = 0, wave_iter2 = 0, lane_iter2 = 0, ext_nodes = 0, ext_max = 0, allpos_steps = 0;
    const int nwalk = N / B;
    long *cost1 = calloc(nwalk, sizeof(long)), *cost2 = calloc(nwalk, sizeof(long));
    const long OVERHEAD = 6; /* iterations a search costs besides its candidates (transition, loads), a guess */
    for (long c = first; c < first + nch; c++) {
        zc_fill_chunk(0, 0x5EED5117ull, (uint64_t)c, b);
        memset(b + N, 0, 300);
        const int n = N;
        memset(head, 0, sizeof head);
        for (int p = 0; p + 3 <= n; p++) {
            const unsigned h = (((unsigned)(b[p] & 31) << 10) ^ ((unsigned)b[p + 1] << 5) ^ b[p + 2]) & 0x7fff;
            linkp[p] = head[h]; head[h] = p;
        }
        /* all-position search cost (what match3 does): chain steps with the full budget from every position */
        for (int p = 1; p + 3 <= n; p++) { int cur = linkp[p], k = chain_; const int limit = p > MAX_DIST ? p - MAX_DIST : 0; if (cur == 0) continue; do allpos_steps++; while ((cur = linkp[cur]) > limit && --k); }
        /* exact parse */
        c_search = c_steps = c_pass = c_cmpbytes = 0;
        for (int s = 0; s < n;) s = game(n, s);
        T_search += c_search; T_steps += c_steps; T_pass += c_pass;
        /* block walks, phase 1: walker w from w*B until it leaves its block; marks its neutral positions */
        memset(neu, 0, sizeof neu); memset(neu2, 0, sizeof neu2);
        static int exitp[N];
        c_search = c_steps = c_pass = c_cmpbytes = 0;
        for (int w = 0; w < nwalk; w++) {
            const long s0 = c_steps, q0 = c_search;
            int s = w * B;
            while (s < (w + 1) * B && s < n) { neu[s] = 1; s = game(n, s); }
            exitp[w] = s;
            cost1[w] = (c_steps - s0) + OVERHEAD * (c_search - q0);
        }
        const long p1_search = c_search;
        /* phase 2: every walker goes on until it stands on a position marked in phase 1 */
        for (int w = 0; w < nwalk; w++) {
            const long s0 = c_steps, q0 = c_search;
            int s = exitp[w]; long nodes = 0;
            while (s < n && !neu[s]) { neu2[s] = 1; s = game(n, s); nodes++; }
            ext_nodes += nodes; if (nodes > ext_max) ext_max = nodes;
            cost2[w] = (c_steps - s0) + OVERHEAD * (c_search - q0);
        }
        W_search += c_search; W_steps += c_steps; W_pass += c_pass; W_ext_search += c_search - p1_search; W_cmp += c_cmpbytes;
        for (int w0 = 0; w0 < nwalk; w0 += 64) {
            long m1 = 0, m2 = 0;
            for (int w = w0; w < w0 + 64 && w < nwalk; w++) { if (cost1[w] > m1) m1 = cost1[w]; if (cost2[w] > m2) m2 = cost2[w]; lane_iter1 += cost1[w]; lane_iter2 += cost2[w]; }
            wave_iter1 += m1; wave_iter2 += m2;
            if (m1 + m2 > W_maxlane) W_maxlane = m1 + m2;
        }
    }
    const double P = (double)nch * N;
    printf("level %d, %ld chunks from %ld, block %d\n", level, nch, first, B);
    printf("all positions, full budget : %.2f chain steps per input byte\n", allpos_steps / P);
    printf("exact parse                : %.3f searches, %.2f chain steps, %.3f quick-check passes per input byte\n", T_search / P, T_steps / P, T_pass / P);
    printf("block walks                : %.3f searches (%.3f in extensions), %.2f chain steps, %.3f passes, %.2f compared bytes per input byte\n", W_search / P, W_ext_search / P, W_steps / P, W_pass / P, W_cmp / P);
    printf("extension: %.2f neutral nodes per walker on average, longest %ld\n", (double)ext_nodes / (nch * nwalk), ext_max);
    printf("lockstep waves of 64 walkers (a search = its candidates + %ld iterations):\n", OVERHEAD);
    printf("  phase 1: %.1f wave-iterations per chunk, lane utilisation %.1f%%\n", (double)wave_iter1 / nch, 100.0 * lane_iter1 / (64.0 * wave_iter1));
    printf("  phase 2: %.1f wave-iterations per chunk, lane utilisation %.1f%%\n", (double)wave_iter2 / nch, 100.0 * lane_iter2 / (64.0 * (wave_iter2 ? wave_iter2 : 1)));
    printf("  longest wave: %ld iterations; match3 today: %.0f wave-steps per chunk\n", W_maxlane, allpos_steps / P * N / 64 / 0.77);
    return 0;
}
