/* Experiment for the next round -- NOT part of the product or of the parity tests.
 * Which positions of a 64 KiB chunk does deflate_slow hand to longest_match (deflate.c:1588-1595), and how well does a cheap
 * parse predict that set?  A compact model of the level 4-9 path with static hash chains (the window
 * slide and the NIL corner are left out: they move a handful of positions and this program only counts).
 *   searched_stats(buf, n, good, lazy, nice, chain, depth, out[6]):
 *     out[0] positions searched by the exact parse, out[1] by the parse whose searches stop after `depth` candidates,
 *     out[2] searched by the exact parse but not by the cheap one (misses), out[3] the same after widening the cheap set by
 *     one position on each side, out[4] size of that widened set, out[5] chain steps of the exact parse's searches. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_DIST 32506
#define TOO_FAR 4096

static long steps;

static int lm(const uint8_t *b, int n, const int *link, int p, int prev_length, int chain, int good, int nice, int *mstart)
{
    int best = prev_length, chain_length = chain, look = n - p;
    if (prev_length >= good) chain_length >>= 2;
    if (chain_length < 1) chain_length = 1;
    if (nice > look) nice = look;
    const int limit = p > MAX_DIST ? p - MAX_DIST : 0;
    int cur = link[p];
    do {
        steps++;
        int len = 0;
        const int maxl = look < 258 ? look : 258;
        while (len < maxl && b[cur + len] == b[p + len]) len++;
        if (len > best) { best = len; *mstart = cur; if (len >= nice) break; }
    } while ((cur = link[cur]) > limit && --chain_length != 0);
    return best < look ? best : look;
}

static void parse(const uint8_t *b, int n, const int *link, int good, int lazy, int nice, int chain, uint8_t *searched)
{
    int s = 0, match_length = 2, match_start = 0, match_available = 0;
    memset(searched, 0, (size_t)n);
    while (s < n) {
        const int hash_head = (s + 3 <= n) ? link[s] : 0;
        const int prev_length = match_length;
        match_length = 2;
        if (hash_head != 0 && prev_length < lazy && s - hash_head <= MAX_DIST) {
            searched[s] = 1;
            match_length = lm(b, n, link, s, prev_length, chain, good, nice, &match_start);
            if (match_length <= 5 && match_length == 3 && s - match_start > TOO_FAR) match_length = 2;
        }
        if (prev_length >= 3 && match_length <= prev_length) { s += prev_length - 1; match_available = 0; match_length = 2; }
        else if (match_available) s++;
        else { match_available = 1; s++; }
    }
}

void searched_stats(const uint8_t *buf, int n, int good, int lazy, int nice, int chain, int depth, long *out)
{
    int *head = calloc(32768, sizeof(int)), *link = calloc((size_t)n + 1, sizeof(int));
    uint8_t *a = malloc((size_t)n), *c = malloc((size_t)n);
    for (int p = 0; p + 3 <= n; p++) { /* link(p): the previous position with the same 3-byte hash; 0 = none (NIL) */
        const unsigned h = (((unsigned)(buf[p] & 31) << 10) ^ ((unsigned)buf[p + 1] << 5) ^ buf[p + 2]) & 0x7fff;
        link[p] = head[h]; head[h] = p;
    }
    steps = 0;
    parse(buf, n, link, good, lazy, nice, chain, a);
    out[5] = steps;
    parse(buf, n, link, good, lazy, nice, depth, c);
    out[0] = out[1] = out[2] = out[3] = out[4] = 0;
    for (int p = 0; p < n; p++) {
        const int wide = c[p] || (p > 0 && c[p - 1]) || (p + 1 < n && c[p + 1]);
        out[0] += a[p]; out[1] += c[p]; out[2] += a[p] && !c[p]; out[3] += a[p] && !wide; out[4] += wide;
    }
    free(head); free(link); free(a); free(c);
}
