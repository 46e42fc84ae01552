/* Experiment ("demand-driven search") -- NOT part of the product or of the parity tests.
 *
 * Timing model of the walker kernel: LANES walkers per chunk in waves of 64, each walker a lane that runs deflate_slow's
 * loop (deflate.c:1554-1674) from the first position of a block of B positions until it stands, with no match in
 * hand, on a position some walker has been at in that state; blocks are handed out in order.  A wave advances in
 * "bodies" of four candidate steps (one group load of four candidates per lane and body, as in match3_kernel); searches
 * start and end at body boundaries: a lane whose search is over waits for the wave's next transition pass (every
 * TRANS bodies), the pass plays the parse and starts the next search, whose candidates arrive DELAY1 bodies later if it is at
 * the next position (its index was fetched with the last one) and DELAY2 bodies later after a jump.
 * Output: bodies, fold/transition passes and lane utilisation per chunk -> an instruction estimate.
 *
 *   gcc -O2 -o walk_sim walk_sim.c && ./walk_sim [first nchunks level B LANES TRANS]
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
static long c_pass;

/* longest_match; *steps = candidates visited */
static int lm(int n, int p, int prev_length, int *mstart, int *steps)
{
    int best = prev_length, chain_length = chain_, look = n - p, nice = nice_, st = 0;
    if (prev_length >= good) chain_length >>= 2;
    if (nice > look) nice = look;
    const int limit = p > MAX_DIST ? p - MAX_DIST : 0;
    int cur = linkp[p];
    do {
        st++;
        if (b[cur + best] != b[p + best] || b[cur + best - 1] != b[p + best - 1] || b[cur] != b[p] || b[cur + 1] != b[p + 1]) continue;
        c_pass++;
        int len = 2;
        const int maxl = look < 258 ? look : 258;
        while (len < maxl && b[cur + len] == b[p + len]) len++;
        if (len > best) { best = len; *mstart = cur; if (len >= nice) break; }
    } while ((cur = linkp[cur]) > limit && --chain_length != 0);
    *steps = st;
    return best < look ? best : look;
}

enum { NEED_BLOCK, LIMBO, SEARCH, WAIT, DONE };
typedef struct { int st, pos, seed, hand_L, hand_m, left, delay, res; } Lane;
static uint8_t claimed[N + 1];
static int n_ = N, next_block, nblocks, Bsz;
static long searches, jumps;

static int searchable(int x) { return x + 3 <= n_ && linkp[x] != 0 && x - linkp[x] <= MAX_DIST; }
static void begin_search(Lane *l, int x, int seed, int jump, int D1, int D2)
{
    int ms = 0, steps = 0;
    l->pos = x; l->seed = seed;
    int len = lm(n_, x, seed, &ms, &steps);
    if (len == 3 && x - ms > TOO_FAR) len = 2;
    l->res = len > seed ? len : 2;
    l->left = steps; l->st = LIMBO; l->delay = jump ? D2 : D1;
    searches++; jumps += jump;
}
/* walker stands at x with nothing in hand */
static void at_neutral(Lane *l, int x, int jump, int D1, int D2)
{
    for (;;) {
        if (x >= n_ || claimed[x]) { l->st = NEED_BLOCK; return; }
        claimed[x] = 1;
        l->hand_L = 2;
        if (searchable(x)) { begin_search(l, x, 2, jump, D1, D2); return; }
        x++; /* a literal */
    }
}
static void transition(Lane *l, int D1, int D2)
{
    if (l->st == WAIT) {
        const int len = l->res, x = l->pos;
        if (l->hand_L == 2) {
            if (len < 3) { at_neutral(l, x + 1, 0, D1, D2); goto grab; }
            l->hand_L = len; l->hand_m = x;
        } else {
            if (len > l->hand_L) { l->hand_L = len; l->hand_m = x; }
            else { at_neutral(l, l->hand_m + l->hand_L, 1, D1, D2); goto grab; }
        }
        if (l->hand_L < lazy && searchable(x + 1)) begin_search(l, x + 1, l->hand_L, 0, D1, D2);
        else at_neutral(l, l->hand_m + l->hand_L, 1, D1, D2);
    }
grab:
    while (l->st == NEED_BLOCK) {
        if (next_block >= nblocks) { l->st = DONE; return; }
        at_neutral(l, Bsz * next_block++, 1, D1, D2);
    }
}

int main(int argc, char **argv)
{
    const long first = argc > 1 ? atol(argv[1]) : 0, nch = argc > 2 ? atol(argv[2]) : 16;
    const int level = argc > 3 ? atoi(argv[3]) : 6;
    Bsz = argc > 4 ? atoi(argv[4]) : 64;
    const int LANES = argc > 5 ? atoi(argv[5]) : 512, TRANS = argc > 6 ? atoi(argv[6]) : 1, D1 = 1, D2 = 2;
    good = cfg[level][0]; lazy = cfg[level][1]; nice_ = cfg[level][2]; chain_ = cfg[level][3];
    static int head[32768];
    const int nw = LANES / 64;
    Lane *L = calloc(LANES, sizeof(Lane));
    long bodies = 0, lane_steps = 0, passes = 0, pass_lanes = 0, folds = 0, wall = 0, tail_bodies = 0;
    searches = jumps = c_pass = 0;
    for (long c = first; c < first + nch; c++) {
        zc_fill_chunk(0, 0x5EED5117ull, (uint64_t)c, b);
        memset(b + N, 0, 300);
        memset(head, 0, sizeof head);
        for (int p = 0; p + 3 <= N; p++) {
            const unsigned h = (((unsigned)(b[p] & 31) << 10) ^ ((unsigned)b[p + 1] << 5) ^ b[p + 2]) & 0x7fff;
            linkp[p] = head[h]; head[h] = p;
        }
        memset(claimed, 0, sizeof claimed);
        nblocks = N / Bsz; next_block = 0;
        for (int i = 0; i < LANES; i++) { L[i].st = NEED_BLOCK; L[i].hand_L = 2; }
        for (int i = 0; i < LANES; i++) transition(&L[i], D1, D2);
        long t = 0; int live_waves = nw;
        while (live_waves) {
            live_waves = 0;
            for (int w = 0; w < nw; w++) {
                Lane *lw = L + 64 * w;
                int any = 0, waiting = 0;
                for (int i = 0; i < 64; i++) { if (lw[i].st != DONE) any = 1; if (lw[i].st == WAIT) waiting++; }
                if (!any) continue;
                live_waves++;
                if (waiting && t % TRANS == 0) { passes++; pass_lanes += waiting; folds++; for (int i = 0; i < 64; i++) if (lw[i].st == WAIT) transition(&lw[i], D1, D2); }
                bodies++;
                for (int i = 0; i < 64; i++) {
                    Lane *l = &lw[i];
                    if (l->st == LIMBO) { if (--l->delay <= 0) l->st = SEARCH; continue; }
                    if (l->st == SEARCH) { const int k = l->left < 4 ? l->left : 4; lane_steps += k; l->left -= k; if (l->left == 0) l->st = WAIT; }
                }
            }
            if (live_waves && live_waves < nw) tail_bodies += nw - live_waves;
            t++;
        }
        wall += t;
    }
    const double P = (double)nch * N;
    printf("level %d, block %d, %d lanes per chunk, transition pass every %d bodies\n", level, Bsz, LANES, TRANS);
    printf("per chunk: %.0f searches (%.3f per byte, %.0f%% after a jump), %.0f wave-bodies (%.0f wall bodies, %.0f idle-wave bodies in the tail), %.0f passes (%.1f lanes each)\n",
           searches / (double)nch, searches / P, 100.0 * jumps / searches, bodies / (double)nch, wall / (double)nch, tail_bodies / (double)nch, passes / (double)nch, (double)pass_lanes / passes);
    printf("lane utilisation of the candidate steps: %.1f%%;  quick-check passes %.3f per byte\n", 100.0 * lane_steps / (bodies * 256.0), c_pass / P);
    const double step_i = 25, pass_i = 140; /* guesses: instructions per candidate step, per fold+transition pass */
    const double instr = bodies / (double)nch * 4 * step_i + passes / (double)nch * pass_i;
    printf("estimate: %.0f wave-instructions per chunk (match3+parse2 today: ~1 850 000) -> x%.2f\n", instr, 1850000.0 / instr);
    return 0;
}
