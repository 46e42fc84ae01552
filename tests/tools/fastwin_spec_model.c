/* Model of the SPECULATIVE form of the window algorithm (tests/tools/fastwin_model.c has the plain one): several waves of a workgroup share one
 * chunk's ring and chain bits; wave w takes windows w, w+W, ...; it evaluates and walks its window AHEAD of the parse -- from a guessed entry
 * position, under whatever the bits of the windows in front happen to hold -- and when its turn comes (the windows in front are final, the true
 * entry is known) it only checks: walk again from the true entry over the lengths it has, bring its own bits in line, and let every lane compare
 * the bits it SAW with the bits as they are now over the range its search examined (both directions: a bit that went away, a bit that appeared).
 * Stale lanes are evaluated again, as in the plain form.  This model checks that the check is sufficient: the speculative phase is fed a wrong
 * entry and randomly damaged bits for the windows in front, and the tokens must still be the reference's.  NOT part of the product.
 *   gcc -O2 -o fastwin_spec_model fastwin_spec_model.c && ./fastwin_spec_model [first nchunks level kind n base seed] */
#include "../../zlib_amd/csrc/corpus.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define NMAX 65536
#define MAX_DIST 32506
#define DEPTH 32
#define EXT_STEPS 4
static const int cfg[4][4] = {{0,0,0,0},{4,4,8,4},{4,5,16,8},{4,6,32,32}};
static uint8_t b[NMAX + 600];
static int S[NMAX], idx[NMAX], rnk[NMAX], bstart[32769];
static uint8_t F[NMAX];
static int n, base, maxins, nice, chain, npos;
static uint32_t rng = 12345;
static uint32_t rnd(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }
static unsigned hash3(const uint8_t *p) { return (((unsigned)(p[0] & 31) << 10) ^ ((unsigned)p[1] << 5) ^ p[2]) & 0x7fff; }
static int lcp(int q, int p, int cap) { int l = 0; while (l < cap && b[q + l] == b[p + l]) l++; return l; }
static uint32_t ref_tok[NMAX]; static int ref_ntok; static uint8_t rf[NMAX];
static void reference_parse(void)
{
    memset(rf, 0, sizeof rf); ref_ntok = 0;
    for (int p = 0; p < n;) {
        int len = 2, mstart = 0; const int look = n - p;
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
            rf[p] = 1; len = first ? 2 : (best < look ? best : look);
        }
        if (len >= 3) { ref_tok[ref_ntok++] = (uint32_t)(len - 3) | ((uint32_t)(p - mstart) << 8); if (len <= maxins && look - len >= 3) for (int k = 1; k < len; k++) rf[p + k] = 1; p += len; }
        else { ref_tok[ref_ntok++] = b[p]; p++; }
    }
}
enum { R_LIT, R_MATCH, R_INC, R_LONGTERM };
typedef struct { int kind, len, mstart; uint32_t seen, range; } lane_res;
static uint32_t read_bits(int p) { uint32_t m = 0; const int i = idx[p], r = rnk[p] < DEPTH ? rnk[p] : DEPTH; for (int k = 0; k < r; k++) if (F[i - 1 - k]) m |= 1u << k; return m; }
static void eval_lane(int p, lane_res *o)
{
    const int i = idx[p], r = rnk[p], look = n - p;
    const int w = p + base, limit = w > MAX_DIST ? w - MAX_DIST : 0, cap = look < 258 ? look : 258, ni = nice < look ? nice : look;
    int best = 2, first = 1, stopped = 0, nsel = 0, term = 0, klast = -1;
    uint32_t m = read_bits(p);
    o->kind = R_LIT; o->len = 1; o->mstart = 0; o->seen = m; o->range = 0;
    while (m && nsel < chain) {
        const int k = __builtin_ctz(m); m &= m - 1; klast = k;
        const int q = S[i - 1 - k], wq = q + base;
        if (first) { if (wq <= 0 || w - wq > MAX_DIST) { stopped = 1; break; } first = 0; } else if (wq <= limit) { stopped = 1; break; }
        nsel++;
        int l = lcp(q, p, cap < nice ? cap : nice);
        if (l > best) { best = l; o->mstart = q; if (l >= ni) { stopped = 1; term = l < cap; break; } }
    }
    if (nsel == chain) stopped = 1;
    o->range = stopped ? (klast >= 31 ? ~0u : (2u << klast) - 1u) : ~0u; /* the search did not stop: any of the 32 bits matters */
    if (!stopped && r > DEPTH) { o->kind = R_INC; return; }
    if (!first && best >= 3) {
        o->kind = R_MATCH; o->len = best;
        if (term) { int l = best, it = 0; for (; it < EXT_STEPS; it++) { int d = 0; while (d < 8 && l + d < cap && b[o->mstart + l + d] == b[p + l + d]) d++; l += d; if (d < 8 || l >= cap) break; }
                    o->len = l; if (it == EXT_STEPS) o->kind = R_LONGTERM; }
    }
}
static void slow_lane(int p, lane_res *o)
{
    const int i = idx[p], r = rnk[p], look = n - p;
    const int w = p + base, limit = w > MAX_DIST ? w - MAX_DIST : 0, cap = look < 258 ? look : 258, ni = nice < look ? nice : look;
    int best = 2, first = 1, ch = chain;
    o->kind = R_LIT; o->len = 1; o->mstart = 0;
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
static long st_win, st_spec_evals, st_val_evals, st_val_rounds, st_slow_val;
/* the rounds of one window from `entry`; commit: tokens are written.  Returns the position behind the window's last token. */
static int rounds(int w0, int entry, int cross_short, lane_res *res, int have_evals, int commit, int *cs_out)
{
    int guard = 0;
    const int lend = n - w0 < 64 ? n - w0 : 64;
    /* the bits of the lanes in front of the entry */
    for (int L = 0; L < entry && L < 64; L++) if (w0 + L < npos) F[idx[w0 + L]] = (uint8_t)(cross_short != 0);
    int start = entry, need_eval = !have_evals, cs = cross_short, pos = w0 + entry;
    if (entry >= lend) { /* nothing to walk; ahead of the parse the lanes are evaluated all the same (the true entry may lie in front) */
        if (!have_evals) for (int L = 0; L < 64; L++) { const int p = w0 + L; if (p < npos && rnk[p]) eval_lane(p, &res[L]); else { res[L].kind = R_LIT; res[L].len = 1; res[L].seen = 0; res[L].range = 0; } }
        *cs_out = cs; return pos;
    }
    for (;;) {
        if (++guard > 1000) { printf("no progress: w0 %d start %d commit %d\n", w0, start, commit); exit(2); }
        if (need_eval) {
            for (int L = (guard == 1 && !commit) ? 0 : start; L < 64; L++) { const int p = w0 + L; if (p < npos) { if (rnk[p]) eval_lane(p, &res[L]); else { res[L].kind = R_LIT; res[L].len = 1; res[L].seen = 0; res[L].range = 0; } }
                                               else { res[L].kind = R_LIT; res[L].len = 1; res[L].seen = 0; res[L].range = 0; } }
            if (commit) st_val_evals++; else st_spec_evals++;
        }
        if (commit) st_val_rounds++;
        /* the walk */
        uint64_t T = 0, C = 0; int L = start, stop_inc = 64, csn = cs;
        while (L < lend) {
            if (res[L].kind == R_INC) { stop_inc = L; break; }
            if (res[L].kind == R_LONGTERM) { const int p = w0 + L, look = n - p; res[L].len = lcp(res[L].mstart, p, look < 258 ? look : 258); res[L].kind = R_MATCH; }
            T |= 1ull << L;
            const int len = res[L].len;
            if (res[L].kind == R_MATCH) {
                const int sh = len <= maxins && n - (w0 + L) - len >= 3;
                if (!sh) for (int k = 1; k < len && L + k < 64; k++) C |= 1ull << (L + k);
                if (L + len >= 64) csn = sh;
            }
            L += len;
        }
        /* own bits as this walk wants them */
        for (int j = start; j < 64; j++) if (w0 + j < npos) F[idx[w0 + j]] = (uint8_t)!((C >> j) & 1);
        /* stale lanes: what they saw against what is there, over what they examined */
        int Lstale = 64;
        for (int j = start; j < lend; j++) if (((T >> j) & 1) && w0 + j < npos && rnk[w0 + j]) { if ((res[j].seen ^ read_bits(w0 + j)) & res[j].range) { Lstale = j; break; } }
        const int Lacc = Lstale < stop_inc ? Lstale : stop_inc;
        if (commit) for (int j = start; j < (Lacc < lend ? Lacc : lend); j++) if ((T >> j) & 1) {
            const int p = w0 + j;
            tok[ntok++] = res[j].kind == R_MATCH ? (uint32_t)(res[j].len - 3) | ((uint32_t)(p - res[j].mstart) << 8) : b[p];
        }
        if (Lacc >= lend) { pos = w0 + L; cs = csn; break; }
        for (int j = Lacc; j < 64; j++) if (w0 + j < npos) F[idx[w0 + j]] = 1; /* guesses again */
        start = Lacc;
        if (Lstale <= stop_inc) { need_eval = 1; continue; }
        if (!commit) { pos = w0 + Lacc; break; } /* ahead of the parse the bits below are not final: the whole-bucket search waits for the turn */
        { lane_res o; slow_lane(w0 + Lacc, &o); o.seen = 0; o.range = 0; res[Lacc] = o; st_slow_val++; }
        need_eval = 0;
    }
    *cs_out = cs;
    return pos;
}
int main(int argc, char **argv)
{
    const long first = argc > 1 ? atol(argv[1]) : 0, nch = argc > 2 ? atol(argv[2]) : 16;
    const int level = argc > 3 ? atoi(argv[3]) : 1, kind = argc > 4 ? atoi(argv[4]) : 0;
    n = argc > 5 ? atoi(argv[5]) : NMAX; base = argc > 6 ? atoi(argv[6]) : 0; rng = argc > 7 ? (uint32_t)atoi(argv[7]) : 1u;
    const int damage = argc > 8 ? atoi(argv[8]) : 30; /* percent of windows whose speculation sees damaged bits / a wrong entry */
    maxins = cfg[level][1]; nice = cfg[level][2]; chain = cfg[level][3];
    long bad = 0;
    for (long c = first; c < first + nch; c++) {
        if (kind < 2) zc_fill_chunk(kind, kind ? 0x10C7E47ull : 0x5EED5117ull, (uint64_t)c, b);
        else { uint32_t x = (uint32_t)c * 2654435761u + 12345u;
            for (int i = 0; i < NMAX; i++) { x = x * 1664525u + 1013904223u; const uint32_t r = x >> 8;
                b[i] = kind == 2 ? (uint8_t)("ab"[(r >> 3) & 1]) : kind == 3 ? (uint8_t)((i % ((int)(c % 7) + 1)) + 'a') : (uint8_t)((r % 5 == 0) ? 'x' : 'a' + (r >> 5) % 3); } }
        memset(b + n, 0, 600);
        static int cnt[32769], fill[32768];
        memset(cnt, 0, sizeof cnt);
        npos = n >= 3 ? n - 2 : 0;
        for (int p = 0; p < npos; p++) cnt[hash3(b + p) + 1]++;
        bstart[0] = 0; for (int h = 0; h < 32768; h++) bstart[h + 1] = bstart[h] + cnt[h + 1];
        memcpy(fill, bstart, sizeof fill);
        for (int p = 0; p < npos; p++) { const unsigned h = hash3(b + p); idx[p] = fill[h]; rnk[p] = fill[h] - bstart[h]; S[fill[h]++] = p; }
        reference_parse();
        memset(F, 0, sizeof F); ntok = 0;
        int pos = 0, cs = 0;
        for (int w0 = 0; w0 < n; w0 += 64) {
            st_win++;
            lane_res res[64];
            /* ---- ahead of the parse: a guessed entry, the bits of the windows in front possibly wrong ---- */
            static uint8_t keep[NMAX];
            const int lo = w0 >= 384 ? w0 - 384 : 0;
            int nk = 0;
            const int dmg = (int)(rnd() % 100) < damage;
            for (int p = lo; p < w0 && p < npos; p++) { keep[nk++] = F[idx[p]]; if (dmg && rnd() % 4 == 0) F[idx[p]] = (uint8_t)(rnd() & 1); }
            int e_guess = pos >= w0 + 64 ? 0 : pos - w0, cs_guess = cs;
            if (dmg) { e_guess = (int)(rnd() % 64); cs_guess = (int)(rnd() & 1); }
            for (int L = 0; L < 64; L++) if (w0 + L < npos) F[idx[w0 + L]] = 1;
            int dummy;
            (void)rounds(w0, e_guess, cs_guess, res, 0, 0, &dummy);
            nk = 0;
            for (int p = lo; p < w0 && p < npos; p++) F[idx[p]] = keep[nk++];
            /* ---- its turn: the true entry, the windows in front final ---- */
            if (pos >= w0 + 64) { for (int L = 0; L < 64; L++) if (w0 + L < npos) F[idx[w0 + L]] = (uint8_t)(cs != 0); continue; }
            pos = rounds(w0, pos - w0, cs, res, 1, 1, &cs);
        }
        if (ntok != ref_ntok || memcmp(tok, ref_tok, sizeof(uint32_t) * (size_t)ntok)) {
            bad++; int k = 0; while (k < ntok && k < ref_ntok && tok[k] == ref_tok[k]) k++;
            printf("chunk %ld: MISMATCH at token %d (ntok %d / %d)\n", c, k, ntok, ref_ntok);
        }
    }
    printf("level %d kind %d n %d base %d damage %d%%: %s; per window: %.2f evaluations ahead, %.2f rounds and %.2f evaluations at the turn; %.1f whole-bucket searches at the turn per chunk\n",
           level, kind, n, base, damage, bad ? "MISMATCH" : "all chunks exact", (double)st_spec_evals / st_win, (double)st_val_rounds / st_win, (double)st_val_evals / st_win, st_slow_val / (double)nch);
    return bad != 0;
}
