/* corpus.h -- seeded synthetic corpora for the benchmark configurations.
 *
 * Measurement harness, not codec: every 64 KiB chunk is a pure function of (kind, seed, chunk index),
 * computed with integer arithmetic only, so the host build (gcc, used for golden fixtures and the CPU
 * baseline) and the device build (one lane per chunk, used by bench.py to fill HBM) produce identical
 * bytes.  tests/test_corpus.py checks that identity.
 *
 * kind 0  "silesia-mix": 1 MiB segments (16 chunks) cycling through 20 slots: 9 English-like text
 *          (Zipf-ish ranks over a 50 000-word procedural vocabulary), 3 markup (XML/JSON-ish), 3 log
 *          lines, 2 source-code-like, 1 numeric table, 1 low-entropy binary, 1 high-entropy bytes.
 * kind 1  "log-text": timestamped log lines only (config 5).
 */
#ifndef ZAMD_CORPUS_H
#define ZAMD_CORPUS_H
#include <stdint.h>

#ifdef __HIPCC__
#define ZC_FN __host__ __device__ static inline
#else
#define ZC_FN static inline
#endif

#define ZC_CHUNK 65536u
#define ZC_SEED_SILESIA 0x5EED5117ull
#define ZC_SEED_LOGTEXT 0x10C7E47ull

typedef struct {
    uint8_t *out;      /* chunk buffer (8-byte aligned) */
    uint32_t pos;      /* bytes emitted so far */
    uint64_t stage;    /* up to 8 bytes waiting to be stored */
    uint64_t rng;      /* xorshift64* state */
    uint32_t col;      /* current text column */
} zc_gen;

ZC_FN uint64_t zc_mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
ZC_FN uint32_t zc_next(zc_gen *g) { uint64_t x = g->rng; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; g->rng = x; return (uint32_t)((x * 0x2545F4914F6CDD1Dull) >> 32); }
ZC_FN uint32_t zc_below(zc_gen *g, uint32_t n) { return (uint32_t)(((uint64_t)zc_next(g) * n) >> 32); }

ZC_FN void zc_put(zc_gen *g, uint32_t c)
{
    if (g->pos >= ZC_CHUNK) return;
    g->stage |= (uint64_t)(c & 0xff) << ((g->pos & 7) * 8);
    g->pos++;
    if ((g->pos & 7) == 0) { *(uint64_t *)(g->out + g->pos - 8) = g->stage; g->stage = 0; }
    g->col = (c == '\n') ? 0 : g->col + 1;
}
ZC_FN void zc_puts(zc_gen *g, const char *s) { while (*s) zc_put(g, (uint8_t)*s++); }
ZC_FN void zc_putdec(zc_gen *g, uint32_t v, int mindigits)
{
    char tmp[10]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n < mindigits) tmp[n++] = '0';
    while (n) zc_put(g, (uint8_t)tmp[--n]);
}
ZC_FN void zc_puthex(zc_gen *g, uint32_t v, int digits) { for (int i = digits - 1; i >= 0; i--) zc_put(g, (uint8_t)"0123456789abcdef"[(v >> (4 * i)) & 15]); }

/* Zipf-ish rank in [0, 50000): octave k holds ranks [2^k-1, 2^(k+1)-1) with mass ~ sum r^-1.1 */
ZC_FN uint32_t zc_rank(zc_gen *g)
{
    /* cumulative octave masses scaled to 2^32 (s = 1.1, 16 octaves), computed offline */
    const uint32_t cum[16] = {
        0x2394493Du, 0x3ECDAAC9u, 0x55BEF1E5u, 0x6A1819EEu, 0x7C9BAA96u, 0x8DAAD34Eu, 0x9D7C2C91u, 0xAC3297FEu,
        0xB9E751DFu, 0xC6AE87BAu, 0xD2997D7Bu, 0xDDB79534u, 0xE816D2ACu, 0xF1C4210Du, 0xFACB7B37u, 0xFFFFFFFFu};
    uint32_t u = zc_next(g), k = 0;
    while (k < 15 && u > cum[k]) k++;
    if (k == 15) return 32767 + zc_below(g, 50000 - 32767); /* tail ranks */
    uint32_t lo = (1u << k) - 1, r = lo + zc_below(g, 1u << k);
    return r < 50000 ? r : r % 50000;
}

/* spelling of vocabulary word r: procedural, built from onset/vowel/coda syllables so that the
 * trigram statistics (and therefore hash-chain lengths) resemble natural-language text */
ZC_FN void zc_word(zc_gen *g, uint32_t r, int capital)
{
    const char *onset[16] = {"", "b", "c", "d", "f", "g", "h", "l", "m", "n", "p", "r", "s", "t", "st", "th"};
    const char *vowel[8] = {"a", "e", "i", "o", "u", "ea", "ou", "e"};
    const char *coda[8] = {"", "", "n", "r", "s", "t", "l", "ng"};
    uint64_t h = zc_mix(0x9E3779B97F4A7C15ull ^ (uint64_t)r * 0xD1B54A32D192ED03ull);
    uint32_t nsyl = r < 24 ? 1 : r < 512 ? 1 + (uint32_t)(h & 1) : r < 8192 ? 2 + (uint32_t)(h % 3 == 0) : 2 + (uint32_t)(h & 1) + (uint32_t)((h >> 1) & 1);
    h >>= 2;
    int first = 1;
    for (uint32_t i = 0; i < nsyl; i++) {
        const char *parts[3];
        parts[0] = onset[h & 15]; parts[1] = vowel[(h >> 4) & 7]; parts[2] = coda[(i + 1 == nsyl) ? ((h >> 7) & 7) : ((h >> 7) & 1) * 2];
        h = zc_mix(h + i + 1);
        for (int k = 0; k < 3; k++)
            for (const char *s = parts[k]; *s; s++) { char c = *s; if (capital && first) c = (char)(c - 32); first = 0; zc_put(g, (uint8_t)c); }
    }
}

/* English-like running text: Zipf-ish ranks plus short-range phrase repetition (topic locality) */
ZC_FN void zc_text(zc_gen *g)
{
    uint16_t ring[64]; uint32_t nring = 0, replay = 0, rpos = 0; int cap = 1;
    for (int i = 0; i < 64; i++) ring[i] = 0;
    while (g->pos < ZC_CHUNK) {
        uint32_t r;
        if (replay) { r = ring[rpos & 63]; rpos++; replay--; }
        else {
            if (nring >= 24 && zc_below(g, 5) == 0) { replay = 2 + zc_below(g, 5); rpos = nring - 2 - zc_below(g, 14) - replay; }
            r = zc_rank(g);
        }
        ring[nring & 63] = (uint16_t)r; nring++;
        zc_word(g, r, cap); cap = 0;
        uint32_t u = zc_below(g, 64);
        if (u < 4) zc_put(g, ',');
        else if (u < 8) { zc_put(g, u == 4 ? '?' : '.'); cap = 1; }
        else if (u == 8) zc_put(g, ';');
        if (g->col >= 68 + (u & 7)) { zc_put(g, '\n'); if (u < 8 && (u & 1)) zc_put(g, '\n'); }
        else zc_put(g, ' ');
    }
}

ZC_FN void zc_markup(zc_gen *g)
{
    const char *keys[8] = {"id", "name", "title", "value", "status", "owner", "created", "score"};
    uint32_t id = zc_next(g) % 900000;
    while (g->pos < ZC_CHUNK) {
        id += 1 + zc_below(g, 7);
        if (zc_below(g, 2)) {
            zc_puts(g, "  <record id=\""); zc_putdec(g, id, 1); zc_puts(g, "\" type=\""); zc_word(g, zc_below(g, 12), 0); zc_puts(g, "\">\n");
            uint32_t nf = 2 + zc_below(g, 5);
            for (uint32_t f = 0; f < nf; f++) {
                const char *k = keys[zc_below(g, 8)];
                zc_puts(g, "    <"); zc_puts(g, k); zc_put(g, '>');
                if (zc_below(g, 3) == 0) { zc_putdec(g, zc_below(g, 100000), 1); zc_put(g, '.'); zc_putdec(g, zc_below(g, 100), 2); }
                else { uint32_t nw = 1 + zc_below(g, 4); for (uint32_t w = 0; w < nw; w++) { if (w) zc_put(g, ' '); zc_word(g, zc_rank(g), w == 0); } }
                zc_puts(g, "</"); zc_puts(g, k); zc_puts(g, ">\n");
            }
            zc_puts(g, "  </record>\n");
        } else {
            zc_puts(g, "{\"id\": "); zc_putdec(g, id, 1);
            uint32_t nf = 2 + zc_below(g, 5);
            for (uint32_t f = 0; f < nf; f++) {
                zc_puts(g, ", \""); zc_puts(g, keys[zc_below(g, 8)]); zc_puts(g, "\": ");
                uint32_t t = zc_below(g, 4);
                if (t == 0) zc_putdec(g, zc_below(g, 1000000), 1);
                else if (t == 1) zc_puts(g, zc_below(g, 2) ? "true" : "false");
                else if (t == 2) { zc_put(g, '['); uint32_t m = 1 + zc_below(g, 5); for (uint32_t j = 0; j < m; j++) { if (j) zc_puts(g, ", "); zc_putdec(g, zc_below(g, 1000), 1); } zc_put(g, ']'); }
                else { zc_put(g, '"'); zc_word(g, zc_rank(g), 0); zc_put(g, ' '); zc_word(g, zc_rank(g), 0); zc_put(g, '"'); }
            }
            zc_puts(g, "}\n");
        }
    }
}

ZC_FN void zc_logs(zc_gen *g)
{
    const char *lvl[8] = {"INFO ", "INFO ", "INFO ", "INFO ", "DEBUG", "DEBUG", "WARN ", "ERROR"};
    uint32_t sec = zc_next(g) % 86400, ms = zc_below(g, 1000), day = 1 + zc_below(g, 28);
    uint32_t salt = zc_next(g); /* per-chunk pools of hosts / ids / paths */
    while (g->pos < ZC_CHUNK) {
        ms += zc_below(g, 40); if (ms >= 1000) { ms -= 1000; sec = (sec + 1) % 86400; }
        zc_puts(g, "2026-03-"); zc_putdec(g, day, 2); zc_put(g, 'T');
        zc_putdec(g, sec / 3600, 2); zc_put(g, ':'); zc_putdec(g, sec / 60 % 60, 2); zc_put(g, ':'); zc_putdec(g, sec % 60, 2);
        zc_put(g, '.'); zc_putdec(g, ms, 3); zc_puts(g, "Z ");
        zc_puts(g, lvl[zc_below(g, 8)]); zc_put(g, ' ');
        uint32_t comp = zc_rank(g) % 12; zc_word(g, 200 + comp, 0); zc_put(g, '['); zc_putdec(g, 1000 + comp * 37 % 9000, 1); zc_puts(g, "]: ");
        /* message template: Zipf-ish choice among 64 templates, each a fixed word sequence with slots */
        uint32_t t = zc_rank(g) & 63; uint64_t th = zc_mix(0xABCDEF12345ull + t);
        uint32_t nw = 5 + (uint32_t)(th & 7);
        for (uint32_t w = 0; w < nw; w++) {
            uint32_t sel = (uint32_t)((th >> (4 + 4 * w)) & 15);
            if (w) zc_put(g, ' ');
            if (sel == 0) { zc_puts(g, "n="); zc_putdec(g, zc_below(g, 2000), 1); }
            else if (sel == 1) { zc_puts(g, "id=0x"); zc_puthex(g, (uint32_t)zc_mix(salt + zc_below(g, 48)), 8); }
            else if (sel == 2 && w > 1) { uint32_t ip = (uint32_t)zc_mix(salt ^ (zc_below(g, 24) + 77)); zc_puts(g, "10."); zc_putdec(g, (ip >> 8) & 3, 1); zc_put(g, '.'); zc_putdec(g, (ip >> 16) & 15, 1); zc_put(g, '.'); zc_putdec(g, ip & 255, 1); }
            else if (sel == 3 && w > 1) { uint32_t pp = zc_below(g, 32); zc_puts(g, "/var/"); zc_word(g, 300 + (pp & 7), 0); zc_put(g, '/'); zc_word(g, 400 + pp, 0); zc_puts(g, ".dat"); }
            else zc_word(g, (uint32_t)(zc_mix(th + w) % 600), 0);
        }
        zc_put(g, '\n');
    }
}

ZC_FN void zc_code(zc_gen *g)
{
    const char *kw[12] = {"if", "for", "while", "return", "int", "void", "static", "const", "struct", "else", "size_t", "char"};
    uint32_t depth = 0;
    while (g->pos < ZC_CHUNK) {
        for (uint32_t i = 0; i < depth; i++) zc_puts(g, "    ");
        uint32_t u = zc_below(g, 16);
        if (u < 3 && depth < 6) {
            zc_puts(g, kw[zc_below(g, 3)]); zc_puts(g, " ("); zc_word(g, 500 + zc_below(g, 300), 0);
            zc_puts(g, u == 0 ? " < " : u == 1 ? " != " : " == "); zc_word(g, 500 + zc_below(g, 300), 0); zc_puts(g, ") {\n"); depth++;
        } else if (u < 6 && depth > 0) { zc_puts(g, "}\n"); depth--; }
        else if (u < 8) { zc_puts(g, kw[4 + zc_below(g, 8)]); zc_put(g, ' '); zc_word(g, 500 + zc_below(g, 1500), 0); zc_puts(g, " = "); zc_putdec(g, zc_below(g, 4096), 1); zc_puts(g, ";\n"); }
        else if (u < 10) { zc_puts(g, "/* "); uint32_t nw = 2 + zc_below(g, 8); for (uint32_t w = 0; w < nw; w++) { zc_word(g, zc_rank(g), 0); zc_put(g, ' '); } zc_puts(g, "*/\n"); }
        else if (u < 11) { zc_puts(g, "return "); zc_word(g, 500 + zc_below(g, 300), 0); zc_puts(g, ";\n"); }
        else {
            zc_word(g, 500 + zc_below(g, 1500), 0); zc_puts(g, u & 1 ? "->" : "."); zc_word(g, 500 + zc_below(g, 300), 0);
            zc_puts(g, " = "); zc_word(g, 800 + zc_below(g, 200), 0); zc_put(g, '('); zc_word(g, 500 + zc_below(g, 300), 0);
            if (u & 2) { zc_puts(g, ", "); zc_word(g, 500 + zc_below(g, 300), 0); zc_puts(g, " + "); zc_putdec(g, zc_below(g, 64), 1); }
            zc_puts(g, ");\n");
        }
    }
}

ZC_FN void zc_numeric(zc_gen *g)
{
    uint32_t base = zc_below(g, 100000), lvl[8];
    for (int c = 0; c < 8; c++) lvl[c] = zc_below(g, 100000);
    while (g->pos < ZC_CHUNK) {
        uint32_t cols = 6 + (zc_below(g, 16) == 0);
        base += 1 + zc_below(g, 3);
        for (uint32_t c = 0; c < cols; c++) {
            if (c) zc_put(g, c & 1 ? '\t' : ',');
            if (c == 0) zc_putdec(g, base, 6);
            else {
                lvl[c] = (lvl[c] + zc_below(g, 9) + 99996) % 100000; /* slow random walk */
                zc_putdec(g, lvl[c] / 10, 1); zc_put(g, '.'); zc_putdec(g, lvl[c] % 10 * 1000 + zc_below(g, c < 4 ? 10 : 1000), 4);
            }
        }
        zc_put(g, '\n');
    }
}

ZC_FN void zc_lowbin(zc_gen *g)
{
    while (g->pos < ZC_CHUNK) {
        uint32_t u = zc_below(g, 8);
        if (u < 3) { uint32_t run = 4 + zc_below(g, 120), v = zc_below(g, 4) * 85; for (uint32_t i = 0; i < run; i++) zc_put(g, v); }
        else { /* 16-byte record: small little-endian ints, mostly zero high bytes */
            uint32_t a = zc_below(g, 300), b = zc_below(g, 16), c = zc_next(g) & 0xffff;
            zc_put(g, a & 0xff); zc_put(g, a >> 8); zc_put(g, 0); zc_put(g, 0);
            zc_put(g, b); zc_put(g, 0); zc_put(g, 0); zc_put(g, 0x80);
            zc_put(g, c & 0xff); zc_put(g, c >> 8); zc_put(g, 0); zc_put(g, 0);
            zc_put(g, 0xff); zc_put(g, 0xff); zc_put(g, u); zc_put(g, 0);
        }
    }
}

ZC_FN void zc_random(zc_gen *g)
{
    while (g->pos < ZC_CHUNK) { uint32_t v = zc_next(g); zc_put(g, v); zc_put(g, v >> 8); zc_put(g, v >> 16); zc_put(g, v >> 24); }
}

/* class of a chunk of the silesia-mix: 0 text 1 markup 2 logs 3 code 4 numeric 5 lowbin 6 random */
ZC_FN uint32_t zc_class_of(uint64_t chunk_index)
{
    const uint8_t slots[20] = {0, 1, 0, 2, 0, 3, 0, 1, 2, 0, 4, 0, 3, 0, 1, 5, 0, 2, 0, 6};
    return slots[(chunk_index >> 4) % 20];
}

/* fill out[0..65536) (8-byte aligned) with chunk `chunk_index` of corpus `kind` */
ZC_FN void zc_fill_chunk(uint32_t kind, uint64_t seed, uint64_t chunk_index, uint8_t *out)
{
    zc_gen g; g.out = out; g.pos = 0; g.stage = 0; g.col = 0;
    g.rng = zc_mix(seed ^ zc_mix(chunk_index + 0x632BE59BD9B4E019ull)) | 1;
    uint32_t cls = kind == 1 ? 2u : zc_class_of(chunk_index);
    switch (cls) {
    case 0: zc_text(&g); break;
    case 1: zc_markup(&g); break;
    case 2: zc_logs(&g); break;
    case 3: zc_code(&g); break;
    case 4: zc_numeric(&g); break;
    case 5: zc_lowbin(&g); break;
    default: zc_random(&g); break;
    }
}
#endif
