/* checksum_oracle.c -- Adler-32 / CRC-32 restatement.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Parity: PINNED against oracle/_ref/libzref.so (tests/test_oracle_vs_reference.py).
 *   adler32          /root/reference/qcsrc/adler32.c:57-125   (BASE 65521, NMAX 5552 deferred modulo)
 *   adler32_combine  /root/reference/qcsrc/adler32.c:128-149
 *   crc32            /root/reference/qcsrc/crc32.c:219-251    (polynomial 0xedb88320, bit-serial here)
 */
#include "oracle.h"

#define ADLER_BASE 65521u
#define ADLER_NMAX 5552u

uint32_t ora_adler32(uint32_t adler, const uint8_t *buf, size_t len)
{
    uint32_t a = adler & 0xffff, b = (adler >> 16) & 0xffff;
    while (len) {
        size_t run = len < ADLER_NMAX ? len : ADLER_NMAX;
        len -= run;
        while (run--) { a += *buf++; b += a; }
        a %= ADLER_BASE; b %= ADLER_BASE;
    }
    return a | (b << 16);
}

uint32_t ora_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2)
{
    uint32_t rem = (uint32_t)(len2 % ADLER_BASE);
    uint32_t sum1 = adler1 & 0xffff, sum2 = (rem * sum1) % ADLER_BASE;
    sum1 += (adler2 & 0xffff) + ADLER_BASE - 1;
    sum2 += ((adler1 >> 16) & 0xffff) + ((adler2 >> 16) & 0xffff) + ADLER_BASE - rem;
    if (sum1 > ADLER_BASE) sum1 -= ADLER_BASE;
    if (sum1 > ADLER_BASE) sum1 -= ADLER_BASE;
    if (sum2 > (ADLER_BASE << 1)) sum2 -= (ADLER_BASE << 1);
    if (sum2 > ADLER_BASE) sum2 -= ADLER_BASE;
    return sum1 | (sum2 << 16);
}

uint32_t ora_crc32(uint32_t crc, const uint8_t *buf, size_t len)
{
    uint32_t c = crc ^ 0xffffffffu;
    while (len--) {
        c ^= *buf++;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1)));
    }
    return c ^ 0xffffffffu;
}

/* crc32_combine, crc32.c:343-423: the CRC register after len2 further zero bytes is a linear map of the register, applied
 * as a 32x32 bit matrix over GF(2) that is squared once per bit of len2 (operator for 1 zero bit -> 2 -> 4 -> 8 = one byte
 * -> 2 bytes ...), then crc2 is added. */
static uint32_t ora_gf2_apply(const uint32_t *mat, uint32_t vec)
{
    uint32_t sum = 0;
    for (int i = 0; vec; vec >>= 1, i++) if (vec & 1) sum ^= mat[i];
    return sum;
}
static void ora_gf2_square(uint32_t *dst, const uint32_t *mat)
{
    for (int n = 0; n < 32; n++) dst[n] = ora_gf2_apply(mat, mat[n]);
}
uint32_t ora_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    uint32_t even[32], odd[32];
    if (len2 == 0) return crc1;
    odd[0] = 0xedb88320u; /* one zero bit: shift right, feed the polynomial back */
    for (int n = 1; n < 32; n++) odd[n] = 1u << (n - 1);
    ora_gf2_square(even, odd); /* two zero bits */
    ora_gf2_square(odd, even); /* four */
    do {
        ora_gf2_square(even, odd); /* first pass: eight zero bits = one byte */
        if (len2 & 1) crc1 = ora_gf2_apply(even, crc1);
        len2 >>= 1;
        if (len2 == 0) break;
        ora_gf2_square(odd, even);
        if (len2 & 1) crc1 = ora_gf2_apply(odd, crc1);
        len2 >>= 1;
    } while (len2 != 0);
    return crc1 ^ crc2;
}
