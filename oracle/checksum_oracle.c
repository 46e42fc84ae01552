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
