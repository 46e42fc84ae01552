/* zamd_zip.c -- PKZIP bulk entry points (include/zamd_zip.h): the container around the chunked engine, host side only.
 * Layout of what is written follows /root/reference/qcsrc/zip.c: local header :822-858 (sizes and CRC zero, patched at :1104-1124),
 * central header :780-817 + :1090-1096, end record :1176-1206; flag bits from the level :760-766; internal attribute Z_ASCII when the
 * compressor said text :1092-1093.  Reading follows qcsrc/unzip.c: end record search :320-383, central directory :583-717, local
 * header check :983-1050. */
#include "../../include/zamd_zip.h"
#include "../../include/zamd_zlib.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

struct zamd_zip { FILE *fp; uint8_t *cdir; size_t cdir_len, cdir_cap; unsigned long entries; };

static void put16(uint8_t *p, unsigned long v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, unsigned long v) { put16(p, v); put16(p + 2, v >> 16); }
static unsigned long get16(const uint8_t *p) { return (unsigned long)p[0] | ((unsigned long)p[1] << 8); }
static unsigned long get32(const uint8_t *p) { return get16(p) | (get16(p + 2) << 16); }

EXPORT zamd_zip *zamd_zip_open(const char *path)
{
    if (!path) return NULL;
    zamd_zip *z = (zamd_zip *)calloc(1, sizeof *z);
    if (!z) return NULL;
    z->fp = fopen(path, "wb");
    if (!z->fp) { free(z); return NULL; }
    return z;
}

/* the member's raw deflate data, CRC-32 and data type in one pass over the GPU: a gzip member is header + raw data + CRC + length */
static int deflate_member(const uint8_t *data, unsigned long len, int level, uint8_t **out, size_t *out_len, unsigned long *crc, int *data_type)
{
    z_stream s; memset(&s, 0, sizeof s);
    int rc = deflateInit2(&s, level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
    if (rc != Z_OK) return ZAMD_ZIP_INTERNALERROR;
    size_t cap = (size_t)deflateBound(&s, len) + 64, have = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) { deflateEnd(&s); return ZAMD_ZIP_INTERNALERROR; }
    const unsigned long slice = 0xC0000000ul; /* avail_in is 32 bits wide (a multiple of the chunk size keeps the chunking) */
    unsigned long left = len;
    const uint8_t *src = data;
    for (;;) {
        const int last = left <= slice;
        s.next_in = (Bytef *)src; s.avail_in = (uInt)(last ? left : slice); src += s.avail_in; left -= s.avail_in;
        do {
            const size_t room = cap - have > 0xFFFFFFFFul ? 0xFFFFFFFFul : cap - have;
            s.next_out = buf + have; s.avail_out = (uInt)room;
            rc = deflate(&s, last ? Z_FINISH : Z_NO_FLUSH);
            have += room - s.avail_out;
        } while (rc == Z_OK && s.avail_out == 0);
        if (rc != Z_OK && rc != Z_STREAM_END) { free(buf); deflateEnd(&s); return ZAMD_ZIP_INTERNALERROR; }
        if (last) break;
    }
    *data_type = s.data_type;
    deflateEnd(&s);
    if (rc != Z_STREAM_END || have < 18) { free(buf); return ZAMD_ZIP_INTERNALERROR; }
    *crc = get32(buf + have - 8);
    *out = buf; *out_len = have;
    return ZAMD_ZIP_OK;
}

EXPORT int zamd_zip_add(zamd_zip *z, const char *name, const void *data, unsigned long len, int level, unsigned long dos_date, const char *comment)
{
    if (!z || !z->fp || (!data && len) || level < -1 || level > 9 || len >= 0xFFFFFFFFul) return ZAMD_ZIP_PARAMERROR;
    if (!name) name = "-";
    const size_t nname = strlen(name), ncomm = comment ? strlen(comment) : 0;
    if (nname > 0xFFFFu || ncomm > 0xFFFFu || z->entries >= 0xFFFFu) return ZAMD_ZIP_PARAMERROR; /* 16-bit fields (no zip64, as minizip 1.01e) */
    const int method = level != 0 ? Z_DEFLATED : 0;
    unsigned long flag = 0;
    if (level == 8 || level == 9) flag |= 2;
    if (level == 2) flag |= 4;
    if (level == 1) flag |= 6;
    uint8_t *gz = NULL; size_t gz_len = 0; unsigned long crc = 0; int data_type = Z_UNKNOWN;
    const uint8_t *body = (const uint8_t *)data; size_t body_len = len;
    if (method) {
        const int rc = deflate_member((const uint8_t *)data, len, level, &gz, &gz_len, &crc, &data_type);
        if (rc != ZAMD_ZIP_OK) return rc;
        body = gz + 10; body_len = gz_len - 18;
    } else {
        for (unsigned long o = 0; o < len; o += 0x40000000ul) crc = crc32(crc, (const Bytef *)data + o, (uInt)(len - o < 0x40000000ul ? len - o : 0x40000000ul));
    }
    const long pos = ftell(z->fp);
    uint8_t lh[30];
    put32(lh, 0x04034b50ul); put16(lh + 4, 20); put16(lh + 6, flag); put16(lh + 8, (unsigned long)method); put32(lh + 10, dos_date);
    put32(lh + 14, crc); put32(lh + 18, (unsigned long)body_len); put32(lh + 22, len); put16(lh + 26, (unsigned long)nname); put16(lh + 28, 0);
    int ok = fwrite(lh, 1, 30, z->fp) == 30 && fwrite(name, 1, nname, z->fp) == nname && (body_len == 0 || fwrite(body, 1, body_len, z->fp) == body_len);
    free(gz);
    if (!ok || pos < 0) return ZAMD_ZIP_ERRNO;
    const size_t need = 46 + nname + ncomm;
    if (z->cdir_len + need > z->cdir_cap) {
        const size_t ncap = (z->cdir_len + need) * 2 + 4096;
        uint8_t *np = (uint8_t *)realloc(z->cdir, ncap);
        if (!np) return ZAMD_ZIP_INTERNALERROR;
        z->cdir = np; z->cdir_cap = ncap;
    }
    uint8_t *c = z->cdir + z->cdir_len;
    put32(c, 0x02014b50ul); put16(c + 4, 0); put16(c + 6, 20); put16(c + 8, flag); put16(c + 10, (unsigned long)method); put32(c + 12, dos_date);
    put32(c + 16, crc); put32(c + 20, (unsigned long)body_len); put32(c + 24, len); put16(c + 28, (unsigned long)nname); put16(c + 30, 0);
    put16(c + 32, (unsigned long)ncomm); put16(c + 34, 0); put16(c + 36, method && data_type == Z_ASCII ? Z_ASCII : 0); put32(c + 38, 0);
    put32(c + 42, (unsigned long)pos);
    memcpy(c + 46, name, nname);
    if (ncomm) memcpy(c + 46 + nname, comment, ncomm);
    z->cdir_len += need; z->entries++;
    return ZAMD_ZIP_OK;
}

EXPORT int zamd_zip_close(zamd_zip *z, const char *global_comment)
{
    if (!z) return ZAMD_ZIP_PARAMERROR;
    int err = ZAMD_ZIP_OK;
    if (z->fp) {
        const long cpos = ftell(z->fp);
        const size_t ncomm = global_comment ? strlen(global_comment) : 0;
        uint8_t e[22];
        put32(e, 0x06054b50ul); put16(e + 4, 0); put16(e + 6, 0); put16(e + 8, z->entries); put16(e + 10, z->entries);
        put32(e + 12, (unsigned long)z->cdir_len); put32(e + 16, (unsigned long)cpos); put16(e + 20, (unsigned long)ncomm);
        if (cpos < 0 || (z->cdir_len && fwrite(z->cdir, 1, z->cdir_len, z->fp) != z->cdir_len) || fwrite(e, 1, 22, z->fp) != 22 ||
            (ncomm && fwrite(global_comment, 1, ncomm, z->fp) != ncomm))
            err = ZAMD_ZIP_ERRNO;
        if (fclose(z->fp)) err = ZAMD_ZIP_ERRNO;
    }
    free(z->cdir); free(z);
    return err;
}

/* ------------------------------------------------------------------ reading */
struct zamd_unzip { FILE *fp; zamd_zip_entry *ent; int n; long fsize; };

EXPORT zamd_unzip *zamd_unzip_open(const char *path)
{
    if (!path) return NULL;
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    zamd_unzip *u = (zamd_unzip *)calloc(1, sizeof *u);
    uint8_t *tail = NULL, *cd = NULL;
    if (!u || fseek(fp, 0, SEEK_END)) goto bad;
    {
        const long fsize = ftell(fp);
        const long span = fsize < 0xFFFF + 22 ? fsize : 0xFFFF + 22; /* the end record sits in the last 64 KiB + 22 bytes (unzip.c:320-383) */
        if (fsize < 22) goto bad;
        tail = (uint8_t *)malloc((size_t)span);
        if (!tail || fseek(fp, fsize - span, SEEK_SET) || fread(tail, 1, (size_t)span, fp) != (size_t)span) goto bad;
        long at = -1;
        for (long i = span - 22; i >= 0; i--) if (tail[i] == 0x50 && tail[i + 1] == 0x4b && tail[i + 2] == 5 && tail[i + 3] == 6) { at = i; break; }
        if (at < 0) goto bad;
        const uint8_t *e = tail + at;
        if (get16(e + 4) != 0 || get16(e + 6) != 0 || get16(e + 8) != get16(e + 10)) goto bad; /* one disk (unzip.c:456-460) */
        const unsigned long n = get16(e + 10), csize = get32(e + 12), coff = get32(e + 16);
        if ((long)(coff + csize) > fsize - span + at) goto bad;
        cd = (uint8_t *)malloc(csize + 1);
        u->ent = (zamd_zip_entry *)calloc(n ? n : 1, sizeof *u->ent);
        if (!cd || !u->ent || fseek(fp, (long)coff, SEEK_SET) || fread(cd, 1, csize, fp) != csize) goto bad;
        size_t o = 0;
        for (unsigned long i = 0; i < n; i++) {
            if (o + 46 > csize || get32(cd + o) != 0x02014b50ul) goto bad;
            zamd_zip_entry *t = &u->ent[i];
            const unsigned long nn = get16(cd + o + 28), nx = get16(cd + o + 30), nc = get16(cd + o + 32);
            if (o + 46 + nn + nx + nc > csize) goto bad;
            t->flag = (int)get16(cd + o + 8); t->method = (int)get16(cd + o + 10); t->dos_date = get32(cd + o + 12); t->crc32 = get32(cd + o + 16);
            t->compressed_size = get32(cd + o + 20); t->uncompressed_size = get32(cd + o + 24); t->internal_fa = (int)get16(cd + o + 36);
            t->local_header_offset = get32(cd + o + 42);
            const size_t keep = nn < sizeof t->name - 1 ? nn : sizeof t->name - 1;
            memcpy(t->name, cd + o + 46, keep); t->name[keep] = 0;
            o += 46 + nn + nx + nc;
        }
        u->n = (int)n;
        u->fsize = fsize;
    }
    free(tail); free(cd);
    u->fp = fp;
    return u;
bad:
    free(tail); free(cd);
    if (u) { free(u->ent); free(u); }
    fclose(fp);
    return NULL;
}
EXPORT int zamd_unzip_count(const zamd_unzip *u) { return u ? u->n : ZAMD_ZIP_PARAMERROR; }
EXPORT int zamd_unzip_stat(const zamd_unzip *u, int i, zamd_zip_entry *out)
{
    if (!u || !out || i < 0 || i >= u->n) return ZAMD_ZIP_PARAMERROR;
    *out = u->ent[i];
    return ZAMD_ZIP_OK;
}
EXPORT int zamd_unzip_locate(const zamd_unzip *u, const char *name)
{
    if (!u || !name) return ZAMD_ZIP_PARAMERROR;
    for (int i = 0; i < u->n; i++) if (strcmp(u->ent[i].name, name) == 0) return i;
    return ZAMD_ZIP_PARAMERROR;
}
EXPORT long zamd_unzip_read(zamd_unzip *u, int i, void *out, unsigned long cap)
{
    if (!u || i < 0 || i >= u->n || (!out && cap)) return ZAMD_ZIP_PARAMERROR;
    const zamd_zip_entry *t = &u->ent[i];
    if (t->uncompressed_size > cap) return ZAMD_ZIP_PARAMERROR;
    if (t->method != 0 && t->method != Z_DEFLATED) return ZAMD_ZIP_BADZIPFILE;
    /* (a directory that claims more than the file holds is damaged: nothing is allocated on its word) */
    if ((unsigned long)u->fsize < 30ul || t->local_header_offset > (unsigned long)u->fsize - 30ul || t->compressed_size > (unsigned long)u->fsize - t->local_header_offset - 30ul) return ZAMD_ZIP_BADZIPFILE;
    uint8_t lh[30];
    if (fseek(u->fp, (long)t->local_header_offset, SEEK_SET) || fread(lh, 1, 30, u->fp) != 30) return ZAMD_ZIP_ERRNO;
    if (get32(lh) != 0x04034b50ul || (int)get16(lh + 8) != t->method) return ZAMD_ZIP_BADZIPFILE; /* unzip.c:983-1050 */
    if (fseek(u->fp, (long)(get16(lh + 26) + get16(lh + 28)), SEEK_CUR)) return ZAMD_ZIP_ERRNO;
    if (t->method == 0) {
        if (t->compressed_size != t->uncompressed_size) return ZAMD_ZIP_BADZIPFILE;
        if (fread(out, 1, t->uncompressed_size, u->fp) != t->uncompressed_size) return ZAMD_ZIP_ERRNO;
        unsigned long crc = 0;
        for (unsigned long o = 0; o < t->uncompressed_size; o += 0x40000000ul)
            crc = crc32(crc, (const Bytef *)out + o, (uInt)(t->uncompressed_size - o < 0x40000000ul ? t->uncompressed_size - o : 0x40000000ul));
        return crc == t->crc32 ? (long)t->uncompressed_size : ZAMD_ZIP_CRCERROR;
    }
    /* the member as a gzip member: header, the raw deflate data, CRC-32 and length -- inflate checks both on the device */
    const size_t glen = 10 + (size_t)t->compressed_size + 8;
    uint8_t *gz = (uint8_t *)malloc(glen);
    if (!gz) return ZAMD_ZIP_INTERNALERROR;
    static const uint8_t gh[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    memcpy(gz, gh, 10);
    if (fread(gz + 10, 1, t->compressed_size, u->fp) != t->compressed_size) { free(gz); return ZAMD_ZIP_ERRNO; }
    put32(gz + 10 + t->compressed_size, t->crc32); put32(gz + 14 + t->compressed_size, t->uncompressed_size);
    z_stream s; memset(&s, 0, sizeof s);
    if (inflateInit2(&s, 31) != Z_OK) { free(gz); return ZAMD_ZIP_INTERNALERROR; }
    size_t ipos = 0; unsigned long opos = 0; int rc = Z_OK;
    while (rc == Z_OK) { /* 32-bit windows over buffers of any size */
        const size_t in_now = glen - ipos > 0xC0000000ul ? 0xC0000000ul : glen - ipos;
        const unsigned long out_now = cap - opos > 0xFFFFFFFFul ? 0xFFFFFFFFul : cap - opos;
        s.next_in = gz + ipos; s.avail_in = (uInt)in_now; s.next_out = (Bytef *)out + opos; s.avail_out = (uInt)out_now;
        rc = inflate(&s, ipos + in_now == glen ? Z_FINISH : Z_NO_FLUSH);
        const int moved = s.avail_in != in_now || s.avail_out != out_now;
        ipos += in_now - s.avail_in; opos += out_now - s.avail_out;
        if (rc == Z_BUF_ERROR && ipos < glen && moved) rc = Z_OK; /* (no progress: more data than the directory announced -- out of the loop, a bad file) */
        if (rc == Z_OK && !moved) rc = Z_BUF_ERROR;
    }
    inflateEnd(&s);
    free(gz);
    if (rc == Z_STREAM_END && opos == t->uncompressed_size) return (long)opos;
    return rc == Z_DATA_ERROR ? ZAMD_ZIP_CRCERROR : ZAMD_ZIP_BADZIPFILE;
}
EXPORT int zamd_unzip_close(zamd_unzip *u)
{
    if (!u) return ZAMD_ZIP_PARAMERROR;
    fclose(u->fp); free(u->ent); free(u);
    return ZAMD_ZIP_OK;
}
