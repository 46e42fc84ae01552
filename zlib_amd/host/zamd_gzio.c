/* zamd_gzio.c -- the gz* file functions of the zlib 1.2.3 API (/root/reference/qcsrc/gzio.c) as a thin stdio driver over this
 * library's own deflate() / inflate().  Host side only: the codec work happens on the GPU behind those two calls.
 *
 * What gzio.c does and this file restates in its own form:
 *   gzopen / gzdopen   mode string "rb", "wb6", "wb9f" ... (gzio.c:93-203): level digit, strategy letter f / h / R
 *   writing            a 10-byte gzip header (magic, method 8, no flags, no time, no extra flags, OS code), the raw deflate
 *                      data, CRC-32 and length of the uncompressed data, both little-endian (gzio.c:175-184, 559-596, 980-1005)
 *   reading            the header of every member is checked and skipped (gzio.c:281-347); a file that does not start with
 *                      the gzip magic is handed through unchanged ("transparent", gzio.c:300-307); members may follow one
 *                      another (gzio.c:459-476); garbage behind the last member is ignored (gzio.c:293-298)
 *   gzseek             writing: forward only, by writing zeros; reading: backwards = rewind and read forward (gzio.c:775-874)
 */
#include "../../include/zamd_zlib.h"
#include <errno.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define EXPORT __attribute__((visibility("default")))
#define GZ_BUF (1u << 20) /* file-side buffer: the engine works on 64 KiB chunks, a megabyte keeps sixteen of them in one call */
#define GZ_OS_CODE 3

typedef struct {
    z_stream strm;
    int err;          /* last error (Z_OK, Z_STREAM_END at the end of the data when reading, Z_ERRNO, Z_DATA_ERROR ...) */
    int eof;          /* reading: the file has no more bytes */
    FILE *fp;
    uint8_t *buf;     /* reading: compressed bytes from the file; writing: compressed bytes for the file */
    uLong crc;        /* CRC-32 of the uncompressed data of the current member */
    char *path, *msg;
    int transparent;  /* reading: the file is not gzip, bytes are copied through */
    char mode;        /* 'r' or 'w' */
    z_off_t start;    /* reading: file offset of the first member's deflate data (gzrewind) */
    z_off_t in, out;  /* uncompressed bytes written (w) / compressed bytes consumed and uncompressed bytes delivered (r) */
    int back;         /* one byte pushed back by gzungetc, or -1 */
    int last;         /* reading: the pushed-back byte is the last one of the data */
    int level, strategy;
    int started;      /* reading: the header of the first member has been looked at */
    uint8_t *carry;   /* reading: input behind the end of a member that inflate() had taken in already, in front of the file's next bytes */
} gz_file;

void zamd_inflate_rest(z_streamp strm, const unsigned char **p, size_t *n); /* zamd_zlib.c */
size_t zamd_inflate_pending(z_streamp strm);                                       /* zamd_zlib.c */

static int gz_destroy(gz_file *s)
{
    int err = Z_OK;
    if (!s) return Z_STREAM_ERROR;
    free(s->msg);
    if (s->strm.state != Z_NULL) err = s->mode == 'w' ? deflateEnd(&s->strm) : inflateEnd(&s->strm);
    if (s->fp != NULL && fclose(s->fp)) err = Z_ERRNO;
    if (s->err < 0) err = s->err;
    free(s->buf); free(s->carry); free(s->path); free(s);
    return err;
}

static gzFile gz_open_any(const char *path, const char *mode, int fd)
{
    if (!path || !mode) return Z_NULL;
    gz_file *s = (gz_file *)calloc(1, sizeof *s);
    if (!s) return Z_NULL;
    s->back = -1; s->level = Z_DEFAULT_COMPRESSION; s->strategy = Z_DEFAULT_STRATEGY;
    s->path = strdup(path);
    char fmode[8]; size_t m = 0;
    for (const char *p = mode; *p; p++) {
        if (*p == 'r') s->mode = 'r';
        if (*p == 'w' || *p == 'a') s->mode = 'w';
        if (*p >= '0' && *p <= '9') s->level = *p - '0';
        else if (*p == 'f') s->strategy = Z_FILTERED;
        else if (*p == 'h') s->strategy = Z_HUFFMAN_ONLY;
        else if (*p == 'R') s->strategy = Z_RLE;
        else if (m < sizeof fmode - 2 && *p != '+') fmode[m++] = *p; /* what fopen understands */
    }
    fmode[m] = 0;
    if (!s->path || s->mode == 0) { gz_destroy(s); return Z_NULL; }
    int rc;
    if (s->mode == 'w') rc = deflateInit2(&s->strm, s->level, Z_DEFLATED, -15, 8, s->strategy);
    else rc = inflateInit2(&s->strm, -15);
    s->buf = (uint8_t *)malloc(GZ_BUF);
    if (rc != Z_OK || !s->buf) { gz_destroy(s); return Z_NULL; }
    if (s->mode == 'w') { s->strm.next_out = s->buf; s->strm.avail_out = GZ_BUF; }
    errno = 0;
    s->fp = fd < 0 ? fopen(path, fmode) : fdopen(fd, fmode);
    if (!s->fp) { gz_destroy(s); return Z_NULL; }
    if (s->mode == 'w') {
        const uint8_t h[10] = {0x1f, 0x8b, Z_DEFLATED, 0, 0, 0, 0, 0, 0, GZ_OS_CODE};
        if (fwrite(h, 1, 10, s->fp) != 10) { gz_destroy(s); return Z_NULL; }
        s->start = 10;
    }
    return (gzFile)s;
}
EXPORT gzFile gzopen(const char *path, const char *mode) { return gz_open_any(path, mode, -1); }
EXPORT gzFile gzdopen(int fd, const char *mode)
{
    char name[46];
    if (fd < 0) return Z_NULL;
    snprintf(name, sizeof name, "<fd:%d>", fd);
    return gz_open_any(name, mode, fd);
}

/* ------------------------------------------------------------------ reading */
static void gz_fill(gz_file *s) /* more compressed bytes behind what next_in still holds */
{
    if (s->eof || s->strm.avail_in > GZ_BUF / 2) return; /* (a carried-over tail can be longer than the buffer: it is used up first) */
    if (s->strm.avail_in && s->strm.next_in != s->buf) memmove(s->buf, s->strm.next_in, s->strm.avail_in);
    s->strm.next_in = s->buf;
    errno = 0;
    const size_t got = fread(s->buf + s->strm.avail_in, 1, GZ_BUF - s->strm.avail_in, s->fp);
    if (got == 0) { s->eof = 1; if (ferror(s->fp)) s->err = Z_ERRNO; }
    s->strm.avail_in += (uInt)got;
}
static int gz_byte(gz_file *s) /* next compressed byte, or -1 */
{
    if (s->strm.avail_in == 0) { gz_fill(s); if (s->strm.avail_in == 0) return -1; }
    s->strm.avail_in--; s->in++;
    return *s->strm.next_in++;
}
/* the header of a member (gzio.c:281-347): Z_OK and positioned at the deflate data, or transparent, or an error / the end */
static void gz_header_in(gz_file *s)
{
    if (s->strm.avail_in < 2) gz_fill(s);
    if (s->strm.avail_in < 2 || s->strm.next_in[0] != 0x1f || s->strm.next_in[1] != 0x8b) {
        if (!s->started) s->transparent = s->strm.avail_in != 0; /* not a gzip file: copy through */
        else s->err = Z_STREAM_END;                              /* behind the last member: the end of the data, garbage ignored */
        if (s->strm.avail_in == 0 && !s->started) s->err = Z_STREAM_END;
        s->started = 1;
        return;
    }
    s->started = 1;
    s->strm.next_in += 2; s->strm.avail_in -= 2; s->in += 2;
    const int method = gz_byte(s), flags = gz_byte(s);
    if (method != Z_DEFLATED || flags < 0 || (flags & 0xe0)) { s->err = Z_DATA_ERROR; return; }
    for (int i = 0; i < 6; i++) (void)gz_byte(s); /* time, extra flags, OS */
    if (flags & 4) { int len = gz_byte(s); len += gz_byte(s) << 8; while (len-- > 0 && gz_byte(s) != -1) { } }
    if (flags & 8) { int c; while ((c = gz_byte(s)) != 0 && c != -1) { } }
    if (flags & 16) { int c; while ((c = gz_byte(s)) != 0 && c != -1) { } }
    if (flags & 2) { (void)gz_byte(s); (void)gz_byte(s); }
    s->err = s->eof && s->strm.avail_in == 0 ? Z_DATA_ERROR : Z_OK;
}
static uLong gz_long(gz_file *s)
{
    uLong x = (uLong)gz_byte(s);
    x += (uLong)gz_byte(s) << 8; x += (uLong)gz_byte(s) << 16;
    const int c = gz_byte(s);
    if (c == -1) s->err = Z_DATA_ERROR;
    return x + ((uLong)c << 24);
}

EXPORT int gzread(gzFile file, voidp buf, unsigned len)
{
    gz_file *s = (gz_file *)file;
    if (!s || s->mode != 'r') return Z_STREAM_ERROR;
    if (s->err == Z_DATA_ERROR || s->err == Z_ERRNO) return -1;
    if (s->err == Z_STREAM_END || len == 0) return 0;
    unsigned from_back = 0;
    s->strm.next_out = (Bytef *)buf; s->strm.avail_out = len;
    if (s->back != -1) { /* the byte gzungetc pushed back comes first (gzio.c:416-427) */
        *s->strm.next_out++ = (Bytef)s->back; s->strm.avail_out--; s->back = -1; s->out++; from_back = 1;
        if (s->last) { s->err = Z_STREAM_END; return 1; }
    }
    Bytef *start = s->strm.next_out; /* bytes from here on have not been through the CRC */
    if (!s->started) gz_header_in(s);
    if (s->err == Z_STREAM_END) return (int)from_back;
    if (s->err != Z_OK) return from_back ? (int)from_back : -1;
    while (s->strm.avail_out != 0) {
        if (s->transparent) { /* copy: first what the look-ahead holds, then straight from the file (gzio.c:431-457) */
            const uInt n = s->strm.avail_in < s->strm.avail_out ? s->strm.avail_in : s->strm.avail_out;
            if (n) { memcpy(s->strm.next_out, s->strm.next_in, n); s->strm.next_out += n; s->strm.next_in += n; s->strm.avail_out -= n; s->strm.avail_in -= n; }
            if (s->strm.avail_out) { const size_t got = fread(s->strm.next_out, 1, s->strm.avail_out, s->fp); s->strm.next_out += got; s->strm.avail_out -= (uInt)got; if (got == 0) s->eof = 1; }
            const unsigned got = (unsigned)(s->strm.next_out - start);
            s->in += got; s->out += got;
            if (got == 0 && s->eof) s->err = Z_STREAM_END;
            return (int)(from_back + got);
        }
        if (s->strm.avail_in == 0 && !s->eof && zamd_inflate_pending(&s->strm) == 0) gz_fill(s); /* (what is decoded already goes out first) */
        const uInt in0 = s->strm.avail_in;
        const int rc = inflate(&s->strm, s->eof ? Z_FINISH : Z_NO_FLUSH);
        s->in += in0 - s->strm.avail_in;
        if (rc == Z_OK) continue;
        if (rc == Z_BUF_ERROR) {
            if (s->strm.avail_out == 0 || !s->eof) continue; /* the caller's buffer is full (the loop ends), or more of the file is needed */
            s->err = Z_BUF_ERROR;                              /* the file ends inside the deflate data */
            break;
        }
        if (rc != Z_STREAM_END) { s->err = rc; break; }
        /* the member is complete: CRC and length, then maybe another member (gzio.c:459-476) */
        {   /* what inflate() took in beyond the end of the deflate data in earlier calls comes first */
            const unsigned char *rest; size_t nrest;
            zamd_inflate_rest(&s->strm, &rest, &nrest);
            if (nrest) {
                uint8_t *c = (uint8_t *)malloc(nrest + s->strm.avail_in + 1);
                if (!c) { s->err = Z_MEM_ERROR; break; }
                memcpy(c, rest, nrest);
                if (s->strm.avail_in) memcpy(c + nrest, s->strm.next_in, s->strm.avail_in);
                free(s->carry); s->carry = c;
                s->strm.next_in = c; s->strm.avail_in += (uInt)nrest; s->in -= (z_off_t)nrest;
            }
        }
        s->crc = crc32(s->crc, start, (uInt)(s->strm.next_out - start));
        s->out += (z_off_t)(s->strm.next_out - start);
        start = s->strm.next_out;
        const uLong want = gz_long(s);
        (void)gz_long(s); /* the length is not checked: in some files it is the length of all members modulo 2^32 (gzio.c:466-470) */
        if (s->err == Z_DATA_ERROR || want != s->crc) { s->err = Z_DATA_ERROR; break; }
        gz_header_in(s);
        if (s->err != Z_OK) break; /* Z_STREAM_END: that was the last member */
        inflateReset(&s->strm);
        s->crc = crc32(0L, Z_NULL, 0);
    }
    s->crc = crc32(s->crc, start, (uInt)(s->strm.next_out - start));
    s->out += (z_off_t)(s->strm.next_out - start);
    const unsigned got = len - s->strm.avail_out;
    if (got == 0 && s->err != Z_OK && s->err != Z_STREAM_END) return -1;
    return (int)got;
}
EXPORT int gzgetc(gzFile file)
{
    unsigned char c;
    return gzread(file, &c, 1) == 1 ? c : -1;
}
EXPORT int gzungetc(int c, gzFile file)
{
    gz_file *s = (gz_file *)file;
    if (!s || s->mode != 'r' || c == -1 || s->back != -1) return -1;
    s->back = c; s->out--;
    s->last = s->err == Z_STREAM_END;
    if (s->last) s->err = Z_OK;
    return c;
}
EXPORT char *gzgets(gzFile file, char *buf, int len)
{
    char *b = buf;
    if (buf == Z_NULL || len <= 0) return Z_NULL;
    while (--len > 0 && gzread(file, buf, 1) == 1 && *buf++ != '\n') { }
    *buf = '\0';
    return b == buf && len > 0 ? Z_NULL : b;
}

/* ------------------------------------------------------------------ writing */
static int gz_drain(gz_file *s, int flush) /* deflate with `flush` until it has nothing more to hand out; the output goes to the file */
{
    for (;;) {
        if (s->strm.avail_out == 0 || flush != Z_NO_FLUSH) {
            const size_t n = GZ_BUF - s->strm.avail_out;
            if (n && fwrite(s->buf, 1, n, s->fp) != n) { s->err = Z_ERRNO; return Z_ERRNO; }
            s->strm.next_out = s->buf; s->strm.avail_out = GZ_BUF;
        }
        const uInt out0 = s->strm.avail_out;
        if (s->strm.avail_in == 0 && flush == Z_NO_FLUSH) break;
        s->err = deflate(&s->strm, flush);
        if (s->err == Z_BUF_ERROR) s->err = Z_OK; /* nothing to do is not an error here (gzio.c:575-578) */
        const int done = s->strm.avail_out != 0 || s->err == Z_STREAM_END;
        if (flush != Z_NO_FLUSH && (out0 != s->strm.avail_out || !done)) { /* hand over what came out, ask again */
            const size_t n = GZ_BUF - s->strm.avail_out;
            if (n && fwrite(s->buf, 1, n, s->fp) != n) { s->err = Z_ERRNO; return Z_ERRNO; }
            s->strm.next_out = s->buf; s->strm.avail_out = GZ_BUF;
        }
        if (s->err != Z_OK && s->err != Z_STREAM_END) break;
        if (done && (flush != Z_NO_FLUSH || s->strm.avail_in == 0)) break;
    }
    return s->err == Z_STREAM_END ? Z_OK : s->err;
}
EXPORT int gzwrite(gzFile file, const void *buf, unsigned len)
{
    gz_file *s = (gz_file *)file;
    if (!s || s->mode != 'w') return Z_STREAM_ERROR;
    s->strm.next_in = (Bytef *)buf; s->strm.avail_in = len;
    while (s->strm.avail_in != 0) {
        const uInt in0 = s->strm.avail_in;
        if (gz_drain(s, Z_NO_FLUSH) != Z_OK) break;
        s->in += in0 - s->strm.avail_in;
        if (in0 == s->strm.avail_in && s->strm.avail_out != 0) break; /* (cannot happen: deflate takes all input) */
    }
    s->crc = crc32(s->crc, (const Bytef *)buf, len);
    return (int)(len - s->strm.avail_in);
}
EXPORT int gzprintf(gzFile file, const char *format, ...)
{
    char buf[4096];
    va_list va;
    va_start(va, format);
    const int len = vsnprintf(buf, sizeof buf, format, va);
    va_end(va);
    if (len <= 0 || len >= (int)sizeof buf) return 0; /* gzio.c:631-646: nothing is written when it does not fit */
    return gzwrite(file, buf, (unsigned)len);
}
EXPORT int gzputs(gzFile file, const char *str) { return gzwrite(file, str, (unsigned)strlen(str)); }
EXPORT int gzputc(gzFile file, int c)
{
    unsigned char cc = (unsigned char)c;
    return gzwrite(file, &cc, 1) == 1 ? (int)cc : -1;
}
EXPORT int gzflush(gzFile file, int flush)
{
    gz_file *s = (gz_file *)file;
    if (!s || s->mode != 'w') return Z_STREAM_ERROR;
    s->strm.avail_in = 0;
    const int err = gz_drain(s, flush);
    if (err) return err;
    fflush(s->fp);
    return s->err == Z_STREAM_END ? Z_OK : s->err;
}
EXPORT int gzsetparams(gzFile file, int level, int strategy)
{
    gz_file *s = (gz_file *)file;
    if (!s || s->mode != 'w') return Z_STREAM_ERROR;
    if (s->strm.avail_out == 0) { /* room for what deflateParams may hand out (gzio.c:214-222) */
        if (fwrite(s->buf, 1, GZ_BUF, s->fp) != GZ_BUF) s->err = Z_ERRNO;
        s->strm.next_out = s->buf; s->strm.avail_out = GZ_BUF;
    }
    return deflateParams(&s->strm, level, strategy);
}

/* ------------------------------------------------------------------ positions */
EXPORT int gzrewind(gzFile file)
{
    gz_file *s = (gz_file *)file;
    if (!s || s->mode != 'r') return -1;
    s->err = Z_OK; s->eof = 0; s->back = -1; s->strm.avail_in = 0; s->strm.next_in = s->buf; s->crc = crc32(0L, Z_NULL, 0);
    s->in = s->out = 0; s->started = 0; s->transparent = 0;
    inflateReset(&s->strm);
    return fseek(s->fp, 0L, SEEK_SET);
}
EXPORT z_off_t gzseek(gzFile file, z_off_t offset, int whence)
{
    gz_file *s = (gz_file *)file;
    if (!s || whence == SEEK_END || s->err == Z_ERRNO || s->err == Z_DATA_ERROR) return -1L;
    if (s->mode == 'w') {
        if (whence == SEEK_SET) offset -= s->in;
        if (offset < 0) return -1L;
        uint8_t *zeros = (uint8_t *)calloc(1, 65536); /* the gap is written as zeros (gzio.c:795-811) */
        if (!zeros) return -1L;
        while (offset > 0) {
            const unsigned n = offset < 65536 ? (unsigned)offset : 65536u;
            if (gzwrite(file, zeros, n) != (int)n) { free(zeros); return -1L; }
            offset -= n;
        }
        free(zeros);
        return s->in;
    }
    if (whence == SEEK_CUR) offset += s->out;
    if (offset < 0) return -1L;
    if (offset < s->out) { if (gzrewind(file) < 0) return -1L; } /* backwards: from the start again */
    else if (s->err == Z_STREAM_END && s->back == -1) { /* already at the end of the data: stay there */ }
    offset -= s->out;
    if (offset && s->err == Z_STREAM_END) return s->out;
    uint8_t *skip = offset ? (uint8_t *)malloc(65536) : NULL;
    if (offset && !skip) return -1L;
    while (offset > 0) {
        const unsigned n = offset < 65536 ? (unsigned)offset : 65536u;
        const int got = gzread(file, skip, n);
        if (got <= 0) { free(skip); return got < 0 ? -1L : s->out; }
        offset -= got;
    }
    free(skip);
    return s->out;
}
EXPORT z_off_t gztell(gzFile file) { return gzseek(file, 0L, SEEK_CUR); }
EXPORT int gzeof(gzFile file)
{
    const gz_file *s = (const gz_file *)file;
    if (!s || s->mode != 'r') return 0;
    return (s->eof && s->strm.avail_in == 0 && s->back == -1) || s->err == Z_STREAM_END; /* gzio.c:911-921 */
}
EXPORT int gzdirect(gzFile file)
{
    const gz_file *s = (const gz_file *)file;
    return s && s->mode == 'r' ? s->transparent : 0;
}
EXPORT int gzclose(gzFile file)
{
    gz_file *s = (gz_file *)file;
    if (!s) return Z_STREAM_ERROR;
    if (s->mode == 'w') {
        s->strm.avail_in = 0;
        if (gz_drain(s, Z_FINISH) == Z_OK) { /* trailer: CRC-32 and length of the uncompressed data (gzio.c:980-1005) */
            const uint32_t c = (uint32_t)s->crc, l = (uint32_t)s->in;
            const uint8_t t[8] = {(uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24), (uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
            if (fwrite(t, 1, 8, s->fp) != 8) s->err = Z_ERRNO;
        }
    }
    return gz_destroy(s);
}
EXPORT const char *gzerror(gzFile file, int *errnum)
{
    gz_file *s = (gz_file *)file;
    if (!s) { *errnum = Z_STREAM_ERROR; return zError(Z_STREAM_ERROR); }
    *errnum = s->err;
    if (s->err == Z_OK) return "";
    const char *m = s->err == Z_ERRNO ? strerror(errno) : s->strm.msg;
    if (m == Z_NULL || *m == '\0') m = zError(s->err);
    free(s->msg);
    s->msg = (char *)malloc(strlen(s->path) + strlen(m) + 3);
    if (!s->msg) return zError(Z_MEM_ERROR);
    sprintf(s->msg, "%s: %s", s->path, m);
    return s->msg;
}
EXPORT void gzclearerr(gzFile file)
{
    gz_file *s = (gz_file *)file;
    if (!s) return;
    if (s->err != Z_STREAM_END) s->err = Z_OK;
    s->eof = 0;
    clearerr(s->fp);
}
