/* zamd_infback.c -- inflateBackInit_ / inflateBack / inflateBackEnd of the zlib 1.2.3 API (/root/reference/qcsrc/infback.c:28-72,
 * 241-612, 614-623): raw deflate data pulled through a caller's in() and pushed, one window at a time, through a caller's out().
 * Host side only: the decoding is this library's inflate() (raw, -15), whose codec work happens on the GPU.
 *
 * Contract restated (h/zlib.h:880-947): in() hands over the next piece of input (0 bytes: there is no more -> Z_BUF_ERROR with
 * strm->next_in == Z_NULL); out() receives the caller's window buffer and a byte count whenever the window is full and once more
 * at the end (non-zero return -> Z_BUF_ERROR with strm->next_in != Z_NULL); Z_STREAM_END leaves the unused input in
 * strm->next_in / avail_in; Z_DATA_ERROR sets strm->msg.
 */
#include "../../include/zamd_zlib.h"
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))
#define KIND_BACK 0x5a42

struct back_state { int kind; z_stream inner; unsigned char *window; unsigned wsize; unsigned char *carry; };

void zamd_inflate_rest(z_streamp strm, const unsigned char **p, size_t *n); /* zamd_zlib.c */
size_t zamd_inflate_pending(z_streamp strm);                                       /* zamd_zlib.c */

EXPORT int inflateBackInit_(z_streamp strm, int windowBits, unsigned char *window, const char *version, int stream_size)
{
    if (version == Z_NULL || version[0] != ZLIB_VERSION[0] || stream_size != (int)sizeof(z_stream)) return Z_VERSION_ERROR;
    if (strm == Z_NULL || window == Z_NULL || windowBits < 8 || windowBits > 15) return Z_STREAM_ERROR;
    strm->msg = Z_NULL;
    struct back_state *b = (struct back_state *)calloc(1, sizeof *b);
    if (!b) return Z_MEM_ERROR;
    b->kind = KIND_BACK; b->window = window; b->wsize = 1u << windowBits;
    b->inner.zalloc = strm->zalloc; b->inner.zfree = strm->zfree; b->inner.opaque = strm->opaque;
    const int rc = inflateInit2(&b->inner, -15);
    if (rc != Z_OK) { strm->msg = b->inner.msg; free(b); return rc; }
    strm->state = (struct internal_state *)b;
    return Z_OK;
}

EXPORT int inflateBack(z_streamp strm, in_func in, void *in_desc, out_func out, void *out_desc)
{
    if (strm == Z_NULL || strm->state == Z_NULL || ((struct back_state *)strm->state)->kind != KIND_BACK) return Z_STREAM_ERROR;
    struct back_state *b = (struct back_state *)strm->state;
    z_stream *z = &b->inner;
    strm->msg = Z_NULL;
    inflateReset(z); /* every call decodes one deflate stream from its start (infback.c:265-270) */
    z->next_in = strm->next_in; z->avail_in = strm->next_in != Z_NULL ? strm->avail_in : 0;
    z->next_out = b->window; z->avail_out = b->wsize;
    int no_more = 0;
    for (;;) {
        if (z->avail_in == 0 && !no_more && zamd_inflate_pending(z) == 0) { /* (what is decoded already goes through the window first) */
            unsigned char *next = Z_NULL;
            const unsigned have = in(in_desc, &next);
            if (have == 0) no_more = 1; else { z->next_in = next; z->avail_in = have; }
        }
        const int rc = inflate(z, no_more ? Z_FINISH : Z_NO_FLUSH);
        if (z->avail_out == 0 || (rc == Z_STREAM_END && z->avail_out != b->wsize)) {
            if (out(out_desc, b->window, b->wsize - z->avail_out)) { strm->next_in = z->next_in; strm->avail_in = z->avail_in; return Z_BUF_ERROR; }
            z->next_out = b->window; z->avail_out = b->wsize;
            if (rc != Z_STREAM_END) continue; /* more may be waiting inside */
        }
        if (rc == Z_STREAM_END) { /* the unused input: what the decoder had taken in beyond the end, then what the last piece still holds */
            const unsigned char *rest; size_t nrest;
            zamd_inflate_rest(z, &rest, &nrest);
            strm->next_in = z->next_in; strm->avail_in = z->avail_in;
            if (nrest) {
                free(b->carry);
                b->carry = (unsigned char *)malloc(nrest + z->avail_in + 1);
                if (!b->carry) return Z_MEM_ERROR;
                memcpy(b->carry, rest, nrest);
                if (z->avail_in) memcpy(b->carry + nrest, z->next_in, z->avail_in);
                strm->next_in = b->carry; strm->avail_in = (uInt)(nrest + z->avail_in);
            }
            return Z_STREAM_END;
        }
        if (rc == Z_DATA_ERROR || rc == Z_MEM_ERROR || rc == Z_STREAM_ERROR || rc == Z_NEED_DICT) {
            if (z->avail_out != b->wsize) (void)out(out_desc, b->window, b->wsize - z->avail_out); /* what was decoded before the error */
            strm->msg = z->msg; strm->next_in = z->next_in; strm->avail_in = z->avail_in;
            return rc == Z_NEED_DICT ? Z_DATA_ERROR : rc;
        }
        if (no_more && rc == Z_BUF_ERROR) { /* the input ends inside the deflate data (infback.c:PULL -> Z_BUF_ERROR) */
            if (z->avail_out != b->wsize && out(out_desc, b->window, b->wsize - z->avail_out)) { strm->next_in = z->next_in; strm->avail_in = 0; return Z_BUF_ERROR; }
            strm->next_in = Z_NULL; strm->avail_in = 0;
            return Z_BUF_ERROR;
        }
    }
}

EXPORT int inflateBackEnd(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || ((struct back_state *)strm->state)->kind != KIND_BACK) return Z_STREAM_ERROR;
    struct back_state *b = (struct back_state *)strm->state;
    inflateEnd(&b->inner);
    free(b->carry);
    free(b);
    strm->state = Z_NULL;
    return Z_OK;
}
