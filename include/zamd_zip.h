/* zamd_zip.h -- PKZIP bulk entry points of libzamd_z.so.
 *
 * The reference's minizip (qcsrc/zip.c, unzip.c) feeds deflate() 16 KiB slices of a file (zip.c:45, 969-1006) and reads entries
 * back through a 16 KiB buffer (unzip.c:1230-1389).  Compiled from the mount it works against this library as it is
 * (oracle/_ref/minizip_zamd, tests/test_gpu_zip.py); these calls are the bulk form of the same thing: a whole member goes to the
 * chunked engine in one piece, the bytes written to the archive are the ones
 *     zipOpen + zipOpenNewFileInZip + zipWriteInFileInZip* + zipCloseFileInZip + zipClose      (zip.c:502-690, 693-900, 969-1128, 1136-1213)
 * write for the same member data, level, DOS date and comments (local header with the sizes patched in afterwards, central directory,
 * end record; no zip64, no encryption, no spanning -- as minizip 1.01e), and an entry is decoded by one raw inflate of the whole
 * member with the CRC-32 checked on the device.
 */
#ifndef ZAMD_ZIP_H
#define ZAMD_ZIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define ZAMD_ZIP_OK 0
#define ZAMD_ZIP_ERRNO (-1)         /* file I/O failed */
#define ZAMD_ZIP_PARAMERROR (-102)  /* ZIP_PARAMERROR */
#define ZAMD_ZIP_BADZIPFILE (-103)  /* ZIP_BADZIPFILE / UNZ_BADZIPFILE */
#define ZAMD_ZIP_INTERNALERROR (-104)
#define ZAMD_ZIP_CRCERROR (-105)    /* UNZ_CRCERROR */

typedef struct zamd_zip zamd_zip;
typedef struct zamd_unzip zamd_unzip;
typedef struct {
    char name[512];
    unsigned long crc32, compressed_size, uncompressed_size, dos_date, local_header_offset;
    int method, flag, internal_fa;
} zamd_zip_entry;

/* writing: a new archive at `path` */
zamd_zip *zamd_zip_open(const char *path);
/* one member: `len` bytes at `data` (< 4 GiB), level 0 (stored, method 0) .. 9 or -1; dos_date as zip_fileinfo.dosDate (zip.h:93-101) */
int zamd_zip_add(zamd_zip *z, const char *name, const void *data, unsigned long len, int level, unsigned long dos_date, const char *comment);
int zamd_zip_close(zamd_zip *z, const char *global_comment);

/* reading */
zamd_unzip *zamd_unzip_open(const char *path);
int zamd_unzip_count(const zamd_unzip *u);
int zamd_unzip_stat(const zamd_unzip *u, int index, zamd_zip_entry *out);
int zamd_unzip_locate(const zamd_unzip *u, const char *name); /* index, or ZAMD_ZIP_PARAMERROR */
/* the whole member: returns the number of bytes written to out, or a negative ZAMD_ZIP_* code (cap too small: ZAMD_ZIP_PARAMERROR) */
long zamd_unzip_read(zamd_unzip *u, int index, void *out, unsigned long cap);
int zamd_unzip_close(zamd_unzip *u);

#ifdef __cplusplus
}
#endif
#endif
