/* corpus_host.c -- host (gcc) build of the seeded corpus generator zlib_amd/csrc/corpus.h.
 * TEST/BENCH INFRASTRUCTURE: used by oracle/gen_golden.py, tests/ and bench.py's cpu_baseline leg. */
#include "../zlib_amd/csrc/corpus.h"
#include <stddef.h>

/* fill out[0 .. nchunks*65536) with chunks first_chunk .. first_chunk+nchunks-1; out must be 8-byte aligned */
void zc_host_fill(uint32_t kind, uint64_t seed, uint64_t first_chunk, uint64_t nchunks, uint8_t *out)
{
    for (uint64_t k = 0; k < nchunks; k++) zc_fill_chunk(kind, seed, first_chunk + k, out + (size_t)k * ZC_CHUNK);
}
uint32_t zc_host_class_of(uint64_t chunk_index) { return zc_class_of(chunk_index); }
