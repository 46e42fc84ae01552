// zgpu_inflate.hip -- placeholder until the decode kernel lands.
#include "zgpu_common.h"
#include "../../include/zamd_gpu.h"
struct zgpu_engine;
namespace zgpu {
int inflate_run(zgpu_engine *, const uint8_t *, uint64_t, const uint64_t *, uint64_t, uint32_t, uint8_t *, uint64_t, zgpu_inflate_result *, hipStream_t) { return ZGPU_STREAM_ERROR; }
} // namespace zgpu
extern "C" {
#pragma GCC visibility push(default)
int zgpu_inflate_find_chunks_host(zgpu_engine *, const void *, uint64_t, uint32_t, uint64_t *, uint64_t, uint64_t *) { return ZGPU_STREAM_ERROR; }
const char *zgpu_inflate_message(uint32_t) { return ""; }
#pragma GCC visibility pop
}
