// zgpu_lz_parallel.hip -- LZ77 stage, parallel form for levels 4-9 (placeholder until the kernels land).
#include "zgpu_common.h"
namespace zgpu {
bool lz_parallel_available() { return false; }
size_t lz_parallel_workspace_bytes(uint32_t) { return 256; }
void launch_lz_parallel(const ChunkGeom &, LevelCfg, void *, uint32_t *, ChunkMeta *, hipStream_t, void *) {}
} // namespace zgpu
