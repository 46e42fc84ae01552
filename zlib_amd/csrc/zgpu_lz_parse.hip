// zgpu_lz_parse.hip -- the kernels around the parallel parse (zgpu_lz_parse.h, zgpu_lz_parse_body.inc): the deflate_slow parse over
// match3's records, and its lite form over the games walk_kernel has played (used when the two are not fused, ZGPU_WALK_FUSE=0).
#include "zgpu_lz_parse.h"

namespace zgpu {

// LITE: the games have been played by walk_kernel (zgpu_lz_sorted.hip): gmv[r] holds the game of every position r a walker stood on
// with nothing in hand and found a match at (bit r of gsv), in the format of gm[] below -- a subset of the has-positions that
// contains the whole path, which is all that stages A2..D look at.
template <bool LITE, bool TILE>
__global__ void __launch_bounds__(1024, 8) parse2_kernel(ChunkGeom g, LevelCfg cfg, const uint2 *__restrict__ recs, const uint32_t *__restrict__ gmv_all,
                                                         const uint32_t *__restrict__ gsv_all, uint32_t *__restrict__ tokens, ChunkMeta *meta, TileGeom tg)
{
    constexpr uint32_t kP2Threads = 1024;
    constexpr bool FUSED = false;
    __shared__ __attribute__((aligned(16))) uint16_t J[kP2Win]; // successor of a has-position, window-relative (0xffff: leaves the window)
    __shared__ uint32_t HAS[kP2Words], MARK[kP2Words], COV[kP2Words], MAT[kP2Words]; // per position: has a match / on the path / inside a match
                                                                                    // (later: is a token) / starts an emitted match
    __shared__ uint32_t wbase[kP2Words + 1];                   // tokens in front of each 32-position word
    __shared__ uint32_t VIS[kP2Win / 32], EXITS[64];
    __shared__ uint32_t wave_tot[kP2Threads / 64];
    __shared__ uint32_t sh_entry, sh_exit;
    const uint32_t c = blockIdx.x;
#include "zgpu_lz_parse_body.inc"
}

void launch_parse2(const ChunkGeom &g, LevelCfg cfg, const uint2 *recs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st)
{
    hipLaunchKernelGGL((parse2_kernel<false, false>), dim3(g.nchunks), dim3(1024), 0, st, g, cfg, recs, nullptr, nullptr, tokens, meta, TileGeom{});
}

void launch_parse_lite(const ChunkGeom &g, LevelCfg cfg, const uint32_t *gm, const uint32_t *gs, uint32_t *tokens, ChunkMeta *meta, hipStream_t st)
{
    hipLaunchKernelGGL((parse2_kernel<true, false>), dim3(g.nchunks), dim3(1024), 0, st, g, cfg, nullptr, gm, gs, tokens, meta, TileGeom{});
}

// a tile of a continuous stream: the games walk_kernel<2> has played, the entry the chain over the tiles' exits has found (zgpu_cont.hip)
void launch_parse_tile(const ChunkGeom &g, LevelCfg cfg, const uint32_t *gm, const uint32_t *gs, uint32_t *tokens, ChunkMeta *meta, const TileGeom &tg, hipStream_t st)
{
    hipLaunchKernelGGL((parse2_kernel<true, true>), dim3(g.nchunks), dim3(1024), 0, st, g, cfg, nullptr, gm, gs, tokens, meta, tg);
}

} // namespace zgpu
