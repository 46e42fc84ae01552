// zgpu_engine.hip -- host side of the C ABI in include/zamd_gpu.h: workspace ownership, batching, kernel
// sequencing on one HIP stream, per-stage HIP-event timing.  No torch, no C++ types cross the boundary.
#include "zgpu_common.h"
#include "../../include/zamd_gpu.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

namespace zgpu {

struct RunStateHost { uint64_t out_total, in_total, ntokens; uint32_t adler_a, adler_b, data_type, overflow, crc, pad; };

// kernels (other translation units)
void launch_lz_serial(const ChunkGeom &g, LevelCfg cfg, uint4 *tables, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, uint32_t *nostore_bits, bool hand_on, uint32_t tag);
void launch_collect_handed_on(const ChunkMeta *meta, uint32_t n, uint32_t *list, uint32_t *count, hipStream_t st);
void launch_huffman(const ChunkGeom &g, const uint32_t *tokens, ChunkMeta *meta, uint8_t *slots, hipStream_t st, bool fixed_trees);
void launch_adler(const ChunkGeom &g, ChunkMeta *meta, hipStream_t st);
void launch_crc(const ChunkGeom &g, ChunkMeta *meta, hipStream_t st);
void launch_scan(const ChunkMeta *meta, uint32_t nchunks, uint64_t chunk0, uint64_t *offsets, void *run, uint64_t out_cap, hipStream_t st, bool with_crc = false);
void launch_stitch(const uint8_t *slots, const ChunkMeta *meta, const uint64_t *offsets, uint64_t chunk0, uint32_t nchunks, uint8_t *out,
                   uint64_t out_cap, uint32_t slot_stride, hipStream_t st);
void launch_corpus(uint32_t kind, uint64_t seed, uint64_t first_chunk, uint64_t nchunks, uint8_t *out, hipStream_t st);
bool lz_parallel_available();
size_t lz_parallel_workspace_bytes(uint32_t batch_chunks);
void launch_lz_parallel(const ChunkGeom &g, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, void *prof);
size_t lz_sorted_workspace_bytes(uint32_t batch_chunks);
bool launch_lz_sorted(const ChunkGeom &g, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, void *prof, int exact_sort, int walk);
bool lz_fastwin_serves(const LevelCfg &cfg);
uint32_t *lz_sorted_fault_word(void *workspace);
// continuous stream (zgpu_cont.hip, zgpu_lz_sorted.hip)
void launch_lz_tiles(const ChunkGeom &g, const TileGeom &tg, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, uint16_t *comp, uint16_t *gentry, hipStream_t st,
                     void *prof, int exact_sort);
uint32_t chain_groups(uint32_t ntiles);
void launch_lz_tiles_parse(const ChunkGeom &g, const TileGeom &tg, LevelCfg cfg, void *workspace, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, void *prof);
void launch_sort_tiles(const ChunkGeom &g, void *workspace, ChunkMeta *meta, hipStream_t st, void *prof, int exact_sort, const uint16_t **S_out, const uint32_t **ir_out);
void launch_lz_fastwin_tiles(const ChunkGeom &g, const TileGeom &tg, const FastTiles &ft, LevelCfg cfg, const uint16_t *S, const uint32_t *ir, uint32_t *tokens, ChunkMeta *meta, uint32_t ngrid,
                             hipStream_t st);
void launch_fast_init(uint8_t *cur, uint8_t *active, uint16_t *exit_cur, uint32_t n, hipStream_t st);
void launch_fast_flip_list(uint8_t *cur, uint16_t *exit_cur, const uint16_t *exit_new, const uint32_t *list, uint32_t n, hipStream_t st);
void launch_fast_flip(uint8_t *cur, const uint8_t *active, uint8_t *active_next, const uint8_t *changed, uint16_t *exit_cur, const uint16_t *exit_new, uint32_t n, uint32_t round,
                      uint32_t *count, uint32_t *list, const uint8_t *kept, hipStream_t st);
void launch_fast_finish(const uint8_t *cur, const uint16_t *exit_cur, const uint32_t *ins0, const uint32_t *ins1, uint32_t n, uint16_t *entry_after, uint32_t *prev_ins, uint32_t *prev_prev_ins,
                        const uint32_t *low, hipStream_t st);
void launch_fast_hist(const uint32_t *before, const uint32_t *last, uint32_t x0, uint32_t count, uint32_t *out, hipStream_t st);
void launch_cont_tokens(const ChunkGeom &g, const TileGeom &tg, const uint32_t *tokens, const ChunkMeta *tmeta, ContState *st, uint32_t *tokoff, const uint32_t *carry, uint32_t *T,
                        ContBlk *blk, uint64_t seg_end, bool final_block, uint64_t sp, uint32_t nblk_cap, bool slow, hipStream_t s);
void launch_huffman_cont(const ChunkGeom &g, const uint32_t *compact_tokens, uint32_t nblk, ContBlk *blk, ContState *cst, uint8_t *slots, hipStream_t st, bool fixed_trees);
void launch_cont_stitch(const ContBlk *blk, ContState *st, uint64_t *pos, const uint8_t *slots, uint32_t slot_stride, const uint8_t *in, uint64_t abs0, uint8_t *out, uint64_t out_cap,
                        uint32_t nblk_cap, const uint32_t *T, uint32_t *carry, uint64_t seg_end, hipStream_t s);
uint64_t cont_special_pos(uint64_t n);
int inflate_run(struct ::zgpu_engine *e, const uint8_t *d_in, uint64_t in_bytes, const uint64_t *d_offsets, uint64_t nchunks,
                uint32_t chunk_size, uint8_t *d_out, uint64_t out_cap, zgpu_inflate_result *res, hipStream_t st, uint32_t stream_mode = 0,
                const uint64_t *h_offsets = nullptr, bool open_end = false, uint8_t *h_dst = nullptr, uint64_t h_cap = 0);

} // namespace zgpu

struct StageSpan { int stage; hipEvent_t a, b; };

struct zgpu_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr; // H2D of the *_host entry points: the next batch's input moves while this batch's kernels run
    std::vector<hipEvent_t> copy_ev;
    hipStream_t d2h_stream = nullptr;  // D2H of zgpu_deflate_host: finished batches' bytes go home while later batches are compressed (a second host thread)
    std::vector<hipEvent_t> done_ev;
    uint64_t *pin_tot = nullptr;       // pinned: out_total behind every batch
    char err[512] = {0};
    // deflate workspace, sized for `batch_cap` chunks
    uint32_t batch_cap = 0;
    uint32_t *tokens = nullptr;
    zgpu::ChunkMeta *meta = nullptr;
    uint8_t *slots = nullptr;
    uint4 *tables = nullptr;     // serial LZ only: zeroed when allocated, after that `serial_tag` tells one launch's buckets from another's (SerialLzT::insert)
    uint32_t serial_tag = 0;
    uint32_t tables_cap = 0;
    void *par_ws = nullptr;      // parallel LZ only
    int tuned = 0; uint32_t tune[4] = {0, 0, 0, 0}; // zgpu_deflate_set_tuning: good, lazy, nice, chain instead of the level's
    uint64_t handed_on = 0; // (diagnostic: chunks handed on since the engine was made)
    uint32_t *hand_list = nullptr; uint32_t hand_cap = 0; // chunks the lane-per-chunk loop handed on: [0] their number, [1..] their indices in the batch
    int geo_w = 15, geo_m = 8;   // zgpu_deflate_set_geometry: deflateInit2's windowBits and memLevel
    uint8_t *geo_slots = nullptr; uint4 *geo_tables = nullptr; uint32_t *geo_nostore = nullptr; uint32_t geo_cap = 0; // the workspace of a non-default geometry
    int exact_sort = 0;          // sticky: the fast sort's self-check failed once on this engine (zgpu_lz_sorted.hip, pass V)
    uint32_t par_cap = 0;
    uint64_t *offsets = nullptr; // nchunks+1 segment offsets of the current call
    uint64_t offsets_cap = 0;
    void *run = nullptr;         // RunState
    // staging for the *_host entry points
    uint8_t *stage_in = nullptr, *stage_out = nullptr;
    uint64_t stage_in_cap = 0, stage_out_cap = 0;
    // inflate scratch
    void *inf_status = nullptr; uint64_t inf_status_cap = 0;
    zgpu::ChunkMeta *inf_meta = nullptr; uint32_t inf_meta_cap = 0;
    uint64_t *inf_offs = nullptr; uint64_t inf_offs_cap = 0;
    void *inf_slots = nullptr; uint64_t inf_slots_cap = 0;
    uint8_t *inf_dict = nullptr; uint32_t inf_dict_len = 0; // preset dictionary of the next inflate calls (zgpu_inflate_set_dictionary)
    uint32_t inf_checks = 3;                                // checks of the decoded bytes (zgpu_inflate_set_checks)
    // continuous stream (deflate_cont): per batch of tiles / per feed
    uint32_t ct_tiles = 0, ct_nblk = 0; uint64_t ct_feed_tiles = 0;
    uint16_t *ct_exits = nullptr, *ct_entry = nullptr, *ct_comp = nullptr, *ct_gentry = nullptr;
    uint32_t *ct_tokoff = nullptr, *ct_T = nullptr, *ct_carry = nullptr, *ct_carry_in = nullptr;
    zgpu::ContBlk *ct_blk = nullptr; uint64_t *ct_pos = nullptr; uint8_t *ct_slots = nullptr; zgpu::ContState *ct_st = nullptr;
    zgpu::ChunkMeta *ct_ckmeta = nullptr; uint64_t ct_ck_cap = 0;
    uint64_t *ct_excl = nullptr; uint32_t ct_excl_cap = 0;
    // ... levels 1-3: the rounds of fastwin_tile_kernel
    uint32_t cf_tiles = 0;
    uint16_t *cf_exit_a = nullptr, *cf_exit_b = nullptr; uint32_t *cf_ins0 = nullptr, *cf_ins1 = nullptr, *cf_prev = nullptr, *cf_prev2 = nullptr, *cf_hist = nullptr, *cf_count = nullptr;
    uint8_t *cf_cur = nullptr, *cf_act_a = nullptr, *cf_act_b = nullptr, *cf_changed = nullptr, *cf_kept = nullptr; uint32_t *cf_list_a = nullptr, *cf_list_b = nullptr, *cf_used = nullptr; uint16_t *cf_entry_used = nullptr;
    hipStream_t ct_stream = nullptr; hipEvent_t ct_ev_a[2] = {nullptr, nullptr}, ct_ev_b[2] = {nullptr, nullptr}; // levels 4-9: a batch's blocks are made on a second stream under the next batch's walkers
    uint64_t cf_rounds = 0, cf_tile_parses = 0; // (diagnostic: rounds and tile parses since the engine was made)
    // profiling
    bool prof = false;
    double ms[ZGPU_STAGE_COUNT] = {0};
    uint64_t launches[ZGPU_STAGE_COUNT] = {0};
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<StageSpan> spans;
};

namespace zgpu {

int fail_hip(zgpu_engine *e, hipError_t err, const char *what, const char *file, int line)
{
    if (e) snprintf(e->err, sizeof e->err, "HIP error %d (%s) at %s:%d: %s", (int)err, hipGetErrorString(err), file, line, what);
    if (err == hipErrorOutOfMemory) return ZGPU_MEM_ERROR;
    return ZGPU_ERRNO;
}

static int fail(zgpu_engine *e, int code, const char *msg)
{
    if (e) snprintf(e->err, sizeof e->err, "%s", msg);
    return code;
}

static hipEvent_t next_event(zgpu_engine *e)
{
    if (e->ev_used == e->ev_pool.size()) { hipEvent_t ev; hipEventCreate(&ev); e->ev_pool.push_back(ev); }
    return e->ev_pool[e->ev_used++];
}

// stage timing: HIP events recorded on the launch stream around the stage's launches
struct StageTimer {
    zgpu_engine *e; hipStream_t st; int stage; hipEvent_t a{};
    StageTimer(zgpu_engine *e_, hipStream_t st_, int stage_) : e(e_), st(st_), stage(stage_)
    {
        if (e->prof) { a = next_event(e); hipEventRecord(a, st); }
    }
    ~StageTimer()
    {
        if (e->prof) { hipEvent_t b = next_event(e); hipEventRecord(b, st); e->spans.push_back({stage, a, b}); }
    }
};

static void collect_spans(zgpu_engine *e)
{
    for (auto &s : e->spans) { float ms = 0; if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { e->ms[s.stage] += ms; e->launches[s.stage]++; } }
    e->spans.clear(); e->ev_used = 0;
}

template <typename T> static int dev_alloc(zgpu_engine *e, T **p, size_t count)
{
    ZGPU_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return ZGPU_OK;
}

static uint32_t env_u32(const char *name, uint32_t dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    long x = strtol(v, nullptr, 10);
    return x > 0 ? (uint32_t)x : dflt;
}

static int ensure_deflate_ws(zgpu_engine *e, uint32_t batch, bool serial, uint64_t nchunks_total, bool geo = false)
{
    int rc;
    if (geo && batch > e->geo_cap) { // wider slots (blocks that may not be stored grow), head[] of up to 65536 entries, a flag word array per chunk
        hipFree(e->geo_slots); hipFree(e->geo_tables); hipFree(e->geo_nostore); e->geo_slots = nullptr; e->geo_tables = nullptr; e->geo_nostore = nullptr; e->geo_cap = 0;
        if ((rc = dev_alloc(e, &e->geo_slots, (size_t)batch * kGeoSlotStride))) return rc;
        if ((rc = dev_alloc(e, &e->geo_tables, (size_t)batch * kGeoTableEntries))) return rc;
        ZGPU_HIP_CHECK(hipMemsetAsync(e->geo_tables, 0, (size_t)batch * kGeoTableEntries * sizeof(uint4), e->stream));
        ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
        if ((rc = dev_alloc(e, &e->geo_nostore, (size_t)batch * kGeoNostoreWords))) return rc;
        e->geo_cap = batch;
    }
    if (batch > e->batch_cap) {
        hipFree(e->tokens); hipFree(e->meta); hipFree(e->slots); e->tokens = nullptr; e->meta = nullptr; e->slots = nullptr; e->batch_cap = 0;
        if ((rc = dev_alloc(e, &e->tokens, (size_t)batch * kChunkMax))) return rc;
        if ((rc = dev_alloc(e, &e->meta, (size_t)batch))) return rc;
        if ((rc = dev_alloc(e, &e->slots, (size_t)batch * kSlotStride))) return rc;
        e->batch_cap = batch;
    }
    if (serial && !geo && batch > e->tables_cap) {
        hipFree(e->tables); e->tables = nullptr; e->tables_cap = 0;
        if ((rc = dev_alloc(e, &e->tables, (size_t)batch * kSerialTableEntries))) return rc;
        ZGPU_HIP_CHECK(hipMemsetAsync(e->tables, 0, (size_t)batch * kSerialTableEntries * sizeof(uint4), e->stream));
        ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
        e->tables_cap = batch;
    }
    if (!serial && batch > e->par_cap) {
        hipFree(e->par_ws); e->par_ws = nullptr; e->par_cap = 0;
        const size_t wa = lz_parallel_workspace_bytes(batch), wb = lz_sorted_workspace_bytes(batch);
        ZGPU_HIP_CHECK(hipMalloc(&e->par_ws, wa > wb ? wa : wb));
        e->par_cap = batch;
    }
    if (nchunks_total + 1 > e->offsets_cap) {
        hipFree(e->offsets); e->offsets = nullptr; e->offsets_cap = 0;
        if ((rc = dev_alloc(e, &e->offsets, (size_t)nchunks_total + 1))) return rc;
        e->offsets_cap = nchunks_total + 1;
    }
    return ZGPU_OK;
}

static void zlib_header(int level, int strategy, uint8_t hdr[2]) // qcsrc/deflate.c:625-641
{
    unsigned h = (8u + (7u << 4)) << 8, lf = (strategy >= 2 || level < 2) ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
    h |= lf << 6; h += 31 - h % 31;
    hdr[0] = (uint8_t)(h >> 8); hdr[1] = (uint8_t)h;
}

// d_seg (optional): device table of nseg+1 offsets; then every segment is one chunk and chunk_size is ignored.
// h_src (optional, uniform chunking only): the input still lies in host memory; every batch's bytes are copied to d_in on the engine's
// copy stream right before the batch's kernels are queued, so the copy of batch k+1 runs under the kernels of batch k.
static int deflate_device(zgpu_engine *e, const uint8_t *d_in, uint64_t in_bytes, const uint64_t *d_seg, uint64_t nseg,
                          const zgpu_deflate_params *p, uint8_t *d_out, uint64_t out_cap, uint64_t *d_chunk_offsets,
                          zgpu_deflate_result *res, hipStream_t st, uint32_t skip0 = 0, const uint8_t *h_src = nullptr, uint8_t *h_dst = nullptr,
                          uint64_t h_cap = 0, uint64_t *h_copied = nullptr)
{
    if (!e || !p || !res || (!d_in && in_bytes) || !d_out) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    if (p->level < 1 || p->level > 9) return fail(e, ZGPU_STREAM_ERROR, "level must be 1..9");
    const uint32_t chunk_size = p->chunk_size ? p->chunk_size : kChunkMax;
    if (chunk_size > kChunkMax) return fail(e, ZGPU_STREAM_ERROR, "chunk_size must be 1..65536");
    if ((p->flags & (ZGPU_F_ZLIB_WRAP | ZGPU_F_GZIP_WRAP)) && !(p->flags & ZGPU_F_FINAL)) return fail(e, ZGPU_STREAM_ERROR, "a wrapper needs FINAL");
    if ((p->flags & ZGPU_F_ZLIB_WRAP) && (p->flags & ZGPU_F_GZIP_WRAP)) return fail(e, ZGPU_STREAM_ERROR, "one wrapper at a time");
    if (p->prime && ((p->prime >> 16) > 16 || (p->flags & (ZGPU_F_ZLIB_WRAP | ZGPU_F_GZIP_WRAP)))) return fail(e, ZGPU_STREAM_ERROR, "prime: at most 16 bits, no wrapper");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    if (p->strategy < 0 || p->strategy > (int)kFixed) return fail(e, ZGPU_STREAM_ERROR, "strategy must be 0..4");
    LevelCfg cfg = level_cfg(p->level);
    cfg.strategy = (uint32_t)p->strategy;
    if (e->tuned) { // deflateTune (deflate.c:453-470): the level keeps its function, the four parameters are the caller's
        // (a budget of 0 never runs out in the reference: its loop counts down past zero, deflate.c:1163 -- no chain of a chunk is longer than 65535)
        cfg.good = e->tune[0]; cfg.lazy = e->tune[1]; cfg.nice = e->tune[2]; cfg.chain = (e->tune[3] && e->tune[3] < 0xffffu) ? e->tune[3] : 0xffffu;
        if (cfg.nice > kMaxMatch) cfg.nice = kMaxMatch; // (no match is longer: the same searches)
    }
    // the chain budget of the all-position search says it all for two strategies (deflate.c:1594-1599): no candidate at all,
    // or the nearest one only (and then only at distance 1, see match3_kernel)
    if (cfg.slow && cfg.strategy == kHuffmanOnly) cfg.chain = 0;
    if (cfg.slow && cfg.strategy == kRle) cfg.chain = 1;
    // deflateInit2's geometry (zgpu_deflate_set_geometry): anything but windowBits 15 / memLevel 8 goes to the lane-per-chunk loop, the one
    // implementation whose window size, hash width and block length are run-time values
    const bool geo = e->geo_w != 15 || e->geo_m != 8;
    if (geo) { cfg.w_bits = (uint32_t)e->geo_w; cfg.hash_bits = (uint32_t)e->geo_m + 7; }
    const uint32_t max_dist = (geo ? 1u << e->geo_w : kWSize) - kMinLookahead;
    int impl = p->lz_impl;
    if (geo && impl != ZGPU_LZ_AUTO && impl != ZGPU_LZ_SERIAL) return fail(e, ZGPU_STREAM_ERROR, "a non-default windowBits / memLevel is served by ZGPU_LZ_SERIAL only");
    // the parse-driven search plays deflate_slow's own game; the two strategies that are chain budgets of the all-position search
    // (Z_HUFFMAN_ONLY, Z_RLE) stay with that search
    const bool walk_ok = cfg.strategy != kHuffmanOnly && cfg.strategy != kRle;
    // levels 1-3 above the crossover: the lane-per-chunk loop hands the chunks that do not compress on to the wave-per-chunk kernel (lz_serial_chunk)
    bool hand_on = false;
    if (impl == ZGPU_LZ_AUTO) {
        static int auto_env = -1; // ZGPU_LZ_DEFAULT=3: A/B runs of the all-position search
        if (auto_env < 0) { const char *v = getenv("ZGPU_LZ_DEFAULT"); auto_env = v ? atoi(v) : 0; }
        impl = (cfg.slow && lz_parallel_available()) ? ((walk_ok && auto_env != ZGPU_LZ_SORTED) ? ZGPU_LZ_WALK : ZGPU_LZ_SORTED) : ZGPU_LZ_SERIAL;
        // levels 1-3: the head[]/prev[] loop.  deflate_fast on the sorted buckets (ZGPU_LZ_FAST, fast_kernel in zgpu_lz_sorted.hip) is a second
        // implementation for the parity tests and for ZGPU_LZ_DEFAULT=5: it asks memory four times less often and is no faster
        if (!cfg.slow && lz_parallel_available() && walk_ok && !skip0 && auto_env == ZGPU_LZ_FAST) impl = ZGPU_LZ_FAST;
        // levels 1-3: a wave per chunk, window and chain bits in LDS (zgpu_lz_fastwin.hip) -- 5 ms a chunk whatever the size of the call, three chunks per
        // CU; the lane-per-chunk loop takes 28 ms a chunk and needs tens of thousands of them in flight.  Measured crossovers against the loop with its
        // hand-on (chunks per launch, 4 GiB = 65536; profiles/r03_crossover_levels_1_3.txt): levels 1 and 2 about 16000 -- a host batch --, level 3 (32
        // candidates a lane) about 3000.  ZGPU_LZ_DEFAULT=1 keeps the loop, =6 the waves.
        if (!cfg.slow && lz_parallel_available() && walk_ok && !skip0 && lz_fastwin_serves(cfg) && auto_env != ZGPU_LZ_FAST && auto_env != ZGPU_LZ_SERIAL) {
            const uint64_t cs0 = p->chunk_size ? p->chunk_size : kChunkMax;
            uint64_t nch0 = d_seg ? nseg : (in_bytes + cs0 - 1) / cs0;
            { // what counts is the chunks of ONE launch: host input goes batch by batch (see below), and every batch of the loop would pay its latency again
                const uint64_t per_launch = h_src ? env_u32("ZGPU_HOST_BATCH", 65536) : env_u32("ZGPU_BATCH_CHUNKS", 65536); // (levels 1-3: see host_batch below)
                if (per_launch && nch0 > per_launch) nch0 = per_launch;
            }
            // (round 4: without the ring in LDS eleven chunks share a CU and the waves' form scales with the launch -- level 1: 1 GiB 58.6 ms against the loop's 150.9, 3 GiB 164 against 225,
            // 4 GiB 217.6 against 213 with the hand-on; level 2: 2 GiB 160 against 208, 3 GiB 237 against 240; level 3: 256 MiB 110 against 149, 1 GiB 328 against 299)
            const uint64_t upto = cfg.chain == 4 ? 61440 : cfg.chain == 8 ? 45056 : 10240;
            const char *ho = getenv("ZGPU_HAND_ON"); // 0: the loop keeps every chunk (A/B runs); 2: the loop + hand-on whatever the size of the call (tests)
            if (ho && ho[0] == '2' && auto_env == 0) hand_on = true;
            else if (nch0 <= upto || auto_env == ZGPU_LZ_FASTWIN) impl = ZGPU_LZ_FASTWIN;
            else hand_on = !(ho && ho[0] == '0');
        }
    }
    if ((impl == ZGPU_LZ_PARALLEL || impl == ZGPU_LZ_SORTED || impl == ZGPU_LZ_WALK) && (!cfg.slow || !lz_parallel_available()))
        return fail(e, ZGPU_STREAM_ERROR, "parallel LZ77 serves levels 4..9 only");
    if (impl == ZGPU_LZ_FAST && (cfg.slow || !walk_ok || skip0 || !lz_parallel_available())) return fail(e, ZGPU_STREAM_ERROR, "ZGPU_LZ_FAST serves levels 1..3, not Z_HUFFMAN_ONLY / Z_RLE, no dictionary chunk");
    if (impl == ZGPU_LZ_FASTWIN && (cfg.slow || !walk_ok || skip0 || !lz_parallel_available() || !lz_fastwin_serves(cfg)))
        return fail(e, ZGPU_STREAM_ERROR, "ZGPU_LZ_FASTWIN serves levels 1..3 with their own parameters, not Z_HUFFMAN_ONLY / Z_RLE, no dictionary chunk");
    if (impl < ZGPU_LZ_SERIAL || impl > ZGPU_LZ_FASTWIN) return fail(e, ZGPU_STREAM_ERROR, "unknown lz_impl");
    if (impl == ZGPU_LZ_PARALLEL && p->strategy != 0) return fail(e, ZGPU_STREAM_ERROR, "ZGPU_LZ_PARALLEL serves the default strategy only");
    if (impl == ZGPU_LZ_WALK && !walk_ok) return fail(e, ZGPU_STREAM_ERROR, "ZGPU_LZ_WALK does not serve Z_HUFFMAN_ONLY / Z_RLE");
    if (skip0) { // a preset dictionary in front of the one chunk: the lane-per-chunk loop is the implementation that starts mid-window
        if (d_seg || in_bytes > kChunkMax || skip0 < kMinMatch || skip0 > max_dist || skip0 > in_bytes || (p->flags & (ZGPU_F_ZLIB_WRAP | ZGPU_F_GZIP_WRAP | ZGPU_F_POS0 | ZGPU_F_POS0_ALL)))
            return fail(e, ZGPU_STREAM_ERROR, "dictionary chunk: 3..32506 dictionary bytes + data <= 65536, no wrapper");
        impl = ZGPU_LZ_SERIAL;
    }
    if (geo) impl = ZGPU_LZ_SERIAL;
    const bool serial = impl == ZGPU_LZ_SERIAL;
    hand_on = hand_on && serial && !geo && p->lz_impl == ZGPU_LZ_AUTO;
    if (d_seg && ((p->flags & (ZGPU_F_ZLIB_WRAP | ZGPU_F_GZIP_WRAP)) || nseg == 0)) return fail(e, ZGPU_STREAM_ERROR, "segment mode: no zlib wrapper, nseg >= 1");
    const uint64_t nchunks = d_seg ? nseg : (in_bytes ? (in_bytes + chunk_size - 1) / chunk_size : 1);
    // One batch = one launch of every stage.  The lane-per-chunk stages (serial LZ77, parse) need tens of thousands of
    // chunks in flight to fill 256 CUs, so batches are as large as device memory allows (~1 MiB of workspace per chunk).
    uint32_t batch_max = env_u32("ZGPU_BATCH_CHUNKS", 65536); // (small values are for tests: several launches per call)
    if (batch_max == 0) batch_max = 1;
    // host input: the copy of batch k+1 runs under the kernels of batch k.  Batches must stay large: the block-construction and sort kernels are
    // latency-bound and need every workgroup slot of the chip filled several times over (2048 chunks per launch: 2.2x the time per byte of
    // 65536, rocprofv3 timeline of scripts/host_trace.py); the copy moves 57 GB/s and is over long before the kernels are
    // (levels 1-3: both of their kernels want the launch as large as it gets -- the loop's time hardly grows with the chunks, 170 ms for 32768 and 245 for
    // 65536 at level 1 -- and take far longer than the copy they would hide, so host input is not cut up there)
    const uint32_t host_batch = env_u32("ZGPU_HOST_BATCH", cfg.slow ? 16384 : 65536); // (A/B runs; measured at 4 GiB, level 6: 8192 184.8 ms, 16384 176.3, 32768 178.5)
    if (h_src && batch_max > host_batch) batch_max = host_batch ? host_batch : 16384;
    {
        size_t free_b = 0, total_b = 0;
        const size_t per_chunk = (size_t)kChunkMax * 4 + kSlotStride + (geo ? (size_t)kGeoSlotStride + (size_t)kGeoTableEntries * sizeof(uint4) : serial ? (size_t)kSerialTableEntries * sizeof(uint4) + (hand_on ? lz_sorted_workspace_bytes(1) / 4 : 0) : lz_sorted_workspace_bytes(1));
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t held = (size_t)e->batch_cap * per_chunk; // what this engine already owns can be reused
            const size_t budget = (free_b + held) / 10 * 6;
            size_t fit = budget / per_chunk; // chunks that fit; never below 256, whatever the device reports
            if (fit < 256) fit = 256;
            if (fit < batch_max) batch_max = (uint32_t)fit;
        }
    }
    const uint32_t batch = (uint32_t)(nchunks < batch_max ? nchunks : batch_max);
    int rc = ensure_deflate_ws(e, batch, serial, nchunks, geo);
    if (rc) return rc;
    uint32_t hand_piece = 0;
    if (hand_on) {
        hand_piece = batch / 4 < 4096 ? (batch < 4096 ? batch : 4096) : batch / 4; // the sorted buckets' workspace: for a quarter of the batch at a time
        if ((rc = ensure_deflate_ws(e, hand_piece, false, nchunks))) return rc;
        if (e->par_cap > hand_piece) hand_piece = e->par_cap < batch ? e->par_cap : batch; // (an earlier call has left more)
        if (batch > e->hand_cap) {
            hipFree(e->hand_list); e->hand_list = nullptr; e->hand_cap = 0;
            if ((rc = dev_alloc(e, &e->hand_list, (size_t)batch + 1))) return rc;
            e->hand_cap = batch;
        }
    }
    const bool wrap = p->flags & ZGPU_F_ZLIB_WRAP, gz = p->flags & ZGPU_F_GZIP_WRAP;
    const uint32_t head_bytes = wrap ? 2 : gz ? 10 : 0, tail_bytes = wrap ? 4 : gz ? 8 : 0;
    ChunkGeom g{};
    g.in = d_in; g.in_bytes = in_bytes; g.seg_off = d_seg; g.chunk_size = chunk_size;
    g.final_chunk = (!d_seg && (p->flags & ZGPU_F_FINAL)) ? nchunks - 1 : ~0ull;
    g.all_final = (d_seg && (p->flags & ZGPU_F_FINAL)) ? 1u : 0u;
    g.pos0_mode = (p->flags & ZGPU_F_POS0_ALL) ? 2u : (p->flags & ZGPU_F_POS0) ? 1u : 0u;
    g.skip0 = skip0;
    g.block_tokens = geo ? (1u << (e->geo_m + 6)) - 1u : kBlockTokens;
    g.slot_stride = geo ? kGeoSlotStride : kSlotStride;
    g.nostore_bits = geo ? e->geo_nostore : nullptr;
    uint8_t *const slots = geo ? e->geo_slots : e->slots;
    g.prime = (p->prime >> 16) ? ((p->prime & 0xffff0000u) | (p->prime & ((1u << (p->prime >> 16)) - 1u))) : 0u;
    const uint64_t body_cap = tail_bytes ? (out_cap >= head_bytes + tail_bytes ? out_cap - tail_bytes : 0) : out_cap;

    RunStateHost rs{}; rs.out_total = head_bytes; rs.adler_a = 1; rs.adler_b = 0; rs.data_type = 2;
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->run, &rs, sizeof rs, hipMemcpyHostToDevice, st));
    const bool check_sort = (impl == ZGPU_LZ_SORTED || impl == ZGPU_LZ_WALK || impl == ZGPU_LZ_FAST || impl == ZGPU_LZ_FASTWIN || hand_on) && !e->exact_sort;
    uint32_t sort_fault = 0;
    if (check_sort) ZGPU_HIP_CHECK(hipMemsetAsync(lz_sorted_fault_word(e->par_ws), 0, 4, st));
    if (wrap && out_cap >= 2) { uint8_t hdr[2]; zlib_header(p->level, p->strategy, hdr); ZGPU_HIP_CHECK(hipMemcpyAsync(d_out, hdr, 2, hipMemcpyHostToDevice, st)); }
    if (gz && out_cap >= 10) { // the header deflate() writes when no gz_header was set (qcsrc/deflate.c:578-596); OS_CODE 3 as the reference builds here
        const uint8_t hdr[10] = {31, 139, 8, 0, 0, 0, 0, 0, (uint8_t)(p->level == 9 ? 2 : (p->strategy >= 2 || p->level < 2) ? 4 : 0), 3};
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_out, hdr, 10, hipMemcpyHostToDevice, st));
    }

    // h_dst: what the batches so far have produced goes to the caller's buffer while the next ones are compressed.  Copies to pageable memory
    // hold the calling thread, so they are a second thread's: it waits for batch k's event, reads the stream length behind it (pinned)
    // and copies the new bytes on a stream of its own.
    // host input, several batches: nothing runs under the first one's copy, so it is short (2048 chunks) and the next ones double up to the full
    // batch -- a copy moves a chunk 2.3 times as fast as the kernels compress one, so each batch's copy still ends under the kernels of the batch
    // before it -- and what the short launches cost the latency-bound kernels (see above) is less than what the shorter wait saves
    // (1 GiB: 58.3 -> 55.5 ms; profiles/r02_host_batches_ab.txt)
    const uint32_t first_env = env_u32("ZGPU_FIRST_BATCH", 2048); // (chunks; 0: all batches alike, for A/B runs)
    const uint32_t first = (h_src && cfg.slow && first_env && batch >= 2 * first_env && nchunks > first_env) ? first_env : batch; // (levels 1-3: no ramp either)
    auto next_step = [batch](uint32_t s) { return s >= batch / 2 ? batch : s * 2; };
    size_t nbatches = 0;
    for (uint64_t c0 = 0, step = first; c0 < nchunks; c0 += step, step = next_step((uint32_t)step)) nbatches++;
    std::thread homer;
    std::atomic<size_t> issued{0};
    std::atomic<int> homer_rc{0};
    uint64_t homed = 0;
    const bool home = h_dst && h_src && nbatches > 1 && nbatches <= 4096;
    struct HomeGuard { // (the checks below return from the middle of the loop: the thread must not outlive the call)
        std::thread &t; std::atomic<size_t> &n;
        ~HomeGuard() { if (t.joinable()) { n.store((size_t)-1); t.join(); } }
    } home_guard{homer, issued};
    if (home) {
        if (!e->pin_tot) ZGPU_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&e->pin_tot), 4096 * sizeof(uint64_t), hipHostMallocDefault));
        if (!e->d2h_stream) ZGPU_HIP_CHECK(hipStreamCreateWithFlags(&e->d2h_stream, hipStreamNonBlocking));
        while (e->done_ev.size() < nbatches) { hipEvent_t ev; ZGPU_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e->done_ev.push_back(ev); }
        homer = std::thread([&, nbatches]() {
            if (hipSetDevice(e->device) != hipSuccess) { homer_rc = 1; return; }
            for (size_t k = 0; k < nbatches; k++) {
                while (issued.load(std::memory_order_acquire) <= k) std::this_thread::yield();
                if (issued.load() == (size_t)-1) return; // the call gave up
                if (hipEventSynchronize(e->done_ev[k]) != hipSuccess) { homer_rc = 1; return; }
                uint64_t upto = e->pin_tot[k];
                if (upto > h_cap) upto = h_cap;
                if (upto > homed) {
                    if (hipMemcpyAsync(h_dst + homed, d_out + homed, upto - homed, hipMemcpyDeviceToHost, e->d2h_stream) != hipSuccess ||
                        hipStreamSynchronize(e->d2h_stream) != hipSuccess) { homer_rc = 1; return; }
                    homed = upto;
                }
            }
        });
    }
    size_t nbatch = 0;
    for (uint64_t c0 = 0, step = first; c0 < nchunks; c0 += step, step = next_step((uint32_t)step), nbatch++) {
        const uint32_t nb = (uint32_t)(nchunks - c0 < step ? nchunks - c0 : step);
        g.chunk0 = c0; g.nchunks = nb;
        if (h_src && in_bytes) { // this batch's input: host -> device on the copy stream, the kernels below wait for it
            const uint64_t lo = c0 * chunk_size, hi = (c0 + nb) * (uint64_t)chunk_size < in_bytes ? (c0 + nb) * (uint64_t)chunk_size : in_bytes;
            while (e->copy_ev.size() <= nbatch) { hipEvent_t ev; ZGPU_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e->copy_ev.push_back(ev); }
            ZGPU_HIP_CHECK(hipMemcpyAsync(const_cast<uint8_t *>(d_in) + lo, h_src + lo, hi - lo, hipMemcpyHostToDevice, e->copy_stream));
            ZGPU_HIP_CHECK(hipEventRecord(e->copy_ev[nbatch], e->copy_stream));
            ZGPU_HIP_CHECK(hipStreamWaitEvent(st, e->copy_ev[nbatch], 0));
        }
        bool adler_done = false;
        if (serial) {
            StageTimer t(e, st, ZGPU_STAGE_LZ_SERIAL);
            // (a tag that these tables have not seen: 0 is what fresh tables hold; should the counter ever wrap, they are zeroed again)
            if (++e->serial_tag == 0) {
                if (e->tables) ZGPU_HIP_CHECK(hipMemsetAsync(e->tables, 0, (size_t)e->tables_cap * kSerialTableEntries * sizeof(uint4), st));
                if (e->geo_tables) ZGPU_HIP_CHECK(hipMemsetAsync(e->geo_tables, 0, (size_t)e->geo_cap * kGeoTableEntries * sizeof(uint4), st));
                e->serial_tag = 1;
            }
            if (geo) launch_lz_serial(g, cfg, e->geo_tables, e->tokens, e->meta, st, e->geo_nostore, false, e->serial_tag);
            else launch_lz_serial(g, cfg, e->tables, e->tokens, e->meta, st, nullptr, hand_on, e->serial_tag);
            if (hand_on) { // the chunks the loop gave up, as a list, through the wave-per-chunk kernel (their number decides the launch: one word comes home)
                launch_collect_handed_on(e->meta, nb, e->hand_list + 1, e->hand_list, st);
                uint32_t handed = 0;
                ZGPU_HIP_CHECK(hipMemcpyAsync(&handed, e->hand_list, 4, hipMemcpyDeviceToHost, st));
                ZGPU_HIP_CHECK(hipStreamSynchronize(st));
                for (uint32_t at = 0; at < handed; at += hand_piece) {
                    ChunkGeom gl = g; gl.nchunks = handed - at < hand_piece ? handed - at : hand_piece; gl.chunk_map = e->hand_list + 1 + at;
                    launch_lz_sorted(gl, cfg, e->par_ws, e->tokens, e->meta, st, e, e->exact_sort, 3);
                }
                e->handed_on += handed;
            }
        } else {
            if (impl == ZGPU_LZ_SORTED || impl == ZGPU_LZ_WALK || impl == ZGPU_LZ_FAST || impl == ZGPU_LZ_FASTWIN)
                adler_done = launch_lz_sorted(g, cfg, e->par_ws, e->tokens, e->meta, st, e, e->exact_sort, impl == ZGPU_LZ_WALK ? 1 : impl == ZGPU_LZ_FAST ? 2 : impl == ZGPU_LZ_FASTWIN ? 3 : 0);
            else launch_lz_parallel(g, cfg, e->par_ws, e->tokens, e->meta, st, e);
        }
        {
            StageTimer t(e, st, ZGPU_STAGE_HUFFMAN);
            launch_huffman(g, e->tokens, e->meta, slots, st, cfg.strategy == kFixed);
        }
        {
            StageTimer t(e, st, ZGPU_STAGE_STITCH);
            if (!adler_done) launch_adler(g, e->meta, st); // (the sort of the default path has computed it on the way)
            if (gz || (p->flags & ZGPU_F_CRC32)) launch_crc(g, e->meta, st);
            launch_scan(e->meta, nb, c0, e->offsets, e->run, body_cap, st, gz || (p->flags & ZGPU_F_CRC32));
            launch_stitch(slots, e->meta, e->offsets, c0, nb, d_out, body_cap, g.slot_stride, st);
        }
        if (home) {
            hipError_t he = hipMemcpyAsync(&e->pin_tot[nbatch], e->run, sizeof(uint64_t), hipMemcpyDeviceToHost, st); // RunState::out_total
            if (he == hipSuccess) he = hipEventRecord(e->done_ev[nbatch], st);
            if (he != hipSuccess) { issued.store((size_t)-1); homer.join(); return zgpu::fail_hip(e, he, "batch hand-over", __FILE__, __LINE__); }
            issued.store(nbatch + 1, std::memory_order_release);
        }
        {
            const hipError_t he = hipGetLastError();
            if (he != hipSuccess) { if (home) { issued.store((size_t)-1); homer.join(); } return zgpu::fail_hip(e, he, "kernel launch", __FILE__, __LINE__); }
        }
    }
    if (home) { homer.join(); if (homer_rc.load()) return fail(e, ZGPU_ERRNO, "copy of finished batches to the host failed"); if (h_copied) *h_copied = homed; }
    ZGPU_HIP_CHECK(hipMemcpyAsync(&rs, e->run, sizeof rs, hipMemcpyDeviceToHost, st));
    if (check_sort) ZGPU_HIP_CHECK(hipMemcpyAsync(&sort_fault, lz_sorted_fault_word(e->par_ws), 4, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    collect_spans(e);
    if (sort_fault) { // the LDS did not serve an atomic's lanes in lane order: redo the call with the sort that does not rely on it
        e->exact_sort = 1;
        return deflate_device(e, d_in, in_bytes, d_seg, nseg, p, d_out, out_cap, d_chunk_offsets, res, st, skip0, h_src, h_dst, h_cap, h_copied);
    }
    if (rs.overflow || (tail_bytes && out_cap < rs.out_total + tail_bytes)) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    const uint32_t adler = rs.adler_a | (rs.adler_b << 16);
    if (wrap) {
        uint8_t tr[4] = {(uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler};
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_out + rs.out_total, tr, 4, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        rs.out_total += 4;
    }
    if (gz) { // CRC-32 and the input length mod 2^32, both little-endian (qcsrc/deflate.c:833-843)
        const uint32_t isz = (uint32_t)in_bytes;
        uint8_t tr[8] = {(uint8_t)rs.crc, (uint8_t)(rs.crc >> 8), (uint8_t)(rs.crc >> 16), (uint8_t)(rs.crc >> 24),
                         (uint8_t)isz, (uint8_t)(isz >> 8), (uint8_t)(isz >> 16), (uint8_t)(isz >> 24)};
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_out + rs.out_total, tr, 8, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        rs.out_total += 8;
    }
    if (d_chunk_offsets) {
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_chunk_offsets, e->offsets, (nchunks + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    }
    res->out_bytes = rs.out_total; res->nchunks = nchunks; res->adler32 = adler; res->data_type = rs.data_type; res->ntokens = rs.ntokens;
    res->crc32 = (gz || (p->flags & ZGPU_F_CRC32)) ? rs.crc : 0;
    return ZGPU_OK;
}

ChunkMeta *engine_meta(zgpu_engine *e, uint32_t batch);
uint64_t *engine_offsets_scratch(zgpu_engine *e, uint64_t n);

// ---------------------------------------------------------------------------------------------------------------------------------------------
// One FEED of a continuous stream (zgpu_cont.hip has the plan): d_buf holds the history the parse can still reach followed by the bytes that have
// not been parsed yet, cs says where the stream stands (absolute stream positions) and receives where it stands afterwards, d_carry_in holds the
// tokens of the block that is filling (device memory, cs->carry_ntok words, not e->ct_carry); the ones the feed leaves behind are in e->ct_carry.  The output starts at bit cs->bit_count of d_out's byte 0 ... in general at bit
// `prefix_bits` of d_out: the caller has put the bytes in front (a wrapper's header) and the bits of the unfinished byte there, zero-filled to a whole word.
constexpr uint32_t kContCarry = 16384;
static int ensure_cont_ws(zgpu_engine *e, uint32_t batch_tiles, uint64_t feed_tiles)
{
    int rc;
    if (!e->ct_st) {
        if ((rc = dev_alloc(e, &e->ct_st, 1))) return rc;
        if ((rc = dev_alloc(e, &e->ct_carry, kContCarry))) return rc;
        if ((rc = dev_alloc(e, &e->ct_carry_in, kContCarry))) return rc;
    }
    if (feed_tiles + 2 > e->ct_feed_tiles) {
        hipFree(e->ct_entry); e->ct_entry = nullptr; e->ct_feed_tiles = 0;
        if ((rc = dev_alloc(e, &e->ct_entry, feed_tiles + 2 + 1024))) return rc;
        e->ct_feed_tiles = feed_tiles + 2 + 1024;
    }
    if (batch_tiles > e->ct_tiles) {
        hipFree(e->ct_exits); hipFree(e->ct_comp); hipFree(e->ct_gentry); hipFree(e->ct_tokoff); hipFree(e->ct_T); hipFree(e->ct_blk); hipFree(e->ct_pos); hipFree(e->ct_slots);
        e->ct_exits = e->ct_comp = e->ct_gentry = nullptr; e->ct_tokoff = e->ct_T = nullptr; e->ct_blk = nullptr; e->ct_pos = nullptr; e->ct_slots = nullptr; e->ct_tiles = 0;
        const size_t ntok_cap = (size_t)kContCarry + kChunkMax + (size_t)batch_tiles * (kTileStride + kTileSlack); // tile 0 of a feed parses up to 65024 positions, the others 32512, + the last game's overhang
        const uint32_t nblk_cap = (uint32_t)(ntok_cap / kBlockTokens + 2);
        const uint32_t ngroups = chain_groups(batch_tiles);
        if ((rc = dev_alloc(e, &e->ct_exits, (size_t)batch_tiles * kTileExitStride))) return rc;
        if ((rc = dev_alloc(e, &e->ct_comp, (size_t)ngroups * kTileExitStride))) return rc;
        if ((rc = dev_alloc(e, &e->ct_gentry, (size_t)ngroups + 1))) return rc;
        if ((rc = dev_alloc(e, &e->ct_tokoff, (size_t)batch_tiles + 1))) return rc;
        if ((rc = dev_alloc(e, &e->ct_T, ntok_cap))) return rc;
        if ((rc = dev_alloc(e, &e->ct_blk, (size_t)nblk_cap))) return rc;
        if ((rc = dev_alloc(e, &e->ct_pos, (size_t)nblk_cap + 1))) return rc;
        if ((rc = dev_alloc(e, &e->ct_slots, (size_t)nblk_cap * kSlotStride))) return rc;
        e->ct_tiles = batch_tiles; e->ct_nblk = nblk_cap;
    }
    return ZGPU_OK;
}

static int ensure_fast_ws(zgpu_engine *e, uint32_t batch_tiles)
{
    int rc;
    if (!e->cf_prev) {
        if ((rc = dev_alloc(e, &e->cf_prev, kInsWords + 8))) return rc;
        if ((rc = dev_alloc(e, &e->cf_prev2, kInsWords + 8))) return rc;
        if ((rc = dev_alloc(e, &e->cf_hist, kInsWords + 8))) return rc;
        if ((rc = dev_alloc(e, &e->cf_count, 4))) return rc;
    }
    if (batch_tiles > e->cf_tiles) {
        hipFree(e->cf_exit_a); hipFree(e->cf_exit_b); hipFree(e->cf_ins0); hipFree(e->cf_ins1); hipFree(e->cf_cur); hipFree(e->cf_act_a); hipFree(e->cf_act_b); hipFree(e->cf_changed);
        hipFree(e->cf_list_a); hipFree(e->cf_list_b); e->cf_list_a = e->cf_list_b = nullptr;
        hipFree(e->cf_kept); hipFree(e->cf_used); hipFree(e->cf_entry_used); e->cf_kept = nullptr; e->cf_used = nullptr; e->cf_entry_used = nullptr;
        e->cf_exit_a = e->cf_exit_b = nullptr; e->cf_ins0 = e->cf_ins1 = nullptr; e->cf_cur = e->cf_act_a = e->cf_act_b = e->cf_changed = nullptr; e->cf_tiles = 0;
        if ((rc = dev_alloc(e, &e->cf_exit_a, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_exit_b, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_ins0, (size_t)batch_tiles * kInsWords))) return rc;
        if ((rc = dev_alloc(e, &e->cf_ins1, (size_t)batch_tiles * kInsWords))) return rc;
        if ((rc = dev_alloc(e, &e->cf_cur, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_act_a, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_act_b, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_changed, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_list_a, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_list_b, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_kept, (size_t)batch_tiles))) return rc;
        if ((rc = dev_alloc(e, &e->cf_used, (size_t)batch_tiles * kInsWords))) return rc;
        if ((rc = dev_alloc(e, &e->cf_entry_used, (size_t)batch_tiles))) return rc;
        e->cf_tiles = batch_tiles;
    }
    return ZGPU_OK;
}
// levels 1-3: the tiles of one batch by rounds (zgpu_lz_fastwin.hip, fastwin_tile_kernel): until no tile's predecessor has changed
static int lz_tiles_fast(zgpu_engine *e, const ChunkGeom &g, const TileGeom &tg, const LevelCfg &cfg, hipStream_t st)
{
    const uint32_t nb = g.nchunks;
    const uint16_t *S; const uint32_t *ir;
    launch_sort_tiles(g, e->par_ws, e->meta, st, e, e->exact_sort, &S, &ir);
    StageTimer t(e, st, ZGPU_STAGE_MATCH);
    launch_fast_init(e->cf_cur, e->cf_act_a, e->cf_exit_a, nb, st);
    uint8_t *act = e->cf_act_a, *act_next = e->cf_act_b;
    uint32_t *list = nullptr, *list_next = e->cf_list_a, ngrid = nb;
    static uint32_t *dbg = nullptr, *dstat = nullptr;
    const bool trace = getenv("ZGPU_FAST_TRACE") != nullptr;
    if (trace && !dbg) hipMalloc(reinterpret_cast<void **>(&dbg), 65536 * 32);
    if (trace && !dstat) hipMalloc(reinterpret_cast<void **>(&dstat), 32);
    static int keep = -1; // ZGPU_FAST_KEEP=0: every active tile is parsed to its end (A/B runs)
    if (keep < 0) { const char *v = getenv("ZGPU_FAST_KEEP"); keep = v ? atoi(v) : 1; }
    auto fill = [&](FastTiles &ft, uint32_t round, const uint32_t *lst) {
        ft.exit_cur = e->cf_exit_a; ft.exit_new = e->cf_exit_b; ft.ins0 = ft.ins0w = e->cf_ins0; ft.ins1 = ft.ins1w = e->cf_ins1; ft.cur = e->cf_cur; ft.active = act; ft.changed = e->cf_changed;
        ft.prev_ins = e->cf_prev; ft.round = round; ft.list = lst; ft.low_out = e->cf_hist; // (cf_hist: free until the feed's hand-over is put together)
        if (keep) { ft.used_ins = e->cf_used; ft.entry_used = e->cf_entry_used; ft.kept = e->cf_kept; }
        ft.dbg = trace && nb <= 65536 ? dbg : nullptr;
        if (trace) { hipMemsetAsync(dstat, 0, 32, st); ft.stat = dstat; }
    };
    // Round 0 in K phases (ZGPU_FAST_RUN=K; 1: all tiles at once, as until round 4).  A tile that starts from nothing (a warm-up over its history) knows little about which
    // of the positions in front of it are in the chains, and most tiles were parsed again four or five times before their neighbours' guesses had settled.  So only every
    // K-th tile guesses; the K - 1 behind it are parsed one phase after the other, each from where the tile in front ended and with that tile's bits -- by the third or
    // fourth of a run those are the stream's.  (1 + 1 / K) parses per tile instead of 2, and the rounds that follow start from a far better state; the price is K
    // launches with 1 / K of the tiles each, so K grows with the batch.
    uint32_t K = env_u32("ZGPU_FAST_RUN", 0);
    if (K == 0) { K = nb / 256; if (K < 1) K = 1; if (K > 16) K = 16; } // (measured: 64 MiB 146 -> 108 ms with 8, 256 MiB 309 -> 170 with 16, 1 GiB 2029 -> 427; no gain below 16 MiB, where every launch is one tile's latency)
    // ... and no phase larger than the chip holds at once: eleven one-wave workgroups of 13.3 KiB LDS a CU are 2 816 tiles, a phase of 4 128 (a batch of 66 052 in 16) runs as two
    // waves of workgroups, the second a third full -- 4 GiB at level 1: 16 phases 672 ms, 22 (3 003 tiles each) 673, 24 (2 752) 577, 32 654, 44 711
    if (env_u32("ZGPU_FAST_RUN", 0) == 0) { const uint32_t kw = (nb + 2751) / 2752; if (kw > K) K = kw; }
    if (K > 64) K = 64;
    if (K > 1) {
        std::vector<uint32_t> all(nb);
        uint32_t at = 0;
        std::vector<uint32_t> start(K + 1, 0);
        for (uint32_t p = 0; p < K; p++) { start[p] = at; for (uint32_t c = p; c < nb; c += K) all[at++] = c; }
        start[K] = at;
        ZGPU_HIP_CHECK(hipMemcpyAsync(e->cf_list_b, all.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st)); // (the vector goes out of scope)
        for (uint32_t p = 0; p < K; p++) {
            const uint32_t cnt = start[p + 1] - start[p];
            if (!cnt) continue;
            FastTiles ft{};
            fill(ft, 0, e->cf_list_b + start[p]);
            ft.warm_mode = p == 0;
            launch_lz_fastwin_tiles(g, tg, ft, cfg, S, ir, e->tokens, e->meta, cnt, st);
            launch_fast_flip_list(e->cf_cur, e->cf_exit_a, e->cf_exit_b, e->cf_list_b + start[p], cnt, st);
            e->cf_tile_parses += cnt;
        }
        e->cf_rounds++;
        // what the rounds start with: the run heads behind the first -- the tile in front of each was parsed after it
        std::vector<uint32_t> heads;
        std::vector<uint8_t> ab(nb, 0);
        for (uint32_t c = K; c < nb; c += K) { heads.push_back(c); ab[c] = 1; }
        if (heads.empty()) { launch_fast_finish(e->cf_cur, e->cf_exit_a, e->cf_ins0, e->cf_ins1, nb, tg.entry + g.chunk0 + nb, e->cf_prev, e->cf_prev2, g.chunk0 == 0 ? e->cf_hist : nullptr, st); return ZGPU_OK; }
        ZGPU_HIP_CHECK(hipMemcpyAsync(e->cf_list_a, heads.data(), heads.size() * 4, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipMemcpyAsync(act, ab.data(), nb, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        if (trace) fprintf(stderr, "fast tiles: round 0 in %u phases, %zu run heads to parse again\n", K, heads.size());
        list = e->cf_list_a; list_next = e->cf_list_b; ngrid = (uint32_t)heads.size();
        e->cf_tile_parses += ngrid;
    }
    for (uint32_t round = K > 1 ? 1 : 0;; round++) {
        FastTiles ft{};
        fill(ft, round, list);
        ft.warm_mode = round == 0;
        if (ft.dbg) hipMemsetAsync(dbg, 0xff, (size_t)nb * 32, st);
        launch_lz_fastwin_tiles(g, tg, ft, cfg, S, ir, e->tokens, e->meta, ngrid, st);
        if (ft.dbg) {
            std::vector<uint32_t> h((size_t)nb * 8);
            hipMemcpyAsync(h.data(), dbg, (size_t)nb * 32, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
            uint32_t shown = 0;
            for (uint32_t c = 0; c < nb && shown < 6; c++) if (h[c * 8] != 0xffffffffu && ++shown)
                fprintf(stderr, "   tile %u: %u words differ (%u .. %u), exit %u (was %u), entry %u, %u tokens\n", c, h[c * 8 + 1], h[c * 8 + 2], h[c * 8 + 3], h[c * 8 + 4], h[c * 8 + 5], h[c * 8 + 6], h[c * 8 + 7]);
        }
        launch_fast_flip(e->cf_cur, act, act_next, e->cf_changed, e->cf_exit_a, e->cf_exit_b, nb, round, e->cf_count, list_next, ft.kept, st);
        uint32_t active = 0;
        ZGPU_HIP_CHECK(hipMemcpyAsync(&active, e->cf_count, 4, hipMemcpyDeviceToHost, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        e->cf_rounds++; e->cf_tile_parses += round == 0 ? nb : 0;
        static int force = -1; // ZGPU_FAST_FORCE_ROUNDS=n (debug): every tile is parsed again in each of the first n rounds
        if (force < 0) { const char *v = getenv("ZGPU_FAST_FORCE_ROUNDS"); force = v ? atoi(v) : 0; }
        if ((int)round < force && nb > 1) { // (all of them again: the list is 1 .. nb - 1)
            std::vector<uint32_t> all(nb - 1);
            for (uint32_t i = 1; i < nb; i++) all[i - 1] = i;
            ZGPU_HIP_CHECK(hipMemsetAsync(act_next + 1, 1, nb - 1, st));
            ZGPU_HIP_CHECK(hipMemcpyAsync(list_next, all.data(), (size_t)(nb - 1) * 4, hipMemcpyHostToDevice, st));
            ZGPU_HIP_CHECK(hipStreamSynchronize(st));
            active = nb - 1;
        }
        if (trace) {
            static double t_last = 0; timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); const double t_now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
            fprintf(stderr, "[%.2f ms since the last round's end] ", t_now - t_last); t_last = t_now;
            uint32_t hs[8] = {};
            hipMemcpy(hs, dstat, 32, hipMemcpyDeviceToHost);
            fprintf(stderr, "fast tiles: round %u, %u of %u tiles to parse again (this round: %u entered where they had, %u stopped early, %u met a different token, %u had changes within reach to the end; first different token after %u of %u windows)\n", round, active, nb, hs[0], hs[1], hs[2], hs[3], hs[4], hs[5]);
        }
        if (active == 0) break;
        e->cf_tile_parses += active;
        uint8_t *x = act; act = act_next; act_next = x;
        list = list_next; list_next = list == e->cf_list_a ? e->cf_list_b : e->cf_list_a; ngrid = active;
    }
    launch_fast_finish(e->cf_cur, e->cf_exit_a, e->cf_ins0, e->cf_ins1, nb, tg.entry + g.chunk0 + nb, e->cf_prev, e->cf_prev2, g.chunk0 == 0 ? e->cf_hist : nullptr, st);
    return ZGPU_OK;
}

struct ContFeed { // one call of deflate_cont
    const uint8_t *d_buf; uint64_t buf_bytes; // history + unparsed bytes
    uint64_t check_from;                      // Adler-32 / CRC-32 of d_buf[check_from ..): the bytes this feed brought (buf_bytes: none)
    int mode;                                 // ZGPU_CONT_*
    const uint64_t *h_excl; uint32_t nexcl;   // stream positions that are not in the hash chains (in front of earlier flushes), ascending
    uint8_t *d_out; uint64_t out_cap; uint64_t prefix_bits;
    bool want_crc;
    uint32_t *h_hist; // levels 1-3: "in the hash chains", one bit per position of the history (bit j: the position 32512, or all there are, in front of cs->entry, + j); in and out; kInsWords words
};
static int deflate_cont(zgpu_engine *e, const ContFeed &f, const LevelCfg &cfg, zgpu_cont_state *cs, const uint32_t *d_carry_in, zgpu_deflate_result *res, hipStream_t st)
{
    // levels 1-3 with Z_HUFFMAN_ONLY: no match is ever looked for, no chain ever asked: the walkers' path (every byte a literal) serves it
    const bool fast_lz = !cfg.slow && cfg.strategy != kHuffmanOnly;
    if (fast_lz && (!lz_fastwin_serves(cfg) || cfg.strategy == kRle || !f.h_hist))
        return fail(e, ZGPU_STREAM_ERROR, "continuous stream at levels 1..3: the levels' own parameters, not Z_RLE");
    if (cs->entry < cs->abs0 || cs->entry > cs->abs0 + f.buf_bytes || cs->carry_ntok >= kBlockTokens || cs->bit_count > 7) return fail(e, ZGPU_STREAM_ERROR, "continuous stream: state out of range");
    if ((reinterpret_cast<uintptr_t>(f.d_out) & 3) != 0) return fail(e, ZGPU_STREAM_ERROR, "continuous stream: the output must be 4-byte aligned");
    const uint64_t e0 = cs->entry - cs->abs0, w0 = e0 > kTileStride ? e0 - kTileStride : 0;
    const bool ends = f.mode != ZGPU_CONT_MORE;
    const uint64_t end = ends ? f.buf_bytes : (f.buf_bytes > kTileSlack ? f.buf_bytes - kTileSlack : 0);
    const uint64_t ntiles = end > e0 ? (end <= w0 + kTileH1 ? 1 : (end - w0 - kTileH1 + kTileStride - 1) / kTileStride + 1) : 0;
    if (!ends && ntiles == 0) { res->out_bytes = 0; res->nchunks = 0; res->ntokens = 0; res->adler32 = 1; res->crc32 = 0; res->data_type = cs->data_type; if (f.check_from < f.buf_bytes) return fail(e, ZGPU_STREAM_ERROR, "continuous stream: a feed that parses nothing brings no bytes"); return ZGPU_OK; }
    uint32_t batch_max = env_u32("ZGPU_CONT_BATCH_TILES", fast_lz ? 65536 : 32768); // (levels 1-3: every batch ends with a tail of rounds that are one tile's latency each, 4 GiB 1395 -> 1149 ms with twice the batch)
    uint32_t batch_fit = ~0u; // what the device's memory admits
    {
        size_t free_b = 0, total_b = 0;
        const size_t per_tile = lz_sorted_workspace_bytes(1) + (size_t)kChunkMax * 4 + (size_t)(kTileStride + kTileSlack) * 4 + 3 * (size_t)kSlotStride + kSlotStride;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // what this engine holds already and would use again, piece by piece (the chunked path's buffers serve the tiles too): counted as free, or the batch would
            // grow from call to call as more of the same memory is held -- and a later call of a series would stop to allocate (seen in bench.py: 44 035, 64 837, 66 052 tiles)
            const size_t held = (size_t)e->batch_cap * ((size_t)kChunkMax * 4 + kSlotStride) + (size_t)e->par_cap * lz_sorted_workspace_bytes(1) + (size_t)e->ct_tiles * ((size_t)(kTileStride + kTileSlack) * 4 + 3 * (size_t)kSlotStride);
            size_t fit = (free_b + held) / 10 * 6 / per_tile;
            if (fit < 64) fit = 64;
            if (fit < batch_max) batch_max = (uint32_t)fit;
            batch_fit = fit < 0xffffffffull ? (uint32_t)fit : ~0u;
        }
    }
    // levels 1-3: no short batch at the end -- its tail of rounds is as long as a full batch's (2 GiB are 66 052 tiles: one batch, not 65 536 + 516; 4 GiB two of 66 052, not
    // two and 1 032 tiles that take 70 ms).  A remainder below an eighth of a batch is spread over the others if memory admits it, and the batches are of one size.
    if (fast_lz && ntiles > batch_max && !getenv("ZGPU_CONT_BATCH_TILES")) {
        uint64_t nbat = (ntiles + batch_max - 1) / batch_max;
        const uint64_t rem = ntiles % batch_max;
        if (rem && rem < batch_max / 8) nbat--;
        const uint64_t even = (ntiles + nbat - 1) / nbat;
        if (even <= batch_fit) batch_max = (uint32_t)even;
    }
    // levels 4-9, more than one batch: two sets of per-batch buffers, half a batch each -- batch k's tokens, blocks and bits are made on a second stream
    // while batch k + 1 is sorted and walked on the first (ZGPU_CONT_PIPE=0: one after the other, for A/B runs)
    int pipe_env = 0; // 1: feeds of 2048 tiles and more, 2: whatever the size of the feed (tests).  Off by default: measured at 4 GiB, level 6, 242 ms against 238 --
                      // two walker workgroups fill a CU's LDS, so the other stream's kernels run beside the sort only, and both are bound by the same memory system
    { const char *v = getenv("ZGPU_CONT_PIPE"); if (v && *v) pipe_env = atoi(v); }
    const bool pipe = !fast_lz && (pipe_env == 2 ? ntiles >= 2 : pipe_env == 1 && ntiles >= 2048);
    if (pipe) { // at least four batches, so that all but the first one's walkers and the last one's blocks have something running beside them
        const uint64_t want = (ntiles + 3) / 4;
        if (batch_max >= 2048) batch_max /= 2;
        if (want >= 1024 && want < batch_max) batch_max = (uint32_t)want;
    }
    const uint32_t batch = (uint32_t)(ntiles < batch_max ? (ntiles ? ntiles : 1) : batch_max);
    const bool size_trace = getenv("ZGPU_FAST_TRACE") != nullptr;
    timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    if (size_trace) fprintf(stderr, "continuous stream: %llu tiles, memory admits %u, batches of %u (held: %u tiles, %u chunks of tokens, %u of buckets)\n", (unsigned long long)ntiles, batch_fit, batch, e->ct_tiles, e->batch_cap, e->par_cap);
    int rc = ensure_deflate_ws(e, pipe ? 2 * batch + 1 : batch, false, 1);
    if (rc) return rc;
    if ((rc = ensure_cont_ws(e, pipe ? 2 * batch : batch, ntiles))) return rc;
    if (size_trace) { timespec ts1; clock_gettime(CLOCK_MONOTONIC, &ts1); fprintf(stderr, "continuous stream: workspaces in %.1f ms\n", (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6); }
    if (pipe && !e->ct_stream) {
        ZGPU_HIP_CHECK(hipStreamCreateWithFlags(&e->ct_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) { ZGPU_HIP_CHECK(hipEventCreateWithFlags(&e->ct_ev_a[i], hipEventDisableTiming)); ZGPU_HIP_CHECK(hipEventCreateWithFlags(&e->ct_ev_b[i], hipEventDisableTiming)); }
    }
    if (fast_lz) {
        if ((rc = ensure_fast_ws(e, batch))) return rc;
        ZGPU_HIP_CHECK(hipMemcpyAsync(e->cf_prev, f.h_hist, (size_t)kInsWords * 4, hipMemcpyHostToDevice, st)); // (pageable: the copy has read it when the call returns)
    }
    const uint64_t seg_end = ends ? cs->abs0 + f.buf_bytes : ~0ull, sp = ends ? cont_special_pos(seg_end) : ~0ull;
    uint32_t nexcl_dev = 0;
    if (f.nexcl && !fast_lz) { // (levels 1-3 know which positions are in the chains bit by bit: hist bits)
        if (f.nexcl > e->ct_excl_cap) { hipFree(e->ct_excl); e->ct_excl = nullptr; e->ct_excl_cap = 0; if ((rc = dev_alloc(e, &e->ct_excl, (size_t)f.nexcl + 64))) return rc; e->ct_excl_cap = f.nexcl + 64; }
        std::vector<uint64_t> off(f.nexcl);
        uint32_t k = 0;
        for (uint32_t i = 0; i < f.nexcl; i++) if (f.h_excl[i] >= cs->abs0 && f.h_excl[i] - cs->abs0 < f.buf_bytes) off[k++] = f.h_excl[i] - cs->abs0; // (buffer offsets)
        ZGPU_HIP_CHECK(hipMemcpyAsync(e->ct_excl, off.data(), (size_t)k * 8, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        nexcl_dev = k;
    }
    ContState hs{};
    hs.out_bits = f.prefix_bits; hs.block_start = cs->block_start; hs.carry_n = cs->carry_ntok; hs.data_type = cs->data_type; hs.last_eob = cs->last_eob;
    hs.first_block = cs->first_block; hs.e_next = cs->entry;
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->ct_st, &hs, sizeof hs, hipMemcpyHostToDevice, st));
    { const uint16_t z = 0; ZGPU_HIP_CHECK(hipMemcpyAsync(e->ct_entry, &z, 2, hipMemcpyHostToDevice, st)); }
    const bool check_sort = !e->exact_sort;
    uint32_t sort_fault = 0;
    const size_t ws_half = (lz_sorted_workspace_bytes(batch) + 255) & ~(size_t)255;
    uint8_t *const ws_set[2] = {static_cast<uint8_t *>(e->par_ws), static_cast<uint8_t *>(e->par_ws) + (pipe ? ws_half : 0)};
    if (check_sort) { ZGPU_HIP_CHECK(hipMemsetAsync(lz_sorted_fault_word(ws_set[0]), 0, 4, st)); if (pipe) ZGPU_HIP_CHECK(hipMemsetAsync(lz_sorted_fault_word(ws_set[1]), 0, 4, st)); }

    ChunkGeom g{};
    g.in = f.d_buf; g.in_bytes = f.buf_bytes; g.chunk_size = kChunkMax; g.final_chunk = ~0ull; g.block_tokens = kBlockTokens; g.slot_stride = kSlotStride;
    g.tile_w0 = w0; g.tile_stride = kTileStride; g.excl = nexcl_dev ? e->ct_excl : nullptr; g.nexcl = nexcl_dev;
    TileGeom tg{};
    tg.abs0 = cs->abs0; tg.e0 = e0; tg.end = end; tg.nil_pos = (sp != ~0ull && sp >= cs->abs0 && sp < seg_end) ? sp - cs->abs0 : ~0ull; tg.abs0_nil = 1;
    tg.exits = e->ct_exits; tg.entry = e->ct_entry;
    const uint64_t out_cap4 = f.out_cap & ~3ull;
    hipStream_t sb = pipe ? e->ct_stream : st; // where a batch's blocks are made
    if (pipe) { ZGPU_HIP_CHECK(hipEventRecord(e->ct_ev_a[0], st)); ZGPU_HIP_CHECK(hipStreamWaitEvent(sb, e->ct_ev_a[0], 0)); } // (the second stream starts behind what the caller has queued: the input, the state)
    uint64_t kbatch = 0;
    for (uint64_t t0 = 0; t0 == 0 || t0 < ntiles; t0 += batch, kbatch++) { // (a feed without tiles still closes the block that is filling)
        const uint32_t nb = (uint32_t)(ntiles - t0 < batch ? ntiles - t0 : batch);
        const bool last_batch = t0 + nb >= ntiles;
        const int set = pipe ? (int)(kbatch & 1) : 0;
        uint32_t *const tokens = e->tokens + (size_t)set * batch * kChunkMax;
        ChunkMeta *const tmeta = e->meta + (size_t)set * batch;
        tg.exits = e->ct_exits + (size_t)set * batch * kTileExitStride;
        g.chunk0 = t0; g.nchunks = nb;
        if (pipe && kbatch >= 2) ZGPU_HIP_CHECK(hipStreamWaitEvent(st, e->ct_ev_b[set], 0)); // this set's buffers: the batch before last is done with them
        if (nb && !fast_lz) launch_lz_tiles(g, tg, cfg, ws_set[set], tokens, tmeta, e->ct_comp, e->ct_gentry, st, e, e->exact_sort);
        else if (nb && (rc = lz_tiles_fast(e, g, tg, cfg, st))) return rc;
        if (pipe) { ZGPU_HIP_CHECK(hipEventRecord(e->ct_ev_a[set], st)); ZGPU_HIP_CHECK(hipStreamWaitEvent(sb, e->ct_ev_a[set], 0)); }
        if (nb && !fast_lz) launch_lz_tiles_parse(g, tg, cfg, ws_set[set], tokens, tmeta, sb, e);
        const uint64_t seg_here = (ends && last_batch) ? seg_end : ~0ull;
        {
            StageTimer t(e, sb, ZGPU_STAGE_PARSE);
            launch_cont_tokens(g, tg, tokens, tmeta, e->ct_st, e->ct_tokoff, t0 == 0 ? d_carry_in : e->ct_carry, e->ct_T, e->ct_blk, seg_here, f.mode == ZGPU_CONT_FINISH && seg_here != ~0ull, seg_here != ~0ull ? sp : ~0ull, e->ct_nblk, cfg.slow != 0, sb);
        }
        {
            StageTimer t(e, sb, ZGPU_STAGE_HUFFMAN);
            launch_huffman_cont(g, e->ct_T, e->ct_nblk, e->ct_blk, e->ct_st, e->ct_slots, sb, cfg.strategy == kFixed);
        }
        {
            StageTimer t(e, sb, ZGPU_STAGE_STITCH);
            launch_cont_stitch(e->ct_blk, e->ct_st, e->ct_pos, e->ct_slots, kSlotStride, f.d_buf, cs->abs0, f.d_out, out_cap4, e->ct_nblk, e->ct_T, e->ct_carry, seg_here, sb);
        }
        if (pipe) ZGPU_HIP_CHECK(hipEventRecord(e->ct_ev_b[set], sb));
        const hipError_t he = hipGetLastError();
        if (he != hipSuccess) return zgpu::fail_hip(e, he, "kernel launch", __FILE__, __LINE__);
        if (ntiles == 0) break;
    }
    if (pipe) { ZGPU_HIP_CHECK(hipEventRecord(e->ct_ev_b[0], sb)); ZGPU_HIP_CHECK(hipStreamWaitEvent(st, e->ct_ev_b[0], 0)); } // everything the second stream has made lies in front of what follows
    // checksums of the bytes this feed brought
    uint32_t adler = 1, crc = 0;
    if (f.check_from < f.buf_bytes) {
        const uint64_t nbytes = f.buf_bytes - f.check_from, nck = (nbytes + kChunkMax - 1) / kChunkMax;
        const uint32_t cb = (uint32_t)(nck < 65536 ? nck : 65536);
        if (cb > e->ct_ck_cap) { hipFree(e->ct_ckmeta); e->ct_ckmeta = nullptr; e->ct_ck_cap = 0; if ((rc = dev_alloc(e, &e->ct_ckmeta, (size_t)cb))) return rc; e->ct_ck_cap = cb; }
        uint64_t *offs = engine_offsets_scratch(e, nck + 1);
        if (!offs) return fail(e, ZGPU_MEM_ERROR, "checksum scratch");
        StageTimer t(e, st, ZGPU_STAGE_STITCH);
        RunStateHost rs{}; rs.adler_a = 1;
        ZGPU_HIP_CHECK(hipMemcpyAsync(e->run, &rs, sizeof rs, hipMemcpyHostToDevice, st));
        for (uint64_t c0 = 0; c0 < nck; c0 += cb) {
            const uint32_t nbk = (uint32_t)(nck - c0 < cb ? nck - c0 : cb);
            ZGPU_HIP_CHECK(hipMemsetAsync(e->ct_ckmeta, 0, (size_t)nbk * sizeof(ChunkMeta), st));
            ChunkGeom g2{}; g2.in = f.d_buf + f.check_from; g2.in_bytes = nbytes; g2.chunk_size = kChunkMax; g2.chunk0 = c0; g2.nchunks = nbk; g2.final_chunk = ~0ull;
            launch_adler(g2, e->ct_ckmeta, st);
            if (f.want_crc) launch_crc(g2, e->ct_ckmeta, st);
            launch_scan(e->ct_ckmeta, nbk, c0, offs, e->run, ~0ull, st, f.want_crc);
        }
        ZGPU_HIP_CHECK(hipMemcpyAsync(&rs, e->run, sizeof rs, hipMemcpyDeviceToHost, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        adler = rs.adler_a | (rs.adler_b << 16); crc = rs.crc;
    }
    uint16_t k_next = 0;
    ZGPU_HIP_CHECK(hipMemcpyAsync(&hs, e->ct_st, sizeof hs, hipMemcpyDeviceToHost, st));
    if (ntiles) ZGPU_HIP_CHECK(hipMemcpyAsync(&k_next, e->ct_entry + ntiles, 2, hipMemcpyDeviceToHost, st));
    uint32_t sort_fault2 = 0;
    if (check_sort) ZGPU_HIP_CHECK(hipMemcpyAsync(&sort_fault, lz_sorted_fault_word(ws_set[0]), 4, hipMemcpyDeviceToHost, st));
    if (check_sort && pipe) ZGPU_HIP_CHECK(hipMemcpyAsync(&sort_fault2, lz_sorted_fault_word(ws_set[1]), 4, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    collect_spans(e);
    sort_fault |= sort_fault2;
    if (sort_fault) { e->exact_sort = 1; return deflate_cont(e, f, cfg, cs, d_carry_in, res, st); } // (nothing of cs or of the incoming carry has been touched yet)
    if (hs.overflow) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    // where the stream stands now
    if (ntiles) {
        const uint64_t wb_last = w0 + (ntiles - 1) * kTileStride, lim = end - wb_last, h1_last = lim < kTileH1 ? lim : kTileH1;
        cs->entry = cs->abs0 + wb_last + h1_last + k_next;
    }
    if (fast_lz) { // the history's bits for the next feed: the 32512 positions (or all there are) in front of where the parse stands
        uint32_t x0 = 0, count = 0;
        if (ntiles) {
            const uint64_t wb_last = w0 + (ntiles - 1) * kTileStride, e_buf = cs->entry - cs->abs0, nw0 = e_buf > kTileStride && e_buf - kTileStride > w0 ? e_buf - kTileStride : w0;
            x0 = (uint32_t)(nw0 - wb_last); count = (uint32_t)(e_buf - nw0);
            launch_fast_hist(e->cf_prev2, e->cf_prev, x0, count, e->cf_hist, st);
            ZGPU_HIP_CHECK(hipMemcpyAsync(f.h_hist, e->cf_hist, (size_t)kInsWords * 4, hipMemcpyDeviceToHost, st));
            ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        }
    }
    if (ends) cs->entry = seg_end;
    cs->block_start = hs.block_start; cs->carry_ntok = hs.carry_n; cs->data_type = hs.data_type; cs->first_block = hs.first_block; cs->last_eob = hs.last_eob;
    uint64_t out_bytes;
    if (f.mode == ZGPU_CONT_FINISH) { out_bytes = (hs.out_bits + 7) >> 3; cs->bit_count = 0; cs->bit_value = 0; } // bi_windup
    else {
        out_bytes = hs.out_bits >> 3; cs->bit_count = (uint32_t)(hs.out_bits & 7); cs->bit_value = 0;
        if (cs->bit_count) {
            uint8_t last = 0;
            ZGPU_HIP_CHECK(hipMemcpyAsync(&last, f.d_out + out_bytes, 1, hipMemcpyDeviceToHost, st));
            ZGPU_HIP_CHECK(hipStreamSynchronize(st));
            cs->bit_value = last & ((1u << cs->bit_count) - 1u);
        }
    }
    res->out_bytes = out_bytes; res->nchunks = ntiles; res->adler32 = adler; res->crc32 = crc; res->data_type = hs.data_type; res->ntokens = hs.ntokens;
    return ZGPU_OK;
}

static int ensure_stage(zgpu_engine *e, uint64_t in_bytes, uint64_t out_bytes)
{
    if (in_bytes > e->stage_in_cap) {
        hipFree(e->stage_in); e->stage_in = nullptr; e->stage_in_cap = 0;
        uint64_t cap = in_bytes + (in_bytes >> 3) + 4096;
        ZGPU_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&e->stage_in), cap)); e->stage_in_cap = cap;
    }
    if (out_bytes > e->stage_out_cap) {
        hipFree(e->stage_out); e->stage_out = nullptr; e->stage_out_cap = 0;
        uint64_t cap = out_bytes + (out_bytes >> 3) + 4096;
        ZGPU_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&e->stage_out), cap)); e->stage_out_cap = cap;
    }
    return ZGPU_OK;
}

} // namespace zgpu

using namespace zgpu;

extern "C" {
#pragma GCC visibility push(default)

int zgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *zgpu_version(void) { return "zamd-gpu 0.1 (gfx950; zlib 1.2.3 bit-exact, 64 KiB independent chunks)"; }

int zgpu_engine_create(int device, zgpu_engine **out)
{
    if (!out) return ZGPU_STREAM_ERROR;
    *out = nullptr;
    int n = zgpu_device_count();
    if (device < 0 || device >= n) return ZGPU_ERRNO;
    zgpu_engine *e = new zgpu_engine();
    e->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&e->run, 256) != hipSuccess) {
        delete e;
        return ZGPU_ERRNO;
    }
    *out = e;
    return ZGPU_OK;
}

void zgpu_engine_destroy(zgpu_engine *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    hipFree(e->hand_list); hipFree(e->tokens); hipFree(e->meta); hipFree(e->slots); hipFree(e->tables); hipFree(e->geo_slots); hipFree(e->geo_tables); hipFree(e->geo_nostore); hipFree(e->par_ws); hipFree(e->offsets); hipFree(e->run);
    hipFree(e->ct_exits); hipFree(e->ct_entry); hipFree(e->ct_comp); hipFree(e->ct_gentry); hipFree(e->ct_tokoff); hipFree(e->ct_T); hipFree(e->ct_carry); hipFree(e->ct_carry_in); hipFree(e->ct_blk);
    hipFree(e->ct_pos); hipFree(e->ct_slots); hipFree(e->ct_st); hipFree(e->ct_ckmeta); hipFree(e->ct_excl);
    hipFree(e->cf_exit_a); hipFree(e->cf_exit_b); hipFree(e->cf_ins0); hipFree(e->cf_ins1); hipFree(e->cf_prev); hipFree(e->cf_prev2); hipFree(e->cf_hist); hipFree(e->cf_count);
    hipFree(e->cf_cur); hipFree(e->cf_act_a); hipFree(e->cf_act_b); hipFree(e->cf_changed); hipFree(e->cf_list_a); hipFree(e->cf_list_b); hipFree(e->cf_kept); hipFree(e->cf_used); hipFree(e->cf_entry_used);
    hipFree(e->stage_in); hipFree(e->stage_out); hipFree(e->inf_status); hipFree(e->inf_meta); hipFree(e->inf_offs); hipFree(e->inf_slots); hipFree(e->inf_dict);
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    for (auto ev : e->copy_ev) hipEventDestroy(ev);
    for (auto ev : e->done_ev) hipEventDestroy(ev);
    if (e->ct_stream) hipStreamDestroy(e->ct_stream);
    for (int i = 0; i < 2; i++) { if (e->ct_ev_a[i]) hipEventDestroy(e->ct_ev_a[i]); if (e->ct_ev_b[i]) hipEventDestroy(e->ct_ev_b[i]); }
    if (e->d2h_stream) hipStreamDestroy(e->d2h_stream);
    if (e->pin_tot) hipHostFree(e->pin_tot);
    hipStreamDestroy(e->copy_stream);
    hipStreamDestroy(e->stream);
    delete e;
}

const char *zgpu_engine_error(const zgpu_engine *e) { return e ? e->err : "no engine"; }
uint64_t zgpu_debug_handed_on(const zgpu_engine *e) { return e ? e->handed_on : 0; } // chunks the lane-per-chunk loop gave to the wave-per-chunk kernel so far (tests, bench)

uint64_t zgpu_deflate_bound(uint64_t in_bytes, uint32_t chunk_size)
{
    if (chunk_size == 0 || chunk_size > kChunkMax) chunk_size = kChunkMax;
    uint64_t nchunks = in_bytes ? (in_bytes + chunk_size - 1) / chunk_size : 1;
    return in_bytes + nchunks * 40 + 16; // <= 6 block headers of 5 bytes + 5-byte marker per chunk, + zlib framing
}

// The same for a stream made with deflateInit2(windowBits, memLevel): blocks of as few as 127 tokens (a header each), and blocks that may not be
// stored because their start has left a small window (deflateBound's arithmetic, deflate.c:513-515, covers those).
uint64_t zgpu_deflate_bound_geometry(uint64_t in_bytes, uint32_t chunk_size, int window_bits, int mem_level)
{
    if (window_bits == 15 && mem_level == 8) return zgpu_deflate_bound(in_bytes, chunk_size);
    if (chunk_size == 0 || chunk_size > kChunkMax) chunk_size = kChunkMax;
    if (mem_level < 1 || mem_level > 9) mem_level = 1;
    const uint64_t nchunks = in_bytes ? (in_bytes + chunk_size - 1) / chunk_size : 1, blocks = chunk_size / ((1u << (mem_level + 6)) - 1u) + 2;
    return in_bytes + ((in_bytes + 7) >> 3) + ((in_bytes + 63) >> 6) + nchunks * (5 * blocks + 64) + 16;
}

int zgpu_deflate_set_geometry(zgpu_engine *e, int window_bits, int mem_level)
{
    if (!e) return ZGPU_STREAM_ERROR;
    if (window_bits < 9 || window_bits > 15 || mem_level < 1 || mem_level > 9) return fail(e, ZGPU_STREAM_ERROR, "windowBits 9..15, memLevel 1..9");
    e->geo_w = window_bits; e->geo_m = mem_level;
    return ZGPU_OK;
}

int zgpu_deflate_set_tuning(zgpu_engine *e, int on, uint32_t good_length, uint32_t max_lazy, uint32_t nice_length, uint32_t max_chain)
{
    if (!e) return ZGPU_STREAM_ERROR;
    e->tuned = on != 0; e->tune[0] = good_length; e->tune[1] = max_lazy; e->tune[2] = nice_length; e->tune[3] = max_chain;
    return ZGPU_OK;
}

// ZGPU_F_CONTINUOUS: the whole input as ONE stream, in one feed that finishes it
static int deflate_cont_oneshot(zgpu_engine *e, const uint8_t *d_in, uint64_t in_bytes, const zgpu_deflate_params *p, uint8_t *d_out, uint64_t out_cap, zgpu_deflate_result *res, hipStream_t st)
{
    if (!e || !p || !res || (!d_in && in_bytes) || !d_out) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    if (p->level < 1 || p->level > 9 || p->strategy < 0 || p->strategy > (int)kFixed) return fail(e, ZGPU_STREAM_ERROR, "level 1..9, strategy 0..4");
    if (p->prime || (p->flags & (ZGPU_F_POS0 | ZGPU_F_POS0_ALL)) || !(p->flags & ZGPU_F_FINAL) || e->geo_w != 15 || e->geo_m != 8) return fail(e, ZGPU_STREAM_ERROR, "continuous stream: FINAL, no prime, the default geometry");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    LevelCfg cfg = level_cfg(p->level);
    cfg.strategy = (uint32_t)p->strategy;
    if (e->tuned) { cfg.good = e->tune[0]; cfg.lazy = e->tune[1]; cfg.nice = e->tune[2]; cfg.chain = (e->tune[3] && e->tune[3] < 0xffffu) ? e->tune[3] : 0xffffu; if (cfg.nice > kMaxMatch) cfg.nice = kMaxMatch; }
    const bool wrap = p->flags & ZGPU_F_ZLIB_WRAP, gz = p->flags & ZGPU_F_GZIP_WRAP;
    if (wrap && gz) return fail(e, ZGPU_STREAM_ERROR, "one wrapper at a time");
    const uint32_t head_bytes = wrap ? 2 : gz ? 10 : 0, tail_bytes = wrap ? 4 : gz ? 8 : 0;
    if (out_cap < head_bytes + tail_bytes + 8) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    uint8_t hdr[12] = {0};
    if (wrap) zlib_header(p->level, p->strategy, hdr);
    if (gz) { const uint8_t h[10] = {31, 139, 8, 0, 0, 0, 0, 0, (uint8_t)(p->level == 9 ? 2 : (p->strategy >= 2 || p->level < 2) ? 4 : 0), 3}; memcpy(hdr, h, 10); }
    ZGPU_HIP_CHECK(hipMemcpyAsync(d_out, hdr, (head_bytes + 4) & ~3u, hipMemcpyHostToDevice, st)); // (zero-filled to a whole word: the first block's bits are ORed in behind the header)
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    zgpu_cont_state cs{}; cs.data_type = 2; cs.first_block = 1; cs.last_eob = 8;
    ContFeed f{}; f.d_buf = d_in; f.buf_bytes = in_bytes; f.check_from = 0; f.mode = ZGPU_CONT_FINISH; f.d_out = d_out; f.out_cap = out_cap - tail_bytes; f.prefix_bits = 8ull * head_bytes;
    f.want_crc = gz || (p->flags & ZGPU_F_CRC32);
    std::vector<uint32_t> hist(kInsWords + 8, 0u); // (a fresh stream: nothing is in the chains)
    f.h_hist = hist.data();
    int rc = ensure_cont_ws(e, 1, 1);
    if (rc) return rc;
    rc = deflate_cont(e, f, cfg, &cs, e->ct_carry_in, res, st);
    if (rc) return rc;
    if (wrap) {
        const uint32_t a = res->adler32;
        uint8_t tr[4] = {(uint8_t)(a >> 24), (uint8_t)(a >> 16), (uint8_t)(a >> 8), (uint8_t)a};
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_out + res->out_bytes, tr, 4, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        res->out_bytes += 4;
    }
    if (gz) {
        const uint32_t c = res->crc32, isz = (uint32_t)in_bytes;
        uint8_t tr[8] = {(uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24), (uint8_t)isz, (uint8_t)(isz >> 8), (uint8_t)(isz >> 16), (uint8_t)(isz >> 24)};
        ZGPU_HIP_CHECK(hipMemcpyAsync(d_out + res->out_bytes, tr, 8, hipMemcpyHostToDevice, st));
        ZGPU_HIP_CHECK(hipStreamSynchronize(st));
        res->out_bytes += 8;
    }
    if (!f.want_crc) res->crc32 = 0;
    return ZGPU_OK;
}

uint64_t zgpu_deflate_cont_bound(uint64_t in_bytes)
{
    // deflateBound's own arithmetic for a stream whose blocks may not be stored (deflate.c:513-515) plus a word per block boundary and the framing:
    // what a feed of in_bytes can write
    return in_bytes + ((in_bytes + 7) >> 3) + ((in_bytes + 63) >> 6) + 5 * (in_bytes / 16383 + 2) + 64;
}

int zgpu_deflate_cont_host(zgpu_engine *e, const void *hist, uint64_t hist_bytes, const void *in, uint64_t in_bytes, uint64_t check_from, const zgpu_deflate_params *p, int mode,
                           zgpu_cont_state *cs, uint32_t *carry_tok, uint32_t *hist_bits, const uint64_t *excl, uint32_t nexcl, void *out, uint64_t out_cap, zgpu_deflate_result *res)
{
    if (!e || !p || !res || !cs || !carry_tok || (!hist && hist_bytes) || (!in && in_bytes) || !out) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    const uint64_t buf_bytes = hist_bytes + in_bytes;
    if (p->level < 1 || p->level > 9 || p->strategy < 0 || p->strategy > (int)kFixed || mode < ZGPU_CONT_MORE || mode > ZGPU_CONT_FINISH) return fail(e, ZGPU_STREAM_ERROR, "level 1..9, strategy 0..4, a ZGPU_CONT_* mode");
    if (e->geo_w != 15 || e->geo_m != 8) return fail(e, ZGPU_STREAM_ERROR, "continuous stream: the default geometry");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    LevelCfg cfg = level_cfg(p->level);
    cfg.strategy = (uint32_t)p->strategy;
    if (e->tuned) { cfg.good = e->tune[0]; cfg.lazy = e->tune[1]; cfg.nice = e->tune[2]; cfg.chain = (e->tune[3] && e->tune[3] < 0xffffu) ? e->tune[3] : 0xffffu; if (cfg.nice > kMaxMatch) cfg.nice = kMaxMatch; }
    const uint64_t bound = zgpu_deflate_cont_bound(buf_bytes) + kContCarry * 4;
    int rc = ensure_stage(e, buf_bytes, bound);
    if (rc) return rc;
    if ((rc = ensure_cont_ws(e, 1, 1))) return rc;
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (hist_bytes) ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in, hist, hist_bytes, hipMemcpyHostToDevice, e->stream));
    if (in_bytes) ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in + hist_bytes, in, in_bytes, hipMemcpyHostToDevice, e->stream));
    if (cs->carry_ntok) ZGPU_HIP_CHECK(hipMemcpyAsync(e->ct_carry_in, carry_tok, (size_t)cs->carry_ntok * 4, hipMemcpyHostToDevice, e->stream));
    const uint32_t first_word = cs->bit_value & ((1u << cs->bit_count) - 1u);
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_out, &first_word, 4, hipMemcpyHostToDevice, e->stream));
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    ContFeed f{}; f.d_buf = e->stage_in; f.buf_bytes = buf_bytes; f.check_from = check_from; f.mode = mode; f.h_excl = excl; f.nexcl = nexcl;
    f.d_out = e->stage_out; f.out_cap = bound; f.prefix_bits = cs->bit_count; f.want_crc = (p->flags & ZGPU_F_CRC32) != 0; f.h_hist = hist_bits;
    rc = deflate_cont(e, f, cfg, cs, e->ct_carry_in, res, e->stream);
    if (rc) return rc;
    if (res->out_bytes > out_cap) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    if (res->out_bytes) ZGPU_HIP_CHECK(hipMemcpyAsync(out, e->stage_out, res->out_bytes, hipMemcpyDeviceToHost, e->stream));
    if (cs->carry_ntok) ZGPU_HIP_CHECK(hipMemcpyAsync(carry_tok, e->ct_carry, (size_t)cs->carry_ntok * 4, hipMemcpyDeviceToHost, e->stream));
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (!f.want_crc) res->crc32 = 0;
    return ZGPU_OK;
}

int zgpu_deflate_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, const zgpu_deflate_params *p, void *d_out, uint64_t out_cap,
                        uint64_t *d_chunk_offsets, zgpu_deflate_result *res, void *hip_stream)
{
    if (!e) return ZGPU_STREAM_ERROR;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->stream;
    if (p && (p->flags & ZGPU_F_CONTINUOUS)) return deflate_cont_oneshot(e, static_cast<const uint8_t *>(d_in), in_bytes, p, static_cast<uint8_t *>(d_out), out_cap, res, st);
    return deflate_device(e, static_cast<const uint8_t *>(d_in), in_bytes, nullptr, 0, p, static_cast<uint8_t *>(d_out), out_cap, d_chunk_offsets, res, st);
}

int zgpu_deflate_dict_chunk_host(zgpu_engine *e, const void *window, uint32_t window_bytes, uint32_t dict_bytes, const zgpu_deflate_params *p,
                                 void *out, uint64_t out_cap, zgpu_deflate_result *res)
{
    if (!e || !p || !res || !window || !out) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    const uint64_t bound = zgpu_deflate_bound_geometry(window_bytes, kChunkMax, e ? e->geo_w : 15, e ? e->geo_m : 8);
    int rc = ensure_stage(e, window_bytes, bound);
    if (rc) return rc;
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in, window, window_bytes, hipMemcpyHostToDevice, e->stream));
    zgpu_deflate_params q = *p;
    q.chunk_size = kChunkMax;
    rc = deflate_device(e, e->stage_in, window_bytes, nullptr, 0, &q, e->stage_out, bound, nullptr, res, e->stream, dict_bytes);
    if (rc) return rc;
    if (res->out_bytes > out_cap) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    ZGPU_HIP_CHECK(hipMemcpyAsync(out, e->stage_out, res->out_bytes, hipMemcpyDeviceToHost, e->stream));
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    return ZGPU_OK;
}

int zgpu_deflate_host(zgpu_engine *e, const void *in, uint64_t in_bytes, const zgpu_deflate_params *p, void *out, uint64_t out_cap,
                      uint64_t *chunk_offsets, zgpu_deflate_result *res)
{
    if (!e || !p || !res || (!in && in_bytes) || !out) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    if (p->flags & ZGPU_F_CONTINUOUS) { // one continuous stream: staged whole, compressed in one feed
        const uint64_t cb = zgpu_deflate_cont_bound(in_bytes) + 32;
        int rc2 = ensure_stage(e, in_bytes, cb);
        if (rc2) return rc2;
        ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
        if (in_bytes) ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in, in, in_bytes, hipMemcpyHostToDevice, e->stream));
        rc2 = deflate_cont_oneshot(e, e->stage_in, in_bytes, p, e->stage_out, cb, res, e->stream);
        if (rc2) return rc2;
        if (res->out_bytes > out_cap) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
        ZGPU_HIP_CHECK(hipMemcpyAsync(out, e->stage_out, res->out_bytes, hipMemcpyDeviceToHost, e->stream));
        ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
        return ZGPU_OK;
    }
    const uint32_t chunk_size = p->chunk_size ? p->chunk_size : kChunkMax;
    const uint64_t bound = zgpu_deflate_bound_geometry(in_bytes, chunk_size, e ? e->geo_w : 15, e ? e->geo_m : 8);
    int rc = ensure_stage(e, in_bytes, bound);
    if (rc) return rc;
    // (the copy stream must not run ahead of the previous call's kernels, which may still read the staging buffer: it starts behind them)
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    const bool overlap = in_bytes > (uint64_t)8192 * chunk_size; // (more than one batch: otherwise there is nothing to overlap the copy with)
    if (!overlap && in_bytes) ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in, in, in_bytes, hipMemcpyHostToDevice, e->stream));
    uint64_t homed = 0; // bytes of the stream that went to `out` while later batches were compressed
    rc = deflate_device(e, e->stage_in, in_bytes, nullptr, 0, p, e->stage_out, bound, nullptr, res, e->stream, 0, overlap ? static_cast<const uint8_t *>(in) : nullptr,
                        overlap ? static_cast<uint8_t *>(out) : nullptr, out_cap, &homed);
    if (rc) return rc;
    if (res->out_bytes > out_cap) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    if (res->out_bytes > homed) ZGPU_HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(out) + homed, e->stage_out + homed, res->out_bytes - homed, hipMemcpyDeviceToHost, e->stream));
    if (chunk_offsets) ZGPU_HIP_CHECK(hipMemcpyAsync(chunk_offsets, e->offsets, (res->nchunks + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    return ZGPU_OK;
}

int zgpu_deflate_segments_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, const uint64_t *d_seg_offsets, uint64_t nseg,
                                 const zgpu_deflate_params *p, void *d_out, uint64_t out_cap, uint64_t *d_out_offsets,
                                 zgpu_deflate_result *res, void *hip_stream)
{
    if (!e || !d_seg_offsets) return ZGPU_STREAM_ERROR;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->stream;
    return deflate_device(e, static_cast<const uint8_t *>(d_in), in_bytes, d_seg_offsets, nseg, p, static_cast<uint8_t *>(d_out), out_cap,
                          d_out_offsets, res, st);
}

int zgpu_deflate_segments_host(zgpu_engine *e, const void *in, const uint64_t *seg_offsets, uint64_t nseg, const zgpu_deflate_params *p,
                               void *out, uint64_t out_cap, uint64_t *out_offsets, zgpu_deflate_result *res)
{
    if (!e || !p || !res || !in || !out || !seg_offsets || nseg == 0) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    const uint64_t in_bytes = seg_offsets[nseg];
    for (uint64_t k = 0; k < nseg; k++)
        if (seg_offsets[k + 1] < seg_offsets[k] || seg_offsets[k + 1] - seg_offsets[k] > kChunkMax) return fail(e, ZGPU_STREAM_ERROR, "segment longer than 65536 bytes");
    // (every segment may be a chunk of its own: with a non-default geometry each gets the allowance of a full chunk)
    const uint64_t bound = (e->geo_w != 15 || e->geo_m != 8) ? in_bytes + zgpu_deflate_bound_geometry(nseg * (uint64_t)kChunkMax, kChunkMax, e->geo_w, e->geo_m) - nseg * (uint64_t)kChunkMax + ((nseg * (uint64_t)kChunkMax) >> 3)
                                                             : in_bytes + nseg * 40 + 16;
    int rc = ensure_stage(e, in_bytes + (nseg + 1) * sizeof(uint64_t) + 64, bound);
    if (rc) return rc;
    const uint64_t tab_off = (in_bytes + 63) & ~63ull; // segment table staged behind the data
    if (in_bytes) ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in, in, in_bytes, hipMemcpyHostToDevice, e->stream));
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in + tab_off, seg_offsets, (nseg + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
    rc = deflate_device(e, e->stage_in, in_bytes, reinterpret_cast<const uint64_t *>(e->stage_in + tab_off), nseg, p, e->stage_out, bound,
                        nullptr, res, e->stream);
    if (rc) return rc;
    if (res->out_bytes > out_cap) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    ZGPU_HIP_CHECK(hipMemcpyAsync(out, e->stage_out, res->out_bytes, hipMemcpyDeviceToHost, e->stream));
    if (out_offsets) ZGPU_HIP_CHECK(hipMemcpyAsync(out_offsets, e->offsets, (nseg + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    return ZGPU_OK;
}

int zgpu_inflate_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, const uint64_t *d_chunk_offsets, uint64_t nchunks,
                        uint32_t chunk_size, void *d_out, uint64_t out_cap, zgpu_inflate_result *res, void *hip_stream)
{
    if (!e) return ZGPU_STREAM_ERROR;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->stream;
    return inflate_run(e, static_cast<const uint8_t *>(d_in), in_bytes, d_chunk_offsets, nchunks, chunk_size, static_cast<uint8_t *>(d_out), out_cap, res, st);
}

int zgpu_inflate_host(zgpu_engine *e, const void *in, uint64_t in_bytes, const uint64_t *chunk_offsets, uint64_t nchunks, uint32_t chunk_size,
                      void *out, uint64_t out_cap, zgpu_inflate_result *res)
{
    if (!e || !res || !in || !out || !chunk_offsets || nchunks == 0) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    const uint64_t need_out = nchunks * (uint64_t)(chunk_size ? chunk_size : kChunkMax);
    int rc = ensure_stage(e, in_bytes + 64, need_out);
    if (rc) return rc;
    if (nchunks + 1 > e->offsets_cap) {
        hipFree(e->offsets); e->offsets = nullptr; e->offsets_cap = 0;
        ZGPU_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&e->offsets), (nchunks + 1) * sizeof(uint64_t))); e->offsets_cap = nchunks + 1;
    }
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->stage_in, in, in_bytes, hipMemcpyHostToDevice, e->stream));
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->offsets, chunk_offsets, (nchunks + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
    // direct placement with room for every chunk: the output goes to the caller's buffer batch by batch, under the decoding of the next batch
    // (the device buffer has room for nchunks whole chunks; the caller's buffer has out_cap bytes, and no copy into it goes beyond them)
    const bool stream_out = chunk_size != 0 && chunk_size <= kChunkMax;
    rc = inflate_run(e, e->stage_in, in_bytes, e->offsets, nchunks, chunk_size, e->stage_out, need_out, res, e->stream, 0, nullptr, false,
                     stream_out ? static_cast<uint8_t *>(out) : nullptr, out_cap);
    if (rc) return rc;
    if (res->out_bytes > out_cap) return fail(e, ZGPU_BUF_ERROR, "output capacity too small");
    if (!stream_out) ZGPU_HIP_CHECK(hipMemcpyAsync(out, e->stage_out, res->out_bytes, hipMemcpyDeviceToHost, e->stream));
    ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    return ZGPU_OK;
}

int zgpu_inflate_set_dictionary(zgpu_engine *e, const void *dict, uint32_t len)
{
    if (!e || (!dict && len)) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    if (len > kWSize) { dict = static_cast<const uint8_t *>(dict) + (len - kWSize); len = kWSize; } // the window keeps the tail (inflate.c:1222-1226)
    if (len && !e->inf_dict) ZGPU_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&e->inf_dict), kWSize));
    if (len) { ZGPU_HIP_CHECK(hipMemcpyAsync(e->inf_dict, dict, len, hipMemcpyHostToDevice, e->stream)); ZGPU_HIP_CHECK(hipStreamSynchronize(e->stream)); }
    e->inf_dict_len = len;
    return ZGPU_OK;
}

int zgpu_inflate_set_checks(zgpu_engine *e, uint32_t mask)
{
    if (!e || mask > 3u) return fail(e, ZGPU_STREAM_ERROR, "checks: a mask of ZGPU_CHECK_ADLER32 | ZGPU_CHECK_CRC32");
    e->inf_checks = mask;
    return ZGPU_OK;
}

int zgpu_adler32_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, uint32_t *adler_out, void *hip_stream)
{
    if (!e || !adler_out || (!d_in && in_bytes)) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->stream;
    const uint64_t nchunks = in_bytes ? (in_bytes + kChunkMax - 1) / kChunkMax : 1;
    const uint32_t batch = (uint32_t)(nchunks < 65536 ? nchunks : 65536);
    int rc = ensure_deflate_ws(e, 1, false, 1);
    if (rc) return rc;
    // (the engine's grow-only scratch: these two are called per buffer by checksum-only users)
    ChunkMeta *meta = engine_meta(e, batch);
    uint64_t *offs = engine_offsets_scratch(e, nchunks + 1);
    if (!meta || !offs) return fail(e, ZGPU_MEM_ERROR, "checksum scratch");
    RunStateHost rs{}; rs.adler_a = 1;
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->run, &rs, sizeof rs, hipMemcpyHostToDevice, st));
    for (uint64_t c0 = 0; c0 < nchunks; c0 += batch) {
        const uint32_t nb = (uint32_t)(nchunks - c0 < batch ? nchunks - c0 : batch);
        ZGPU_HIP_CHECK(hipMemsetAsync(meta, 0, (size_t)nb * sizeof(ChunkMeta), st));
        ChunkGeom g{}; g.in = static_cast<const uint8_t *>(d_in); g.in_bytes = in_bytes; g.chunk_size = kChunkMax; g.chunk0 = c0; g.nchunks = nb; g.final_chunk = ~0ull;
        launch_adler(g, meta, st);
        launch_scan(meta, nb, c0, offs, e->run, ~0ull, st);
    }
    ZGPU_HIP_CHECK(hipMemcpyAsync(&rs, e->run, sizeof rs, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    *adler_out = rs.adler_a | (rs.adler_b << 16);
    return ZGPU_OK;
}

int zgpu_crc32_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, uint32_t *crc_out, void *hip_stream)
{
    if (!e || !crc_out || (!d_in && in_bytes)) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->stream;
    const uint64_t nchunks = in_bytes ? (in_bytes + kChunkMax - 1) / kChunkMax : 1;
    const uint32_t batch = (uint32_t)(nchunks < 65536 ? nchunks : 65536);
    int rc = ensure_deflate_ws(e, 1, false, 1);
    if (rc) return rc;
    // (the engine's grow-only scratch: these two are called per buffer by checksum-only users)
    ChunkMeta *meta = engine_meta(e, batch);
    uint64_t *offs = engine_offsets_scratch(e, nchunks + 1);
    if (!meta || !offs) return fail(e, ZGPU_MEM_ERROR, "checksum scratch");
    RunStateHost rs{}; rs.adler_a = 1;
    ZGPU_HIP_CHECK(hipMemcpyAsync(e->run, &rs, sizeof rs, hipMemcpyHostToDevice, st));
    for (uint64_t c0 = 0; c0 < nchunks; c0 += batch) {
        const uint32_t nb = (uint32_t)(nchunks - c0 < batch ? nchunks - c0 : batch);
        ZGPU_HIP_CHECK(hipMemsetAsync(meta, 0, (size_t)nb * sizeof(ChunkMeta), st));
        ChunkGeom g{}; g.in = static_cast<const uint8_t *>(d_in); g.in_bytes = in_bytes; g.chunk_size = kChunkMax; g.chunk0 = c0; g.nchunks = nb; g.final_chunk = ~0ull;
        launch_crc(g, meta, st);
        launch_scan(meta, nb, c0, offs, e->run, ~0ull, st, true);
    }
    ZGPU_HIP_CHECK(hipMemcpyAsync(&rs, e->run, sizeof rs, hipMemcpyDeviceToHost, st));
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    *crc_out = rs.crc;
    return ZGPU_OK;
}

void zgpu_profile_enable(zgpu_engine *e, int on) { if (e) e->prof = on != 0; }
void zgpu_profile_reset(zgpu_engine *e) { if (e) { memset(e->ms, 0, sizeof e->ms); memset(e->launches, 0, sizeof e->launches); } }
int zgpu_profile_get(zgpu_engine *e, int stage, double *ms, uint64_t *launches)
{
    if (!e || stage < 0 || stage >= ZGPU_STAGE_COUNT) return ZGPU_STREAM_ERROR;
    if (ms) *ms = e->ms[stage];
    if (launches) *launches = e->launches[stage];
    return ZGPU_OK;
}
const char *zgpu_stage_name(int stage)
{
    static const char *names[ZGPU_STAGE_COUNT] = {"chain", "match", "parse", "lz_serial", "huffman", "stitch", "inflate"};
    return stage >= 0 && stage < ZGPU_STAGE_COUNT ? names[stage] : "?";
}

int zgpu_corpus_fill_device(zgpu_engine *e, uint32_t kind, uint64_t seed, uint64_t first_chunk, uint64_t nchunks, void *d_out, void *hip_stream)
{
    if (!e || !d_out) return fail(e, ZGPU_STREAM_ERROR, "null argument");
    ZGPU_HIP_CHECK(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->stream;
    launch_corpus(kind, seed, first_chunk, nchunks, static_cast<uint8_t *>(d_out), st);
    ZGPU_HIP_CHECK(hipGetLastError());
    ZGPU_HIP_CHECK(hipStreamSynchronize(st));
    return ZGPU_OK;
}

#pragma GCC visibility pop
} // extern "C"

// hooks used by zgpu_lz_parallel.hip to time its sub-stages with the engine's event pool
namespace zgpu {
void *engine_scratch(zgpu_engine *e, size_t bytes)
{
    if (bytes > e->inf_status_cap) {
        hipFree(e->inf_status); e->inf_status = nullptr; e->inf_status_cap = 0;
        if (hipMalloc(&e->inf_status, bytes) != hipSuccess) return nullptr;
        e->inf_status_cap = bytes;
    }
    return e->inf_status;
}
void *engine_scratch2(zgpu_engine *e, size_t bytes)
{
    if (bytes > e->inf_slots_cap) {
        hipFree(e->inf_slots); e->inf_slots = nullptr; e->inf_slots_cap = 0;
        if (hipMalloc(&e->inf_slots, bytes) != hipSuccess) return nullptr;
        e->inf_slots_cap = bytes;
    }
    return e->inf_slots;
}
void *engine_run_state(zgpu_engine *e) { return e->run; }
uint8_t *engine_stage_in(zgpu_engine *e) { return e->stage_in; }
uint8_t *engine_stage_out(zgpu_engine *e) { return e->stage_out; }
const uint8_t *engine_inflate_dict(zgpu_engine *e) { return e->inf_dict; }
uint32_t engine_inflate_dict_len(zgpu_engine *e) { return e->inf_dict_len; }
uint32_t engine_inflate_checks(zgpu_engine *e) { return e->inf_checks; }
int engine_ensure_stage(zgpu_engine *e, uint64_t in_bytes, uint64_t out_bytes) { return ensure_stage(e, in_bytes, out_bytes); }
hipStream_t engine_stream(zgpu_engine *e) { return e->stream; }
hipStream_t engine_copy_stream(zgpu_engine *e) { return e->copy_stream; }
hipEvent_t engine_copy_event(zgpu_engine *e, size_t i)
{
    while (e->copy_ev.size() <= i) { hipEvent_t ev; if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr; e->copy_ev.push_back(ev); }
    return e->copy_ev[i];
}
ChunkMeta *engine_meta(zgpu_engine *e, uint32_t batch)
{
    if (batch > e->inf_meta_cap) {
        hipFree(e->inf_meta); e->inf_meta = nullptr; e->inf_meta_cap = 0;
        if (hipMalloc(reinterpret_cast<void **>(&e->inf_meta), (size_t)batch * sizeof(ChunkMeta)) != hipSuccess) return nullptr;
        e->inf_meta_cap = batch;
    }
    return e->inf_meta;
}
uint64_t *engine_offsets_scratch(zgpu_engine *e, uint64_t n)
{
    if (n > e->inf_offs_cap) {
        hipFree(e->inf_offs); e->inf_offs = nullptr; e->inf_offs_cap = 0;
        if (hipMalloc(reinterpret_cast<void **>(&e->inf_offs), n * sizeof(uint64_t)) != hipSuccess) return nullptr;
        e->inf_offs_cap = n;
    }
    return e->inf_offs;
}
int engine_device(zgpu_engine *e) { return e->device; }
void engine_collect(zgpu_engine *e) { collect_spans(e); }
int engine_fail(zgpu_engine *e, int code, const char *msg) { return fail(e, code, msg); }
void prof_span_begin(void *eng, hipStream_t st, hipEvent_t *a)
{
    zgpu_engine *e = static_cast<zgpu_engine *>(eng);
    if (e && e->prof) { *a = next_event(e); hipEventRecord(*a, st); }
}
void prof_span_end(void *eng, hipStream_t st, int stage, hipEvent_t a)
{
    zgpu_engine *e = static_cast<zgpu_engine *>(eng);
    if (e && e->prof) { hipEvent_t b = next_event(e); hipEventRecord(b, st); e->spans.push_back({stage, a, b}); }
}
} // namespace zgpu
