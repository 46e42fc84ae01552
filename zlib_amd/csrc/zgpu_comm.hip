// zgpu_comm.hip -- the multi-GPU exchange of the chunked engine behind the C ABI: the per-rank raw bodies are gathered into ONE RFC 1950 stream on
// rank 0 over RCCL (xGMI), device to device (BASELINE.json north_star: "an RCCL gather over xGMI for the final concatenated stream").
//
// One process per GPU.  Chunks are pure functions of their bytes, so rank r compresses its contiguous chunk range into a raw body of its own (only
// the last rank's last chunk carries BFINAL) with zgpu_deflate_device; the only exchange is
//   1. ncclAllGather of (body bytes, Adler-32, input bytes): 3 x u64 per rank -- every rank learns the layout, rank 0 the exact size of the stream;
//   2. one ncclSend per peer and, on rank 0, one ncclRecv per peer at the prefix-summed offset inside ONE group (RCCL has no gatherv; on xGMI every
//      peer has a link of its own to GPU 0: posted as a group the transfers run side by side);
//   3. rank 0 writes the 2-byte zlib header in front and the big-endian Adler-32 of the whole input behind (the checksums of the ranks combined:
//      adler32_combine, /root/reference/qcsrc/adler32.c:128-149).
// RCCL is opened with dlopen when the first communicator is made: a process that never shards does not load it, and a process that has PyTorch's
// copy mapped (the same soname) shares it.  The reference has no distributed path.
#include "zgpu_common.h"
#include "../../include/zamd_gpu.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
thread_local char g_comm_err[256] = "";

int comm_fail(int rc, const char *what, const char *detail = nullptr)
{
    snprintf(g_comm_err, sizeof g_comm_err, "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
    return rc;
}
bool rccl_open()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return true;
    const char *named = getenv("ZAMD_RCCL_LIB"); // another build of RCCL (or the test double of tests/tools/fake_rccl.cpp, for ranks that share one GPU)
    void *h = (named && named[0]) ? dlopen(named, RTLD_NOW) : nullptr;
    if (named && named[0] && !h) return false;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD); // PyTorch's copy, if this process has it
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
    if (!h) h = dlopen("librccl.so", RTLD_NOW);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW);
    if (!h) return false;
    Rccl r; r.h = h;
#define ZSYM(field, name) do { r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) return false; } while (0)
    ZSYM(GetUniqueId, "ncclGetUniqueId"); ZSYM(CommInitRank, "ncclCommInitRank"); ZSYM(CommDestroy, "ncclCommDestroy"); ZSYM(AllGather, "ncclAllGather");
    ZSYM(Send, "ncclSend"); ZSYM(Recv, "ncclRecv"); ZSYM(GroupStart, "ncclGroupStart"); ZSYM(GroupEnd, "ncclGroupEnd"); ZSYM(GetErrorString, "ncclGetErrorString");
#undef ZSYM
    g_rccl = r;
    return true;
}
#define ZNCCL(expr) do { const ncclResult_t r__ = (expr); if (r__ != ncclSuccess) return comm_fail(ZGPU_STREAM_ERROR, #expr, g_rccl.GetErrorString(r__)); } while (0)
#define ZHIP(expr) do { const hipError_t r__ = (expr); if (r__ != hipSuccess) return comm_fail(ZGPU_MEM_ERROR, #expr, hipGetErrorString(r__)); } while (0)

// Adler-32 of a concatenation (adler32.c:128-149; zlib_amd/shard.py adler_join)
uint32_t adler_join(uint32_t x, uint32_t y, uint64_t len_y)
{
    const uint32_t B = 65521u, ax = x & 0xffffu, bx = x >> 16, ay = y & 0xffffu, by = y >> 16;
    const uint32_t a = (ax + ay + B - 1) % B;
    const uint32_t b = (uint32_t)((bx + by + (len_y % B) * ((ax + B - 1) % B)) % B);
    return a | (b << 16);
}
} // namespace

struct zgpu_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    uint64_t *d_table = nullptr; // (world + 1) x 3 u64: this rank's triple, then everybody's
    uint8_t *d_frame = nullptr;  // header and trailer on their way to the stream
};

extern "C" {
#pragma GCC visibility push(default)

const char *zgpu_comm_error(void) { return g_comm_err; }

int zgpu_comm_unique_id(void *id128)
{
    if (!id128) return comm_fail(ZGPU_STREAM_ERROR, "null argument");
    if (!rccl_open()) return comm_fail(ZGPU_MEM_ERROR, "RCCL (librccl.so.1) could not be opened", dlerror());
    ncclUniqueId id;
    ZNCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof id == ZGPU_COMM_ID_BYTES, "the id travels as 128 opaque bytes");
    memcpy(id128, &id, sizeof id);
    return ZGPU_OK;
}

int zgpu_comm_create(int device, int world, int rank, const void *id128, zgpu_comm **out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return comm_fail(ZGPU_STREAM_ERROR, "bad communicator arguments");
    if (!rccl_open()) return comm_fail(ZGPU_MEM_ERROR, "RCCL (librccl.so.1) could not be opened", dlerror());
    ZHIP(hipSetDevice(device));
    zgpu_comm *c = new zgpu_comm;
    c->world = world; c->rank = rank; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return comm_fail(ZGPU_STREAM_ERROR, "ncclCommInitRank", g_rccl.GetErrorString(r)); }
    if (hipMalloc(reinterpret_cast<void **>(&c->d_table), (size_t)(world + 1) * 3 * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->d_frame), 64) != hipSuccess) { zgpu_comm_destroy(c); return comm_fail(ZGPU_MEM_ERROR, "hipMalloc"); }
    *out = c;
    return ZGPU_OK;
}

void zgpu_comm_destroy(zgpu_comm *c)
{
    if (!c) return;
    if (c->comm) g_rccl.CommDestroy(c->comm);
    hipFree(c->d_table); hipFree(c->d_frame);
    delete c;
}

// Where the bodies lie in the stream: offsets[r] = first byte of rank r's body (offsets[0] = 2, behind the zlib header), offsets[world] = where the
// trailer goes, *total = the stream's length.  (The one place this arithmetic lives: zgpu_deflate_gather below and zlib_amd/shard.py both call it.)
void zgpu_gather_layout(int world, const uint64_t *table, uint64_t *offsets, uint64_t *total)
{
    uint64_t at = 2;
    for (int r = 0; r < world; r++) { offsets[r] = at; at += table[3 * r]; }
    offsets[world] = at;
    if (total) *total = at + 4;
}

// All ranks: exchange (body bytes, Adler-32, input bytes).  table: world x 3 u64 (host), *total: the length of the stream rank 0 will hold.
int zgpu_deflate_gather_sizes(zgpu_comm *c, uint64_t body_bytes, uint32_t adler32, uint64_t in_bytes, uint64_t *table, uint64_t *total, void *hip_stream)
{
    if (!c || !table) return comm_fail(ZGPU_STREAM_ERROR, "null argument");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ZHIP(hipSetDevice(c->device));
    const uint64_t mine[3] = {body_bytes, adler32, in_bytes};
    ZHIP(hipMemcpyAsync(c->d_table, mine, sizeof mine, hipMemcpyHostToDevice, st));
    ZNCCL(g_rccl.AllGather(c->d_table, c->d_table + 3, 3, ncclUint64, c->comm, st));
    ZHIP(hipMemcpyAsync(table, c->d_table + 3, (size_t)c->world * 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    ZHIP(hipStreamSynchronize(st));
    std::vector<uint64_t> offs((size_t)c->world + 1);
    zgpu_gather_layout(c->world, table, offs.data(), total);
    return ZGPU_OK;
}

// All ranks, after zgpu_deflate_gather_sizes: d_body (device) = this rank's raw body.  Rank 0: d_out (device, out_cap >= *total of the sizes call)
// receives header + bodies in rank order + trailer; *adler_out = Adler-32 of the whole input.  Other ranks pass d_out = NULL.
int zgpu_deflate_gather(zgpu_comm *c, const void *d_body, const uint64_t *table, int level, void *d_out, uint64_t out_cap, uint32_t *adler_out, void *hip_stream)
{
    if (!c || !table || (c->rank == 0 && !d_out)) return comm_fail(ZGPU_STREAM_ERROR, "null argument");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ZHIP(hipSetDevice(c->device));
    std::vector<uint64_t> offs((size_t)c->world + 1);
    uint64_t total = 0;
    zgpu_gather_layout(c->world, table, offs.data(), &total);
    const uint64_t mine = table[3 * c->rank];
    if (mine && !d_body) return comm_fail(ZGPU_STREAM_ERROR, "null body");
    if (c->rank != 0) {
        if (mine) { ZNCCL(g_rccl.GroupStart()); ZNCCL(g_rccl.Send(d_body, mine, ncclUint8, 0, c->comm, st)); ZNCCL(g_rccl.GroupEnd()); }
        ZHIP(hipStreamSynchronize(st)); // (the call returns when the body has been sent, on every rank: d_body may be reused or freed then -- ADVICE round 3)
        return ZGPU_OK;
    }
    if (total > out_cap) return comm_fail(ZGPU_BUF_ERROR, "output capacity too small");
    uint8_t *out = static_cast<uint8_t *>(d_out);
    // header (deflate.c:625-641 for windowBits 15, no dictionary) and trailer
    unsigned h = (8u + (7u << 4)) << 8;
    h |= (level < 2 ? 0u : level < 6 ? 1u : level == 6 ? 2u : 3u) << 6;
    h += 31 - h % 31;
    uint32_t a = 1;
    for (int r = 0; r < c->world; r++) a = adler_join(a, (uint32_t)table[3 * r + 1], table[3 * r + 2]);
    const uint8_t frame[6] = {(uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(a >> 24), (uint8_t)(a >> 16), (uint8_t)(a >> 8), (uint8_t)a};
    ZHIP(hipMemcpyAsync(c->d_frame, frame, sizeof frame, hipMemcpyHostToDevice, st));
    ZHIP(hipMemcpyAsync(out, c->d_frame, 2, hipMemcpyDeviceToDevice, st));
    ZHIP(hipMemcpyAsync(out + offs[c->world], c->d_frame + 2, 4, hipMemcpyDeviceToDevice, st));
    if (mine) ZHIP(hipMemcpyAsync(out + offs[0], d_body, mine, hipMemcpyDeviceToDevice, st));
    bool any = false;
    for (int r = 1; r < c->world; r++) any = any || table[3 * r] != 0;
    if (any) {
        ZNCCL(g_rccl.GroupStart());
        for (int r = 1; r < c->world; r++)
            if (table[3 * r]) ZNCCL(g_rccl.Recv(out + offs[r], table[3 * r], ncclUint8, r, c->comm, st));
        ZNCCL(g_rccl.GroupEnd());
    }
    ZHIP(hipStreamSynchronize(st)); // (`frame` is a local)
    if (adler_out) *adler_out = a;
    return ZGPU_OK;
}

#pragma GCC visibility pop
}
