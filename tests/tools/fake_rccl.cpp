// TEST DOUBLE of the nine RCCL entry points zlib_amd/csrc/zgpu_comm.hip binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather,
// ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), for ranks that share ONE GPU box: RCCL refuses two ranks on one device, so
// the N > 1 arithmetic of the C library's gather (offsets, header, trailer, Adler join, which ranks send at all) could otherwise only run on an 8-GPU
// node.  Data travels through a POSIX shared-memory file named by the unique id; device buffers are copied with hipMemcpy.  Loaded through
// ZAMD_RCCL_LIB (tests/test_gpu_comm_world.py builds it with hipcc).  Not part of the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>

namespace {
constexpr int kMaxRanks = 16;
constexpr size_t kSlotCap = 24u << 20, kSmall = 4096;
struct Slot { volatile uint64_t seq_w, seq_r, size, dst; };
struct Shm {
    volatile uint64_t arrived, generation, attached;
    Slot slot[kMaxRanks];
    uint8_t small[kMaxRanks][kSmall];
};
struct Comm { Shm *m; uint8_t *data; int world, rank; size_t bytes; char name[64]; };
struct Op { bool send; void *buf; size_t n; int peer; Comm *c; hipStream_t st; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

bool wait_until(volatile uint64_t *p, uint64_t at_least)
{
    for (int i = 0; i < 600000; i++) { if (__atomic_load_n(p, __ATOMIC_ACQUIRE) >= at_least) return true; usleep(100); }
    return false;
}
bool barrier(Comm *c)
{
    const uint64_t gen = __atomic_load_n(&c->m->generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&c->m->arrived, 1, __ATOMIC_ACQ_REL) == (uint64_t)c->world) {
        __atomic_store_n(&c->m->arrived, 0, __ATOMIC_RELEASE);
        __atomic_add_fetch(&c->m->generation, 1, __ATOMIC_ACQ_REL);
        return true;
    }
    return wait_until(&c->m->generation, gen + 1);
}
size_t type_bytes(ncclDataType_t t)
{
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; case ncclInt64: case ncclUint64: case ncclFloat64: return 8; default: return 2; }
}
ncclResult_t run(const Op &o)
{
    Comm *c = o.c;
    if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t at = 0; at == 0 || at < o.n; at += kSlotCap) { // (a message longer than a slot goes piece by piece, both sides cut it alike)
        const size_t piece = o.n - at < kSlotCap ? o.n - at : kSlotCap;
        if (o.send) {
            Slot &s = c->m->slot[c->rank];
            if (!wait_until(&s.seq_r, s.seq_w)) return ncclSystemError; // the slot is free again
            if (hipMemcpy(c->data + (size_t)c->rank * kSlotCap, static_cast<uint8_t *>(o.buf) + at, piece, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            s.size = piece; s.dst = (uint64_t)o.peer;
            __atomic_add_fetch(&s.seq_w, 1, __ATOMIC_ACQ_REL);
        } else {
            Slot &s = c->m->slot[o.peer];
            if (!wait_until(&s.seq_w, s.seq_r + 1)) return ncclSystemError;
            if (s.size != piece || s.dst != (uint64_t)c->rank) return ncclInvalidUsage; // a send and its receive must agree in size and peer
            if (hipMemcpy(static_cast<uint8_t *>(o.buf) + at, c->data + (size_t)o.peer * kSlotCap, piece, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            __atomic_add_fetch(&s.seq_r, 1, __ATOMIC_ACQ_REL);
        }
    }
    return ncclSuccess;
}
} // namespace

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/zamd_fake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank)
{
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return ncclInvalidArgument;
    Comm *c = new Comm;
    c->world = world; c->rank = rank; c->bytes = sizeof(Shm) + (size_t)world * kSlotCap;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->m = static_cast<Shm *>(p); c->data = static_cast<uint8_t *>(p) + sizeof(Shm);
    __atomic_add_fetch(&c->m->attached, 1, __ATOMIC_ACQ_REL);
    if (!wait_until(&c->m->attached, (uint64_t)world)) { delete c; return ncclSystemError; }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t h)
{
    Comm *c = reinterpret_cast<Comm *>(h);
    if (!c) return ncclSuccess;
    barrier(c);
    munmap(c->m, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t h, hipStream_t st)
{
    Comm *c = reinterpret_cast<Comm *>(h);
    const size_t n = count * type_bytes(t);
    if (n > kSmall) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->m->small[c->rank], send, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; r++)
        if (hipMemcpy(static_cast<uint8_t *>(recv) + (size_t)r * n, c->m->small[r], n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd()
{
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (int pass = 0; pass < 2 && rc == ncclSuccess; pass++) // sends first: a rank that sends and receives in one group must not wait for itself
        for (const Op &o : g_ops) if (o.send == (pass == 0) && rc == ncclSuccess) rc = run(o);
    g_ops.clear();
    return rc;
}
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t h, hipStream_t st)
{
    const Op o{true, const_cast<void *>(buf), count * type_bytes(t), peer, reinterpret_cast<Comm *>(h), st};
    if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
    return run(o);
}
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t h, hipStream_t st)
{
    const Op o{false, buf, count * type_bytes(t), peer, reinterpret_cast<Comm *>(h), st};
    if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
    return run(o);
}
__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL: the exchange failed"; }
}
