// TEST INFRASTRUCTURE -- not RCCL, not part of the product.  A stand-in for librccl.so that lets the library's RCCL branch
// (pgp_comm kind 1 in csrc/sharded.hip: ncclCommInitRank at world > 1, ncclBroadcast on the communication stream with the
// look-ahead events of pgp_sharded_exact_fit, ncclAllReduce / ncclAllGather of the epilogue and of the host collectives)
// execute at world 2 .. 4 with every rank on ONE GPU, which real RCCL refuses (duplicate device).  The six entry points the
// library binds (csrc/sharded.hip rccl_load) move the data through a POSIX shared-memory segment named after the unique id.
//
// Semantics kept: a collective is ordered on the stream it is given -- the stub drains that stream (hipStreamSynchronize), so
// everything the caller made the stream wait for (hipStreamWaitEvent on the producer's event) has happened before the source is
// read, and whatever the caller queues behind the call sees the result.  Semantics NOT kept: real RCCL returns at once and runs
// asynchronously; the stub blocks the host until every rank has arrived.  The call ORDER must therefore be the same on every rank
// (it is: the library's collectives are issued in program order), and a missing producer-side event shows as wrong numbers.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t SLOT = (size_t)64 << 20;          // bytes per rank and chunk
constexpr int MAXW = 8;

struct Control {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    std::atomic<long> calls[4];                    // broadcast, all-reduce, all-gather, bytes moved (diagnostics)
};

struct Comm {
    int world, rank;
    char name[64];
    void* base;
    size_t bytes;
    Control* ctl;
    char* data;                                    // world slots of SLOT bytes
    int local_gen;
};

struct Uid { char internal[128]; };

bool barrier(Comm* c) {                            // sense-reversing; gives up after ~120 s (a peer died)
    const int gen = c->ctl->generation.load(std::memory_order_acquire);
    if (c->ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        c->ctl->arrived.store(0, std::memory_order_relaxed);
        c->ctl->generation.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    const time_t t0 = time(nullptr);
    long spins = 0;
    while (c->ctl->generation.load(std::memory_order_acquire) == gen) {
        if ((++spins & 1023) == 0) { sched_yield(); if (time(nullptr) - t0 > 120) return false; }
    }
    return true;
}

constexpr int OK = 0, ERR_SYSTEM = 2, ERR_INTERNAL = 3, ERR_ARG = 4;

}  // namespace

extern "C" {

int ncclGetUniqueId(Uid* id) {
    if (!id) return ERR_ARG;
    memset(id, 0, sizeof(*id));
    static std::atomic<int> serial{0};                            // several communicators per process and second
    snprintf(id->internal, sizeof(id->internal), "/pgp_rccl_stub_%d_%ld_%d", (int)getpid(), (long)time(nullptr), serial.fetch_add(1));
    return OK;
}

int ncclCommInitRank(void** comm, int world, Uid id, int rank) {
    if (!comm || world < 1 || world > MAXW || rank < 0 || rank >= world) return ERR_ARG;
    Comm* c = new Comm();
    c->world = world; c->rank = rank; c->local_gen = 0;
    memcpy(c->name, id.internal, sizeof(c->name) - 1);
    c->name[sizeof(c->name) - 1] = 0;
    c->bytes = 4096 + SLOT * (size_t)world;
    int fd = -1;
    for (int tries = 0; tries < 6000 && fd < 0; ++tries) {        // rank 0 creates, the others wait for it
        fd = rank == 0 ? shm_open(c->name, O_CREAT | O_RDWR, 0600) : shm_open(c->name, O_RDWR, 0600);
        if (fd < 0) usleep(10000);
    }
    if (fd < 0) { delete c; return ERR_SYSTEM; }
    if (rank == 0 && ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return ERR_SYSTEM; }
    if (rank != 0) {                                               // wait until rank 0 has sized the segment
        struct stat st;
        for (int tries = 0; tries < 6000; ++tries) { if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break; usleep(10000); }
    }
    c->base = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) { delete c; return ERR_SYSTEM; }
    c->ctl = (Control*)c->base;                                    // a fresh segment is zero: counters start at 0
    c->data = (char*)c->base + 4096;
    c->ctl->attached.fetch_add(1);
    for (int tries = 0; tries < 12000 && c->ctl->attached.load() < world; ++tries) usleep(10000);
    if (c->ctl->attached.load() < world) { munmap(c->base, c->bytes); delete c; return ERR_SYSTEM; }
    *comm = c;
    return OK;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return OK;
    if (getenv("PGP_RCCL_STUB_STATS") && c->rank == 0)
        fprintf(stderr, "[rccl stub] world %d: %ld broadcasts, %ld all-reduces, %ld all-gathers, %ld bytes\n", c->world,
                c->ctl->calls[0].load(), c->ctl->calls[1].load(), c->ctl->calls[2].load(), c->ctl->calls[3].load());
    const bool last = c->ctl->attached.fetch_sub(1) == 1;
    munmap(c->base, c->bytes);
    if (last || c->rank == 0) shm_unlink(c->name);
    delete c;
    return OK;
}

int ncclCommAbort(void* comm) { return ncclCommDestroy(comm); }

const char* ncclGetErrorString(int rc) {
    switch (rc) { case OK: return "ok"; case ERR_SYSTEM: return "stub: shared memory"; case ERR_ARG: return "stub: bad argument";
                  default: return "stub: a peer never arrived / hip error"; }
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (!c || dtype != 8 || root < 0 || root >= c->world) return ERR_ARG;
    if (hipStreamSynchronize(st) != hipSuccess) return ERR_INTERNAL;
    const size_t bytes = count * 8;
    for (size_t off = 0; off < bytes || off == 0; off += SLOT) {
        const size_t nb = bytes - off < SLOT ? bytes - off : SLOT;
        if (c->rank == root && nb && hipMemcpy(c->data, (const char*)send + off, nb, hipMemcpyDeviceToHost) != hipSuccess) return ERR_INTERNAL;
        if (!barrier(c)) return ERR_INTERNAL;
        if (c->rank != root && nb && hipMemcpy((char*)recv + off, c->data, nb, hipMemcpyHostToDevice) != hipSuccess) return ERR_INTERNAL;
        if (c->rank == root && recv != send && nb && hipMemcpy((char*)recv + off, (const char*)send + off, nb, hipMemcpyDeviceToDevice) != hipSuccess) return ERR_INTERNAL;
        if (!barrier(c)) return ERR_INTERNAL;
        if (bytes == 0) break;
    }
    if (c->rank == root) { c->ctl->calls[0].fetch_add(1); c->ctl->calls[3].fetch_add((long)bytes * (c->world - 1)); }
    return OK;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (!c || dtype != 8 || (op != 0 && op != 2)) return ERR_ARG;
    if (hipStreamSynchronize(st) != hipSuccess) return ERR_INTERNAL;
    const size_t per = SLOT / 8;
    for (size_t off = 0; off < count; off += per) {
        const size_t nd = count - off < per ? count - off : per;
        if (hipMemcpy(c->data + (size_t)c->rank * SLOT, (const double*)send + off, nd * 8, hipMemcpyDeviceToHost) != hipSuccess) return ERR_INTERNAL;
        if (!barrier(c)) return ERR_INTERNAL;
        double* tmp = (double*)malloc(nd * 8);
        if (!tmp) return ERR_SYSTEM;
        memcpy(tmp, c->data, nd * 8);                               // rank order: every rank computes the same bits
        for (int r = 1; r < c->world; ++r) {
            const double* s = (const double*)(c->data + (size_t)r * SLOT);
            if (op == 0) for (size_t i = 0; i < nd; ++i) tmp[i] += s[i];
            else for (size_t i = 0; i < nd; ++i) tmp[i] = s[i] > tmp[i] ? s[i] : tmp[i];
        }
        const hipError_t e = hipMemcpy((double*)recv + off, tmp, nd * 8, hipMemcpyHostToDevice);
        free(tmp);
        if (e != hipSuccess) return ERR_INTERNAL;
        if (!barrier(c)) return ERR_INTERNAL;
    }
    if (c->rank == 0) { c->ctl->calls[1].fetch_add(1); c->ctl->calls[3].fetch_add((long)count * 8 * (c->world - 1)); }
    return OK;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (!c || dtype != 8) return ERR_ARG;
    if (count * 8 > SLOT) return ERR_ARG;                           // the library's all-gathers are a few records per rank
    if (hipStreamSynchronize(st) != hipSuccess) return ERR_INTERNAL;
    if (count && hipMemcpy(c->data + (size_t)c->rank * SLOT, send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ERR_INTERNAL;
    if (!barrier(c)) return ERR_INTERNAL;
    for (int r = 0; r < c->world; ++r)
        if (count && hipMemcpy((double*)recv + (size_t)r * count, c->data + (size_t)r * SLOT, count * 8, hipMemcpyHostToDevice) != hipSuccess) return ERR_INTERNAL;
    if (!barrier(c)) return ERR_INTERNAL;
    if (c->rank == 0) { c->ctl->calls[2].fetch_add(1); c->ctl->calls[3].fetch_add((long)count * 8 * (c->world - 1)); }
    return OK;
}

}  // extern "C"
