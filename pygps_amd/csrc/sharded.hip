// ONE exact-GP fit spread over the GPUs of a node: Exact.evaluate (Core/inf.py:353-384) with the
// factorisation of B = K/sn2 + I (tools.jitchol, Core/tools.py:31-77), the solves for alpha (:363), nlZ (:370), the inverse
// behind Q (:373) and the gradient sums (:374-377) all distributed.  No reference counterpart: pyGPs factors on one host.
//
// Layout: 1-D block-cyclic over column panels of w columns (w = 512, 1024 from N = 12288: the single-GPU sweep's panel),
// panel p on rank p % world, one process per GPU.  A panel is stored in the LOGICAL row space of the single-GPU sweep
// (capi.hip: potrf_blocked_v2) -- factor rows | 128 right-hand-side rows | fused-inverse rows E = L^-T -- compacted:
//
//     unfactored panel j  P_j : (np + 128) x w,  row r = logical row - j w :  [ diagonal block w | rows below | rhs 128 | E rows < j w ]
//     solved panel p      Y_p : (np + 128) x w,  row r = logical row - (p+1) w : [ rows below | rhs 128 | E rows < (p+1) w ]
//
// so one contiguous Y_p is everything the other ranks need from panel p: it is what gets broadcast (np + 128 rows for every
// p: the factor's rows shrink as the inverse's rows grow).  Per step p, exactly the single-GPU sweep:
//     owner(p)      D(p): diagonal block -> L_D, E_D = L_D^-T (leaf chain);  S(p): Y_p = P_p[w:] E_D (one MFMA GEMM)
//     broadcast     Y_p from owner(p)                         [RCCL over xGMI; per link bound: (np+128) w 8 B / ~153 GB/s]
//     every rank    TU(p): P_j -= Y_p[rows of j] Y_p[block j]'  for ALL its owned j > p in ONE batched launch
//                   (GemmArgs::batch_dm: the products shrink with j), first-touch of the E rows of block p
// with the depth-1 look-ahead of the single-GPU sweep: the owner of panel p+1 updates that panel first, factors it on the
// panel stream and solves it, and its broadcast travels on the communication stream while every rank (the owner too) works
// through the rest of step p.  Storage: a factored panel's P buffer is dead after S, so Y_k of local panel k goes into the
// buffer P_{k-1} leaves behind: (nloc + 1) buffers per rank, i.e. (np + 128) np / world doubles + one panel.
//
// What a fit needs beyond the factor needs almost no communication, because everything is LINEAR in the panels:
//     alpha = E z / sn2 = sum_p E_p z_p / sn2        per-rank partial matvecs, ONE all-reduce of np + 3 doubles
//     log det, z'z                                    per-panel scalars, in the same all-reduce
//     B^-1 = E E' = sum_p E_p E_p'                    every rank sees every Y_p (it is broadcast), and Y_p carries E_p: rank r
//                                                     accumulates the COLUMN STRIPS j = r, r + world, ... of the lower triangle
//                                                     of B^-1 (strip j: rows >= j w, w columns, stored as tall as it needs to be)
//                                                     from every E_p, p >= j, in ONE batched launch per step (N^3 / 3 / world
//                                                     flops per rank, np^2 / (2 world) doubles per rank; round 3 kept a full
//                                                     np^2 partial on every rank).  No communication: the strips are complete.
//     dnlZ_h = 1/2 sum_ij (B^-1/sn2 - alpha alpha')_ij dK_h,ij    the single-GPU Hadamard reduce, restricted to the tile rows of
//                                                     the rank's strips, and ONE all-reduce of ncov + 1 doubles.
// The posterior stays distributed: pgp_sfactor keeps the rank's Y panels (factor rows + E columns), alpha and the scaled
// coordinates; pgp_sharded_predict forms V = L^-1 Ks / sn = E' Ks / sn panel by panel on the owners (one MFMA product per
// owned panel -- E is there, no triangular solve) and all-reduces the column sums of squares (Core/gp.py:395-417).
// Transport: pgp_comm -- RCCL bound at run time (dlopen of librccl: ncclBroadcast on a communication stream, events between
// it and the compute streams, no host synchronisation inside the sweep), or host call-backs on staged host buffers (the
// self-test transport: gloo through torch.distributed lets several ranks share the one GPU of a test box).
#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"

// ------------------------------------------------------------------------------------------------------------------
// transport
// ------------------------------------------------------------------------------------------------------------------
namespace {

typedef struct { char internal[128]; } rccl_uid;          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
constexpr int RCCL_DOUBLE = 8, RCCL_SUM = 0, RCCL_MAX = 2;   // ncclFloat64, ncclSum, ncclMax (rccl.h)

struct RcclApi {
    void* dl = nullptr;
    int (*GetUniqueId)(rccl_uid*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_uid, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*CommAbort)(rccl_comm_t) = nullptr;             // optional
    const char* (*GetErrorString)(int) = nullptr;
};

static int rccl_load(const char* path, RcclApi& api) {
    const char* names[] = {path && path[0] ? path : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (!nm) continue;
        api.dl = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (api.dl) break;
    }
    if (!api.dl) {
        pgp_set_last_hip_error(hipErrorSharedObjectInitFailed, "dlopen(librccl)", __FILE__, __LINE__);
        return PGP_ERR_HIP;
    }
    api.GetUniqueId = (int (*)(rccl_uid*))dlsym(api.dl, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_uid, int))dlsym(api.dl, "ncclCommInitRank");
    api.CommDestroy = (int (*)(rccl_comm_t))dlsym(api.dl, "ncclCommDestroy");
    api.Broadcast = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(api.dl, "ncclBroadcast");
    api.AllReduce = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(api.dl, "ncclAllReduce");
    api.AllGather = (int (*)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t))dlsym(api.dl, "ncclAllGather");
    api.CommAbort = (int (*)(rccl_comm_t))dlsym(api.dl, "ncclCommAbort");
    api.GetErrorString = (const char* (*)(int))dlsym(api.dl, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Broadcast || !api.AllReduce || !api.AllGather) {
        pgp_set_last_hip_error(hipErrorSharedObjectSymbolNotFound, "dlsym(librccl)", __FILE__, __LINE__);
        return PGP_ERR_HIP;
    }
    return PGP_OK;
}

static int rccl_fail(const RcclApi& api, int rc, const char* what) {
    char msg[256];
    snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", what, rc, api.GetErrorString ? api.GetErrorString(rc) : "?");
    pgp_set_last_hip_error(hipErrorUnknown, msg, __FILE__, __LINE__);
    return PGP_ERR_HIP;
}

}  // namespace

struct pgp_comm {
    pgp_ctx* ctx = nullptr;
    int world = 1, rank = 0;
    int kind = 0;                       // 1 RCCL, 2 host call-backs
    RcclApi api;
    rccl_comm_t comm = nullptr;
    hipStream_t st_comm = nullptr;      // broadcasts travel here, beside the compute streams
    pgp_host_bcast_fn hb = nullptr;
    pgp_host_allreduce_fn har = nullptr;
    void* user = nullptr;
    void* stage = nullptr;              // pinned staging buffer of the host transport
    size_t stage_bytes = 0;
    double* agree = nullptr;            // 8 device doubles that exist before any call can fail to allocate (comm_agree_max)
    // host collectives of the restart / fold searches (pgp_comm_bcast_host, pgp_comm_allgather_host): device staging
    double* hostbuf = nullptr;
    size_t hostbuf_bytes = 0;
    std::vector<void*> hostbuf_retired;   // outgrown staging buffers: freed with the communicator, never beside a resident EP sweep
};

namespace {

static int comm_stage(pgp_comm* m, size_t bytes) {
    if (m->stage_bytes >= bytes) return PGP_OK;
    if (m->stage) (void)hipHostFree(m->stage);
    m->stage = nullptr; m->stage_bytes = 0;
    HIP_TRY(hipHostMalloc(&m->stage, bytes, hipHostMallocDefault));
    m->stage_bytes = bytes;
    return PGP_OK;
}

// Broadcast `bytes` at `buf` (device memory of this rank: the root's source, everybody else's destination).  Starts when
// `wait_ev` has fired (null: at once), records `done_ev` on the communication stream when buf holds the data.
static int comm_bcast(pgp_comm* m, void* buf, size_t bytes, int root, hipEvent_t wait_ev, hipEvent_t done_ev,
                      hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    // t0 / t1 (optional, timing events): recorded on the communication stream right before / after the transfer -- "enqueue to
    // complete" of this broadcast on this rank, the wait for the slowest peer included
    if (m->kind == 1) {
        if (wait_ev) HIP_TRY(hipStreamWaitEvent(m->st_comm, wait_ev, 0));
        if (t0) HIP_TRY(hipEventRecord(t0, m->st_comm));
        const int rc = m->api.Broadcast(buf, buf, bytes / sizeof(double), RCCL_DOUBLE, root, m->comm, m->st_comm);
        if (rc != 0) return rccl_fail(m->api, rc, "ncclBroadcast");
        if (t1) HIP_TRY(hipEventRecord(t1, m->st_comm));
        if (done_ev) HIP_TRY(hipEventRecord(done_ev, m->st_comm));
        return PGP_OK;
    }
    if (m->world == 1) {                             // nothing to move: order the consumers behind the producer
        if (wait_ev) HIP_TRY(hipStreamWaitEvent(m->st_comm, wait_ev, 0));
        if (done_ev) HIP_TRY(hipEventRecord(done_ev, m->st_comm));
        return PGP_OK;
    }
    // host transport: blocking, through pinned host memory (self-test transport)
    if (wait_ev) HIP_TRY(hipEventSynchronize(wait_ev));
    if (t0) HIP_TRY(hipEventRecord(t0, m->st_comm));
    CHK(comm_stage(m, bytes));
    if (m->rank == root) {
        HIP_TRY(hipMemcpyAsync(m->stage, buf, bytes, hipMemcpyDeviceToHost, m->st_comm));
        HIP_TRY(hipStreamSynchronize(m->st_comm));
    }
    if (m->world > 1) {
        const int rc = m->hb(m->user, m->stage, (int64_t)bytes, root);
        if (rc != 0) { pgp_set_last_hip_error(hipErrorUnknown, "host broadcast call-back failed", __FILE__, __LINE__); return PGP_ERR_HIP; }
    }
    if (m->rank != root) {
        HIP_TRY(hipMemcpyAsync(buf, m->stage, bytes, hipMemcpyHostToDevice, m->st_comm));
        HIP_TRY(hipStreamSynchronize(m->st_comm));          // the staging buffer is reused by the next call
    }
    if (t1) HIP_TRY(hipEventRecord(t1, m->st_comm));
    if (done_ev) HIP_TRY(hipEventRecord(done_ev, m->st_comm));
    return PGP_OK;
}

// In-place all-reduce of `count` doubles at `buf`, ordered on stream `st` (after everything queued there; the result is
// visible to whatever is queued on st afterwards).  op: 0 sum, 1 max.
static int comm_allreduce(pgp_comm* m, double* buf, size_t count, int op, hipStream_t st) {
    if (m->kind == 1) {
        const int rc = m->api.AllReduce(buf, buf, count, RCCL_DOUBLE, op ? RCCL_MAX : RCCL_SUM, m->comm, st);
        return rc != 0 ? rccl_fail(m->api, rc, "ncclAllReduce") : PGP_OK;
    }
    CHK(comm_stage(m, count * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(m->stage, buf, count * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (m->world > 1) {
        const int rc = m->har(m->user, (double*)m->stage, (int64_t)count, op);
        if (rc != 0) { pgp_set_last_hip_error(hipErrorUnknown, "host all-reduce call-back failed", __FILE__, __LINE__); return PGP_ERR_HIP; }
    }
    HIP_TRY(hipMemcpyAsync(buf, m->stage, count * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PGP_OK;
}

// max over the ranks of one host double, through the communicator's own 8-double buffer: every copy is ordered on `st` with the
// all-reduce (RCCL's all-reduce is asynchronous on a non-blocking stream: a plain hipMemcpy on the null stream does not wait for it)
static int comm_agree_max(pgp_comm* m, double mine, double* all, hipStream_t st) {
    *all = mine;
    if (m->world <= 1) return PGP_OK;
    if (m->kind != 1) {                              // host transport: no device round trip needed
        CHK(comm_stage(m, sizeof(double)));
        *(double*)m->stage = mine;
        const int rc = m->har(m->user, (double*)m->stage, 1, 1);
        if (rc != 0) { pgp_set_last_hip_error(hipErrorUnknown, "host all-reduce call-back failed", __FILE__, __LINE__); return PGP_ERR_HIP; }
        *all = *(double*)m->stage;
        return PGP_OK;
    }
    CHK(comm_stage(m, 2 * sizeof(double)));
    double* h = (double*)m->stage;
    h[0] = mine;
    HIP_TRY(hipMemcpyAsync(m->agree, h, sizeof(double), hipMemcpyHostToDevice, st));
    CHK(comm_allreduce(m, m->agree, 1, 1, st));
    HIP_TRY(hipMemcpyAsync(h + 1, m->agree, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *all = h[1];
    return PGP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// small kernels of the distributed epilogue
// ------------------------------------------------------------------------------------------------------------------
// the 128 right-hand-side rows of an unfactored panel: row 0 = r = y - m at the panel's columns, rows 1..127 = 0
__global__ __launch_bounds__(256) void panel_rhs_kernel(double* __restrict__ P, long ld, long row0, const double* __restrict__ y,
                                                        const double* __restrict__ mvec, long n, long col0) {
    const long c = blockIdx.x;                         // column of the panel
    const long gc = col0 + c;
    for (int i = threadIdx.x; i < 128; i += 256)
        P[row0 + i + c * ld] = (i == 0 && gc < n) ? y[gc] - mvec[gc] : 0.0;
}

// acc[e] += sum_c E(e, c) z[c * zs]   for e < rows  (E column-major rows x w: coalesced over e); one thread per row,
// eight independent loads in flight, fixed summation order
__global__ __launch_bounds__(256) void panel_matvec_kernel(const double* __restrict__ E, long ld, long rows, int w,
                                                           const double* __restrict__ z, long zs, double* __restrict__ acc) {
    __shared__ double zl[1024];
    for (int c = threadIdx.x; c < w; c += 256) zl[c] = z[(long)c * zs];
    __syncthreads();
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows) return;
    double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < w; c += 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = E[e + (long)(c + q) * ld];
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] = fma(v[q], zl[c + q], a8[q]);
    }
    acc[e] += ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}

// out[0] += sum_i log Ld(i, i) ; out[1] += sum_c z_c^2        (one workgroup; panels are accumulated in stream order)
__global__ __launch_bounds__(256) void panel_scalars_kernel(const double* __restrict__ Ld, int w, const double* __restrict__ z,
                                                            long zs, double* __restrict__ out) {
    __shared__ double ra[256], rb[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < w; i += 256) {
        a += log(Ld[i + (long)i * w]);
        const double zi = z[(long)i * zs];
        b = fma(zi, zi, b);
    }
    ra[threadIdx.x] = a; rb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { ra[threadIdx.x] += ra[threadIdx.x + s]; rb[threadIdx.x] += rb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] += ra[0]; out[1] += rb[0]; }
}

// red[np + 2] = 1 if this rank saw a non-positive pivot;  red[np + 3] = BIG - pivot (max over ranks = first bad pivot)
__global__ void pack_status_kernel(const int* __restrict__ info, double* __restrict__ red2) {
    const int v = info[0];
    red2[0] = v != 0 ? 1.0 : 0.0;
    red2[1] = v != 0 ? 1.0e9 - (double)v : 0.0;
}

// y[i] = s * x[i]
__global__ __launch_bounds__(256) void scale_vec_kernel(const double* __restrict__ x, double s, double* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = s * x[i];
}

// take a buffer out of a PoolScratch (ownership moves to a posterior handle)
static void scratch_release(PoolScratch& scr, void* p) {
    for (auto it = scr.held.begin(); it != scr.held.end(); ++it)
        if (it->second == p) { scr.held.erase(it); return; }
}

}  // namespace

// the distributed posterior of one rank (pgp_sharded_exact_fit -> pgp_sharded_predict)
struct pgp_sfactor {
    long n = 0, np = 0, ldp = 0;
    int w = 0, world = 1, me = 0, nloc = 0, npanel = 0, d = 0, dpad = 0;
    double* Bufs = nullptr; size_t bufs_bytes = 0;     // (nloc + 1) panel buffers; Y of local panel k = Bufs + k ldp w
    double* alpha = nullptr;                           // np
    double* XT = nullptr; size_t xt_bytes = 0;         // dpad x np scaled training coordinates
    CovSpec cs;
    double sn2 = 1.0, kss = 0.0;
};

// ------------------------------------------------------------------------------------------------------------------
extern "C" {

void pgp_sfactor_free(pgp_ctx* c, pgp_sfactor* f) { GateShared device_gate_hold(c);
    if (!f) return;
    if (c) (void)hipSetDevice(c->device);
    spool_give(c, f->bufs_bytes, f->Bufs);
    spool_give(c, (size_t)f->np * sizeof(double), f->alpha);
    spool_give(c, f->xt_bytes, f->XT);
    delete f;
}
int64_t pgp_sfactor_bytes(pgp_sfactor* f) {
    return f ? (int64_t)(f->bufs_bytes + (size_t)f->np * sizeof(double) + f->xt_bytes) : 0;
}

int pgp_comm_unique_id(const char* rccl_path, char* id_out) {
    if (!id_out) return -2;
    RcclApi api;
    CHK(rccl_load(rccl_path, api));
    rccl_uid id;
    memset(&id, 0, sizeof(id));
    const int rc = api.GetUniqueId(&id);
    if (rc != 0) return rccl_fail(api, rc, "ncclGetUniqueId");
    memcpy(id_out, id.internal, 128);
    return PGP_OK;                                   // the library handle stays open: RCCL keeps state behind the id
}

int pgp_comm_init_rccl(pgp_ctx* c, int world, int rank, const char* id, const char* rccl_path, pgp_comm** out) {
    if (!c) return -1;
    if (world < 1) return -2;
    if (rank < 0 || rank >= world) return -3;
    if (!id) return -4;
    if (!out) return -6;
    GateShared device_gate_hold(c);
    HIP_TRY(hipSetDevice(c->device));
    pgp_comm* m = new pgp_comm();
    m->ctx = c; m->world = world; m->rank = rank; m->kind = 1;
    int rc = rccl_load(rccl_path, m->api);
    if (rc != PGP_OK) { delete m; return rc; }
    rccl_uid uid;
    memcpy(uid.internal, id, 128);
    const int nrc = m->api.CommInitRank(&m->comm, world, uid, rank);
    if (nrc != 0) { rc = rccl_fail(m->api, nrc, "ncclCommInitRank"); delete m; return rc; }
    if (hipStreamCreateWithFlags(&m->st_comm, hipStreamNonBlocking) != hipSuccess) { (void)m->api.CommDestroy(m->comm); delete m; return PGP_ERR_HIP; }
    if (hipMalloc((void**)&m->agree, 8 * sizeof(double)) != hipSuccess || comm_stage(m, 64) != PGP_OK) {
        (void)hipStreamDestroy(m->st_comm); (void)m->api.CommDestroy(m->comm); if (m->agree) (void)hipFree(m->agree); delete m; return PGP_ERR_HIP;
    }
    *out = m;
    return PGP_OK;
}

int pgp_comm_init_host(pgp_ctx* c, int world, int rank, pgp_host_bcast_fn bcast, pgp_host_allreduce_fn allreduce, void* user,
                       pgp_comm** out) {
    if (world < 1) return -2;
    if (rank < 0 || rank >= world) return -3;
    if (world > 1 && (!bcast || !allreduce)) return -4;
    if (!out) return -7;
    pgp_comm* m = new pgp_comm();
    m->ctx = c; m->world = world; m->rank = rank; m->kind = 2;
    m->hb = bcast; m->har = allreduce; m->user = user;
    if (c) {                                         // ctx == NULL: a communicator for the host collectives only (no device is touched)
        if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&m->st_comm, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError(); delete m; return PGP_ERR_HIP;
        }
    }
    *out = m;
    return PGP_OK;
}

// ---- host collectives on the communicator (the restart / fold searches of pygps_amd/opt.py, valid.py: Core/opt.py:301-327 sharded) ----
// Buffers are HOST memory.  RCCL transport: staged through a device buffer of the communicator on its own stream; host
// transport: the call-backs, directly.  Every rank calls with the same counts.
static int comm_hostbuf(pgp_comm* m, size_t bytes) {
    if (m->hostbuf_bytes >= bytes) return PGP_OK;
    if (m->hostbuf) m->hostbuf_retired.push_back(m->hostbuf);      // grow-only (doubling): no hipFree on this path
    m->hostbuf = nullptr; m->hostbuf_bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 16);
    size_t cap = (size_t)1 << 16;
    while (cap < want) cap <<= 1;
    HIP_TRY(hipMalloc((void**)&m->hostbuf, cap));
    m->hostbuf_bytes = cap;
    return PGP_OK;
}
static int host_stage(pgp_comm* m, size_t bytes) {        // plain (pageable is fine) host staging for a ctx-less communicator
    if (m->ctx) return comm_stage(m, bytes);
    if (m->stage_bytes >= bytes) return PGP_OK;
    free(m->stage);
    m->stage = malloc(bytes); m->stage_bytes = m->stage ? bytes : 0;
    return m->stage ? PGP_OK : PGP_ERR_HIP;
}

int pgp_comm_bcast_host(pgp_comm* m, double* buf, int64_t count, int root) {
    if (!m) return -1;
    if (!buf && count > 0) return -2;
    if (count < 0) return -3;
    if (root < 0 || root >= m->world) return -4;
    if (count == 0 || (m->world == 1 && m->kind != 1)) return PGP_OK;
    const size_t bytes = (size_t)count * sizeof(double);
    if (m->kind == 1) {
        GateShared device_gate_hold(m->ctx);             // these touch the runtime: never beside a resident EP sweep (ctx.h DeviceGate)
        HIP_TRY(hipSetDevice(m->ctx->device));
        CHK(comm_hostbuf(m, bytes));
        if (m->rank == root) HIP_TRY(hipMemcpyAsync(m->hostbuf, buf, bytes, hipMemcpyHostToDevice, m->st_comm));
        const int rc = m->api.Broadcast(m->hostbuf, m->hostbuf, (size_t)count, RCCL_DOUBLE, root, m->comm, m->st_comm);
        if (rc != 0) return rccl_fail(m->api, rc, "ncclBroadcast");
        if (m->rank != root) HIP_TRY(hipMemcpyAsync(buf, m->hostbuf, bytes, hipMemcpyDeviceToHost, m->st_comm));
        HIP_TRY(hipStreamSynchronize(m->st_comm));
        return PGP_OK;
    }
    const int rc = m->hb(m->user, buf, (int64_t)bytes, root);
    if (rc != 0) { pgp_set_last_hip_error(hipErrorUnknown, "host broadcast call-back failed", __FILE__, __LINE__); return PGP_ERR_HIP; }
    return PGP_OK;
}

int pgp_comm_allreduce_host(pgp_comm* m, double* buf, int64_t count, int op) {
    if (!m) return -1;
    if (!buf && count > 0) return -2;
    if (count < 0) return -3;
    if (op != 0 && op != 1) return -4;
    if (count == 0 || (m->world == 1 && m->kind != 1)) return PGP_OK;
    const size_t bytes = (size_t)count * sizeof(double);
    if (m->kind == 1) {
        GateShared device_gate_hold(m->ctx);
        HIP_TRY(hipSetDevice(m->ctx->device));
        CHK(comm_hostbuf(m, bytes));
        HIP_TRY(hipMemcpyAsync(m->hostbuf, buf, bytes, hipMemcpyHostToDevice, m->st_comm));
        CHK(comm_allreduce(m, m->hostbuf, (size_t)count, op, m->st_comm));
        HIP_TRY(hipMemcpyAsync(buf, m->hostbuf, bytes, hipMemcpyDeviceToHost, m->st_comm));
        HIP_TRY(hipStreamSynchronize(m->st_comm));
        return PGP_OK;
    }
    const int rc = m->har(m->user, buf, count, op);
    if (rc != 0) { pgp_set_last_hip_error(hipErrorUnknown, "host all-reduce call-back failed", __FILE__, __LINE__); return PGP_ERR_HIP; }
    return PGP_OK;
}

// recv: world * count doubles, rank r's `count` doubles at recv + r * count.  Host transport: an all-reduce (sum) of a buffer
// that is zero outside the rank's own slot -- exact (x + 0), infinities and NaNs of the slot's owner included.
int pgp_comm_allgather_host(pgp_comm* m, const double* send, int64_t count, double* recv) {
    if (!m) return -1;
    if ((!send || !recv) && count > 0) return -2;
    if (count < 0) return -3;
    if (count == 0) return PGP_OK;
    const size_t bytes = (size_t)count * sizeof(double);
    if (m->kind == 1) {
        GateShared device_gate_hold(m->ctx);
        HIP_TRY(hipSetDevice(m->ctx->device));
        CHK(comm_hostbuf(m, bytes * (size_t)(m->world + 1)));
        double* sd = m->hostbuf + (size_t)m->world * (size_t)count;
        HIP_TRY(hipMemcpyAsync(sd, send, bytes, hipMemcpyHostToDevice, m->st_comm));
        const int rc = m->api.AllGather(sd, m->hostbuf, (size_t)count, RCCL_DOUBLE, m->comm, m->st_comm);
        if (rc != 0) return rccl_fail(m->api, rc, "ncclAllGather");
        HIP_TRY(hipMemcpyAsync(recv, m->hostbuf, bytes * (size_t)m->world, hipMemcpyDeviceToHost, m->st_comm));
        HIP_TRY(hipStreamSynchronize(m->st_comm));
        return PGP_OK;
    }
    if (m->world == 1) { if (recv != send) memmove(recv, send, bytes); return PGP_OK; }
    CHK(host_stage(m, bytes * (size_t)m->world));
    double* st = (double*)m->stage;
    memset(st, 0, bytes * (size_t)m->world);
    memcpy(st + (size_t)m->rank * (size_t)count, send, bytes);
    const int rc = m->har(m->user, st, count * (int64_t)m->world, 0);
    if (rc != 0) { pgp_set_last_hip_error(hipErrorUnknown, "host all-reduce call-back failed", __FILE__, __LINE__); return PGP_ERR_HIP; }
    memcpy(recv, st, bytes * (size_t)m->world);
    return PGP_OK;
}

void pgp_comm_free(pgp_comm* m) {
    if (!m) return;
    GateShared device_gate_hold(m->ctx);
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->st_comm) { (void)hipStreamSynchronize(m->st_comm); (void)hipStreamDestroy(m->st_comm); }
    if (m->kind == 1 && m->comm) (void)m->api.CommDestroy(m->comm);
    if (m->stage) { if (m->ctx) (void)hipHostFree(m->stage); else free(m->stage); }
    if (m->agree) (void)hipFree(m->agree);
    if (m->hostbuf) (void)hipFree(m->hostbuf);
    for (void* p : m->hostbuf_retired) (void)hipFree(p);
    delete m;
}

int pgp_comm_world(pgp_comm* m) { return m ? m->world : 0; }
int pgp_comm_rank(pgp_comm* m) { return m ? m->rank : -1; }

// Exact.evaluate over the ranks of `comm`.  Every rank passes the same data (pgp_set_data) and arguments and receives the
// same alpha / nlZ / dnlZ.  Status as pgp_exact_fit: > 0 = first non-positive pivot, identical on every rank.
// timings_out (optional, 10): ms of assembly, sweep (+ E E' under it), epilogue (alpha, gradient, collectives), total; then the
// device bytes this call held at its peak (panels, receive buffers, strips of B^-1, scratch) and the bytes the handle keeps.
// L_out (optional, (n,n) row-major, zero-filled by the caller): this rank's columns of the factor in post.L's form (upper R =
// L', Core/inf.py:362): R(j, i) = L(i, j) for the owned columns j; the sum over the ranks is the whole factor.
// factor_out (optional): the rank's part of the distributed posterior for pgp_sharded_predict.
int pgp_sharded_exact_fit(pgp_ctx* c, pgp_comm* m, int kind, const double* covhyp, int ncov, int para, int flags, double log_sn,
                          const double* mvec, const double* dm, int nmean, int want, double* alpha_out, double* nlZ_out,
                          double* dnlZ_out, double* timings_out, double* L_out, pgp_sfactor** factor_out) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (!m || m->ctx != c) return -2;
    if (c->n <= 0) return -1;
    if (!covhyp) return -4;
    if (want < 1 || want > 3) return -12;
    HIP_TRY(hipSetDevice(c->device));
    const long n = c->n, d = c->d;
    const int world = m->world, me = m->rank;
    const int w = c->nb_outer > 0 ? std::min(c->nb_outer, 8) * 128 : (n >= 12288 ? 1024 : 512);
    const long np = round_up(n, w);                  // padded with identity rows / columns, like the single-GPU path
    const int npanel = (int)(np / w);
    const long ldp = np + 128;                       // rows of a panel buffer (P_j and Y_p alike)
    const size_t pbytes = (size_t)ldp * w * sizeof(double);
    std::vector<int> mine;                           // owned panels, ascending
    for (int p = me; p < npanel; p += world) mine.push_back(p);
    const int nloc = (int)mine.size();
    CovSpec cp;
    { const int rc = make_spec(c, kind, covhyp, ncov, para, flags, -1, d, cp); if (rc != PGP_OK) return rc == -11 ? -5 : rc; }
    const double sn2 = exp(2.0 * log_sn);
    const bool grad = want >= 3;
    const int dpad = c->dpad;

    // ---- workspace (pooled: an optimiser calls with identical shapes hundreds of times) --------------------------
    // strips of B^-1: local strip k = columns of panel mine[k], rows >= mine[k] w, leading dimension np - mine[k] w, one
    // behind the other (GemmArgs::batch_dldc / batch_sC2 address them from the batch index)
    std::vector<size_t> strip_off(nloc + 1, 0);
    for (int k = 0; k < nloc; ++k) strip_off[k + 1] = strip_off[k] + (size_t)w * (size_t)(np - (long)mine[k] * w);
    long hblocks = 0;
    for (int k = 0; k < nloc; ++k) hblocks += hadamard_block_count(np, (long)mine[k] * w / 64, w / 64);
    PoolScratch scr(c);
    double *Bufs = nullptr, *Ld = nullptr, *R[2] = {nullptr, nullptr}, *Sbuf = nullptr, *XT = nullptr, *red = nullptr,
           *partial = nullptr, *mdev = nullptr, *gout = nullptr, *adev = nullptr;
    size_t held_bytes = 0;
    int arc = PGP_OK;                                // the first allocation failure; agreed on with the other ranks below
    auto take = [&](double** out, size_t bytes) {
        if (arc == PGP_OK) { arc = scr.alloc(out, bytes); if (arc == PGP_OK) held_bytes += bytes; }
    };
    take(&red, (size_t)(np + 8) * sizeof(double));
    take(&Bufs, (size_t)(nloc + 1) * pbytes);
    take(&Ld, (size_t)std::max(nloc, 1) * w * w * sizeof(double));
    if (world > 1) { take(&R[0], pbytes); take(&R[1], pbytes); }
    const size_t xt_bytes = (size_t)dpad * np * sizeof(double);
    take(&XT, xt_bytes);
    take(&mdev, (size_t)np * sizeof(double));
    take(&gout, (size_t)(ncov + 8) * sizeof(double));
    if (factor_out) take(&adev, (size_t)np * sizeof(double));
    if (grad) {
        take(&Sbuf, std::max<size_t>(strip_off[nloc], 1) * sizeof(double));
        take(&partial, (size_t)(hblocks * (ncov + 1) + hadamard_prep_count(np)) * sizeof(double));
    }
    if (world > 1) {                                 // a rank that ran out of memory must not leave the others in a collective
        double any = 0.0;
        CHK(comm_agree_max(m, arc != PGP_OK ? 1.0 : 0.0, &any, c->st));
        if (arc != PGP_OK) return arc;
        if (any != 0.0) { pgp_set_last_hip_error(hipErrorOutOfMemory, "another rank of the sharded fit ran out of device memory", __FILE__, __LINE__); return PGP_ERR_HIP; }
    } else if (arc != PGP_OK) return arc;
    double kss = 0.0;
    if (factor_out) CHK(cov_point_value(c, cp, 2, &kss));
    auto buf = [&](int k) { return Bufs + (size_t)k * ldp * w; };      // P of local panel k = buf(k + 1); Y of local panel k = buf(k)
    hipStream_t main = c->st, pan = c->st2;
    const bool pan_solve = world > 1;                // the owner's S(p+1) right behind D(p+1): its broadcast is on the critical path
    const int nev = 3 * npanel + 8;
    while ((int)c->la_ev.size() < nev) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->la_ev.push_back(e);
    }
    auto EV_S = [&](int p) { return c->la_ev[p]; };                    // Y_p produced (owner)
    auto EV_Y = [&](int p) { return c->la_ev[npanel + p]; };           // Y_p present on this rank
    auto EV_F = [&](int p) { return c->la_ev[2 * npanel + p]; };       // TU(p) has left the stream (its receive buffer is free)
    hipEvent_t ev_stage[2] = {c->la_ev[3 * npanel], c->la_ev[3 * npanel + 1]};
    hipEvent_t ev_a = c->la_ev[3 * npanel + 2], ev_d = c->la_ev[3 * npanel + 3];

    // ---- inputs ----------------------------------------------------------------------------------------------------
    HIP_TRY(hipEventRecord(c->ev[0], main));
    HIP_TRY(hipMemsetAsync(c->info_dev, 0, sizeof(int), main));
    HIP_TRY(hipMemsetAsync(mdev, 0, np * sizeof(double), main));
    if (mvec) HIP_TRY(hipMemcpyAsync(mdev, mvec, n * sizeof(double), hipMemcpyHostToDevice, main));
    HIP_TRY(hipMemsetAsync(red, 0, (np + 8) * sizeof(double), main));
    CHK(upload_scaled(c, c->x_dev, n, d, cp.scale, XT, np, dpad, c->scale_dev));
    if (grad && strip_off[nloc]) HIP_TRY(hipMemsetAsync(Sbuf, 0, strip_off[nloc] * sizeof(double), main));
    if (L_out) HIP_TRY(hipMemsetAsync(Ld, 0, (size_t)std::max(nloc, 1) * w * w * sizeof(double), main));   // strict upper parts: exact zeros
    // ---- assembly: every rank builds ITS column panels of B = K/sn2 + I straight from the coordinates -------------------
    for (int k = 0; k < nloc; ++k) {
        const long col0 = (long)mine[k] * w;
        { ProfScope ps(c, PC_ASSEMBLE, 0.0, 8.0 * (double)(np - col0) * w);
          CHK(cov_factor_panel_launch(XT, np, n, np, dpad, cp, 1.0 / sn2, col0, w, buf(k + 1), ldp, main)); }
        hipLaunchKernelGGL(panel_rhs_kernel, dim3(w), dim3(256), 0, main, buf(k + 1), ldp, np - col0, c->y_dev, mdev, n, col0);
    }
    HIP_TRY(hipEventRecord(c->ev[1], main));

    // ---- the sweep ---------------------------------------------------------------------------------------------------
    auto Yptr = [&](int p) -> double* { return (p % world == me) ? buf(p / world) : R[p & 1]; };
    auto factor = [&](int p, hipStream_t st, hipEvent_t staged) -> int {          // D(p) on the owner
        const int k = p / world;
        return diag_block_factor(c, buf(k + 1), ldp, w, Ld + (size_t)k * w * w, w, buf(k) + (ldp - w), ldp, p * w, st, staged);
    };
    auto solve = [&](int p, hipStream_t st) -> int {                               // S(p): Y_p = P_p[w:] E_D
        const int k = p / world;
        GemmArgs g{};
        g.A = buf(k + 1) + w; g.lda = ldp; g.a_kc = 0;
        g.B = c->Dk + w; g.ldb = 2L * w; g.b_kc = 1;
        g.C = buf(k); g.ldc = ldp;
        g.M = (int)(ldp - w); g.N = w; g.K = w; g.alpha = 1.0; g.beta = 0.0; g.kmode = KM_LT_J; g.koff = 0;
        const long t128 = (long)(g.M / 128) * (w / 128);
        g.tile = c->s_tile ? c->s_tile : ((t128 < c->small_tile_below || t128 < 512) ? 64 : 128);
        g.rev_cols = 1;
        const double nt = (double)(w / g.tile);
        g.flops = 2.0 * (double)g.M * g.tile * g.tile * nt * (nt + 1.0) * 0.5;
        return gemm_prof(c, PC_GEMM_SOLVE, g, st);
    };
    // TU(p) on the local panels k0 .. k0 + nb - 1 (global j = mine[k], all > p): one batched launch
    auto update = [&](int p, int k0, int nb, hipStream_t st) -> int {
        if (nb <= 0) return PGP_OK;
        const int j = mine[k0];
        GemmArgs g{};
        g.A = Yptr(p) + (long)(j - p - 1) * w; g.lda = ldp; g.a_kc = 0;
        g.B = g.A; g.ldb = ldp; g.b_kc = 0;
        g.C = buf(k0 + 1); g.ldc = ldp;
        g.M = (int)(ldp - (long)(j - p - 1) * w); g.N = w; g.K = w;
        g.alpha = -1.0; g.beta = 1.0; g.tri = 1; g.tri_off = 0; g.mask_diag = 1; g.kmode = KM_FULL;
        g.zero_from = (int)(ldp - (long)(j - p) * w);                  // the E rows of block p: first touch
        g.batch = nb; g.sA = (long)world * w; g.sB = g.sA; g.sC = ldp * (long)w; g.batch_dm = nb > 1 ? world * w : 0;
        double fl = 0.0; long t128 = 0;
        for (int z = 0; z < nb; ++z) {
            const double Mz = (double)g.M - (double)z * world * w;
            fl += 2.0 * w * (Mz * w - 0.5 * (double)w * w);
            t128 += (long)(Mz / 128) * (w / 128) - (long)(w / 128) * (w / 128 - 1) / 2;
        }
        g.tile = t128 < c->small_tile_below ? 64 : 128;
        g.flops = fl;
        if (nb > 1) { CHK(batch_tile_list(c, g.M / g.tile, w / g.tile, nb, world * w / g.tile, &g.order, &g.norder)); g.order_z = 1; }
        return gemm_prof(c, PC_GEMM_TRAIL, g, st);
    };
    // the rank's strips of B^-1 += E_p E_p' : strips j = mine[0 .. nb) <= p, ONE batched launch.  Product z: rows and
    // columns of E_p from j_z w on (M shrinks by world w per strip), k clipped to the triangle of the diagonal block
    auto eet = [&](int p, hipStream_t st) -> int {
        int nb = 0;
        while (nb < nloc && mine[nb] <= p) ++nb;
        if (nb == 0) return PGP_OK;
        const long rows = (long)(p + 1) * w;
        const long j0 = (long)mine[0] * w;
        const double* E = Yptr(p) + (ldp - rows);                                  // E_p: logical row i at E[i], ld = ldp
        GemmArgs g{};
        g.A = E + j0; g.lda = ldp; g.a_kc = 0;
        g.B = g.A; g.ldb = ldp; g.b_kc = 0;
        g.C = Sbuf; g.ldc = np - j0;
        g.M = (int)(rows - j0); g.N = w; g.K = w; g.alpha = 1.0; g.beta = 1.0;
        g.tri = 1; g.tri_off = 0; g.mask_diag = 1;
        g.kmode = KM_GE_I; g.koff = (int)(j0 - (long)p * w);
        g.batch = nb; g.sA = (long)world * w; g.sB = g.sA; g.sC = (long)w * (np - j0);
        if (nb > 1) { g.batch_dm = world * w; g.batch_dk = world * w; g.batch_dldc = world * w; g.batch_sC2 = (long)w * world * w; }
        double fl = 0.0; long t128 = 0;
        for (int z = 0; z < nb; ++z) {
            const double Mz = (double)g.M - (double)z * world * w;
            fl += 2.0 * w * (Mz * w - 0.5 * (double)w * w);
            t128 += (long)(Mz / 128) * (w / 128) - (long)(w / 128) * (w / 128 - 1) / 2;
        }
        g.tile = t128 < c->small_tile_below ? 64 : 128;
        g.flops = fl;
        if (nb > 1) { CHK(batch_tile_list(c, g.M / g.tile, w / g.tile, nb, world * w / g.tile, &g.order, &g.norder)); g.order_z = 1; }
        return gemm_prof(c, PC_GEMM_LAUUM, g, st);
    };

    // A launch failure on one rank must not leave the others inside a collective: the rank goes on taking part in the
    // broadcasts (its compute is skipped), its flag rides in the first all-reduce and EVERY rank returns an error.
    int poison = PGP_OK;
    auto body = [&](int p) -> int {                   // everything of step p that follows the arrival of Y_p
        const int owner = p % world;
        if (p + 1 >= npanel) return grad ? eet(p, main) : PGP_OK;
        const int nxt = p + 1;
        int k_first = 0;                              // local panels with global index > p
        while (k_first < nloc && mine[k_first] <= p) ++k_first;
        if (nxt % world == me) {
            // look-ahead: the next panel first, then its factorisation on the panel stream beside the rest of the step
            const int kn = nxt / world;              // == k_first
            CHK(update(p, kn, 1, main));                                           // TU_a
            HIP_TRY(hipEventRecord(ev_a, main));
            HIP_TRY(hipStreamWaitEvent(pan, ev_a, 0));
            const bool lf = c->leaf_first != 0;
            CHK(factor(nxt, pan, lf ? ev_stage[p & 1] : nullptr));                 // D(p+1)
            if (lf) HIP_TRY(hipStreamWaitEvent(main, ev_stage[p & 1], 0));
            if (pan_solve) {
                CHK(solve(nxt, pan));                                              // S(p+1) at once: the others wait for it
                HIP_TRY(hipEventRecord(EV_S(nxt), pan));
            } else HIP_TRY(hipEventRecord(ev_d, pan));
            CHK(update(p, kn + 1, nloc - kn - 1, main));                           // TU_b: the rest, one batched launch
            if (grad) CHK(eet(p, main));
            if (!pan_solve) {
                HIP_TRY(hipStreamWaitEvent(main, ev_d, 0));
                CHK(solve(nxt, main));                                             // S(p+1)
                HIP_TRY(hipEventRecord(EV_S(nxt), main));
            }
        } else {
            CHK(update(p, k_first, nloc - k_first, main));                         // TU(p): every owned panel beyond p
            if (grad) CHK(eet(p, main));
        }
        (void)owner;
        HIP_TRY(hipEventRecord(EV_F(p), main));
        return PGP_OK;
    };
    // timers of the first multi-GPU runs (timings_out[6..9]): per panel, how long the compute stream stalled for Y_p and how long
    // its broadcast took on this rank from enqueue to complete
    const bool timers = timings_out != nullptr && world > 1;
    if (timers) {
        while ((int)c->tm_ev.size() < 4 * npanel) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            c->tm_ev.push_back(e);
        }
    }
    if (0 % world == me) {                           // panel 0 on its owner, no look-ahead to hide behind
        poison = factor(0, main, nullptr);
        if (poison == PGP_OK) poison = solve(0, main);
        if (poison == PGP_OK && hipEventRecord(EV_S(0), main) != hipSuccess) poison = PGP_ERR_HIP;
    }
    for (int p = 0; p < npanel; ++p) {
        const int owner = p % world;
        // ---- Y_p to every rank (a never-recorded event does not block: a poisoned owner still broadcasts) ----
        hipEvent_t wait = owner == me ? EV_S(p) : (p >= 2 && world > 1 ? EV_F(p - 2) : nullptr);
        const int brc = comm_bcast(m, Yptr(p), pbytes, owner, wait, EV_Y(p), timers ? c->tm_ev[4 * p] : nullptr,
                                   timers ? c->tm_ev[4 * p + 1] : nullptr);
        if (brc != PGP_OK) { (void)hipDeviceSynchronize(); return brc; }           // the transport itself failed: nothing to agree over
        if (poison != PGP_OK) continue;
        if (timers) (void)hipEventRecord(c->tm_ev[4 * p + 2], main);               // the compute stream is ready for Y_p ...
        if (hipStreamWaitEvent(main, EV_Y(p), 0) != hipSuccess) { poison = PGP_ERR_HIP; continue; }
        if (timers) (void)hipEventRecord(c->tm_ev[4 * p + 3], main);               // ... and has it: the difference is the stall
        poison = body(p);
    }
    if (poison != PGP_OK) {
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        if (world == 1) return poison;
        const double one = 1.0;
        (void)hipMemcpy(red + np + 4, &one, sizeof(double), hipMemcpyHostToDevice);
    }
    // a poisoned rank must reach the all-reduce that carries its flag: nothing before it may return (a STICKY device fault makes
    // every HIP call fail, the all-reduce included -- then only ncclCommAbort below can release the peers)
    if (hipEventRecord(c->ev[2], main) != hipSuccess && poison == PGP_OK) {
        if (world == 1) return PGP_ERR_HIP;
        poison = PGP_ERR_HIP;
        (void)hipGetLastError();
        const double one = 1.0;
        (void)hipMemcpy(red + np + 4, &one, sizeof(double), hipMemcpyHostToDevice);
    }

    // ---- epilogue: alpha, log det, z'z from this rank's panels; ONE all-reduce ------------------------------------------
    if (poison == PGP_OK) {
        for (int k = 0; k < nloc; ++k) {
            const int p = mine[k];
            const long rows = (long)(p + 1) * w;
            const double* Y = buf(k);
            const double* z = Y + (np - rows);            // logical row np (r = y - m after the forward substitution)
            { ProfScope ps(c, PC_SMALL, 0.0, 8.0 * (double)rows * w);
              hipLaunchKernelGGL(panel_matvec_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, main, Y + (ldp - rows), ldp,
                                 rows, w, z, ldp, red); }
            hipLaunchKernelGGL(panel_scalars_kernel, dim3(1), dim3(256), 0, main, Ld + (size_t)k * w * w, w, z, ldp, red + np);
        }
        hipLaunchKernelGGL(pack_status_kernel, dim3(1), dim3(1), 0, main, c->info_dev, red + np + 2);
        if (hipGetLastError() != hipSuccess) {           // as inside the sweep: the others learn it from the all-reduce
            if (world == 1) return PGP_ERR_HIP;
            poison = PGP_ERR_HIP;
            (void)hipDeviceSynchronize();
            const double one = 1.0;
            (void)hipMemcpy(red + np + 4, &one, sizeof(double), hipMemcpyHostToDevice);
        }
    }
    {
        const int arc2 = comm_allreduce(m, red, (size_t)np + 5, 0, main);
        if (arc2 != PGP_OK) {                        // the flag cannot travel (sticky fault / transport down): abort the communicator so that
            if (m->kind == 1 && m->api.CommAbort && m->comm) { (void)m->api.CommAbort(m->comm); m->comm = nullptr; }   // the peers' collectives return
            return poison != PGP_OK ? poison : arc2;
        }
    }
    std::vector<double> head(8, 0.0);
    HIP_TRY(hipMemcpyAsync(head.data(), red + np, 5 * sizeof(double), hipMemcpyDeviceToHost, main));
    HIP_TRY(hipStreamSynchronize(main));
    if (head[4] != 0.0) {                            // a rank failed inside the sweep: everybody leaves, with its own code or this one
        (void)hipDeviceSynchronize();
        if (poison != PGP_OK) return poison;
        pgp_set_last_hip_error(hipErrorUnknown, "another rank of the sharded fit failed inside the sweep", __FILE__, __LINE__);
        return PGP_ERR_HIP;
    }
    if (head[2] != 0.0) {                            // a non-positive pivot somewhere: every rank learns the first one
        CHK(comm_allreduce(m, red + np + 3, 1, 1, main));
        double v = 0.0;
        HIP_TRY(hipMemcpyAsync(&v, red + np + 3, sizeof(double), hipMemcpyDeviceToHost, main));
        HIP_TRY(hipStreamSynchronize(main));
        (void)hipDeviceSynchronize();
        const long piv = (long)llround(1.0e9 - v);
        return (int)(piv > n ? n : (piv < 1 ? 1 : piv));
    }
    // alpha = (sum of the partials) / sn2 ; the gradient reduce over the tile rows of the rank's strips (the strips are
    // COMPLETE entries of B^-1: the true alpha, every (row, column) pair on exactly one rank), then one all-reduce
    hipLaunchKernelGGL(scale_vec_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, main, red, 1.0 / sn2, red, np);
    if (grad) {
        double* mu = partial + hblocks * (long)(ncov + 1);
        CHK(hadamard_prepare_launch(XT, np, n, np, dpad, cp, mu, main));
        long done = 0;
        for (int k = 0; k < nloc; ++k) {
            const long j0 = (long)mine[k] * w, ldk = np - j0;
            // Binv[r * ldb + c] (global r in the strip's columns, c >= r) = strip[(r - j0) * ldk + (c - j0)]
            const double* Bv = Sbuf + strip_off[k] - j0 * ldk - j0;
            const long nb_ = hadamard_block_count(np, j0 / 64, w / 64);
            { ProfScope ps(c, PC_HADAMARD, 0.0, 8.0 * (double)w * ldk + 8.0 * (double)n * d);
              CHK(hadamard_partial_launch(XT, np, n, np, dpad, cp, ncov, sn2, Bv, ldk, red, nullptr, partial + done * (ncov + 1), mu,
                                          j0 / 64, w / 64, main)); }
            done += nb_;
        }
        if (done > 0) CHK(hadamard_final_launch(partial, done, ncov, gout, main));
        else HIP_TRY(hipMemsetAsync(gout, 0, (size_t)(ncov + 1) * sizeof(double), main));
        CHK(comm_allreduce(m, gout, (size_t)ncov + 1, 0, main));
    }
    if (factor_out) HIP_TRY(hipMemcpyAsync(adev, red, (size_t)np * sizeof(double), hipMemcpyDeviceToDevice, main));
    HIP_TRY(hipEventRecord(c->ev[3], main));
    std::vector<double> alpha_h(n), g_h(ncov + 1, 0.0);
    HIP_TRY(hipMemcpyAsync(alpha_h.data(), red, n * sizeof(double), hipMemcpyDeviceToHost, main));
    if (grad) HIP_TRY(hipMemcpyAsync(g_h.data(), gout, (ncov + 1) * sizeof(double), hipMemcpyDeviceToHost, main));
    HIP_TRY(hipStreamSynchronize(main));
    HIP_TRY(hipStreamSynchronize(pan));
    if (m->st_comm) HIP_TRY(hipStreamSynchronize(m->st_comm));
    if (c->prof) prof_collect(c);
    if (timings_out) {
        float a = 0, b = 0, e = 0, t = 0;
        (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]); (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
        (void)hipEventElapsedTime(&e, c->ev[2], c->ev[3]); (void)hipEventElapsedTime(&t, c->ev[0], c->ev[3]);
        timings_out[0] = a; timings_out[1] = b; timings_out[2] = e; timings_out[3] = t;
        timings_out[4] = (double)(held_bytes + (size_t)(np + 8) * sizeof(double));
        timings_out[5] = factor_out ? (double)((size_t)(nloc + 1) * pbytes + xt_bytes + (size_t)np * sizeof(double)) : 0.0;
        double wait_ms = 0.0, bc_ms = 0.0, bc_max = 0.0;
        if (timers)
            for (int p = 0; p < npanel; ++p) {
                float wv = 0, bv = 0;
                if (hipEventElapsedTime(&wv, c->tm_ev[4 * p + 2], c->tm_ev[4 * p + 3]) == hipSuccess) wait_ms += wv;
                if (hipEventElapsedTime(&bv, c->tm_ev[4 * p], c->tm_ev[4 * p + 1]) == hipSuccess) { bc_ms += bv; bc_max = std::max(bc_max, (double)bv); }
            }
        (void)hipGetLastError();
        timings_out[6] = wait_ms;                                  // compute stream stalled waiting for a panel (sum over the panels)
        timings_out[7] = bc_ms;                                    // broadcasts, enqueue -> complete on this rank (sum; they overlap compute)
        timings_out[8] = world > 1 ? (double)npanel * (double)pbytes : 0.0;     // bytes this rank sent or received in broadcasts
        timings_out[9] = bc_max;                                   // the slowest single broadcast
    }
    if (L_out) {                                     // owned columns of L: the diagonal block from Ld, the rows below from Y
        for (int k = 0; k < nloc; ++k) {
            const long col0 = (long)mine[k] * w;
            if (col0 >= n) break;
            const long ncol = std::min<long>(w, n - col0);
            // column-major lower L(i, j) at L_out[j * n + i]  (== row-major upper R(j, i))
            HIP_TRY(hipMemcpy2D(L_out + col0 * n + col0, n * sizeof(double), Ld + (size_t)k * w * w, w * sizeof(double),
                                ncol * sizeof(double), ncol, hipMemcpyDeviceToHost));
            const long below = n - (col0 + w);
            if (below > 0)
                HIP_TRY(hipMemcpy2D(L_out + col0 * n + col0 + w, n * sizeof(double), buf(k), ldp * sizeof(double),
                                    below * sizeof(double), ncol, hipMemcpyDeviceToHost));
        }
    }
    if (alpha_out) memcpy(alpha_out, alpha_h.data(), n * sizeof(double));
    if (want >= 2 && nlZ_out) *nlZ_out = 0.5 * head[1] / sn2 + head[0] + 0.5 * (double)n * log(2.0 * M_PI * sn2);   // inf.py:370
    if (grad && dnlZ_out) {
        for (int i = 0; i < nmean; ++i) {                 // Core/inf.py:378-381
            double s = 0.0;
            for (long j = 0; j < n; ++j) s += dm[(long)i * n + j] * alpha_h[j];
            dnlZ_out[i] = -s;
        }
        for (int h = 0; h < ncov; ++h) dnlZ_out[nmean + h] = 0.5 * g_h[h];             // inf.py:377
        dnlZ_out[nmean + ncov] = g_h[ncov];                                            // inf.py:374
    }
    if (factor_out) {                                // the rank's panels, alpha and the coordinates move into the handle
        pgp_sfactor* f = new pgp_sfactor();
        f->n = n; f->np = np; f->ldp = ldp; f->w = w; f->world = world; f->me = me; f->nloc = nloc; f->npanel = npanel;
        f->d = (int)d; f->dpad = dpad; f->cs = cp; f->sn2 = sn2; f->kss = kss;
        f->Bufs = Bufs; f->bufs_bytes = (size_t)(nloc + 1) * pbytes; scratch_release(scr, Bufs);
        f->alpha = adev; scratch_release(scr, adev);
        f->XT = XT; f->xt_bytes = xt_bytes; scratch_release(scr, XT);
        *factor_out = f;
    }
    return PGP_OK;
}

// GP.predict (Core/gp.py:395-417) on the distributed posterior: every rank calls with the same test points and gets the same
// fmu = ms + Ks' alpha and fs2 = max(kss - colsum(V^2), 0), V = L^-1 (sW o Ks) = E' Ks / sn: rank r forms the rows of V that
// belong to ITS panels (V_p = E_p' Ks, one clipped MFMA product per owned panel), sums their squares per test point, and one
// all-reduce of ns doubles per batch finishes fs2.  No triangular solve: E = L^-T came out of the sweep.
int pgp_sharded_predict(pgp_ctx* c, pgp_comm* m, pgp_sfactor* f, const double* xs, int64_t ns, const double* ms, double* fmu,
                        double* fs2) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (!m || m->ctx != c) return -2;
    if (!f || f->world != m->world || f->me != m->rank) return -3;
    if (!xs) return -4;
    if (ns <= 0) return -5;
    if (!fmu) return -7;
    if (!fs2) return -8;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->st;
    const long np = f->np, n = f->n, ldp = f->ldp;
    const int w = f->w, d = f->d, dpad = f->dpad;
    const long NSB = predict_batch_points(c->predict_batch, ns, np);
    const long ldc = NSB;
    PoolScratch tmp(c);
    double *xd = nullptr, *XcT = nullptr, *scd = nullptr, *Ks = nullptr, *msd = nullptr, *o1 = nullptr, *acc = nullptr, *V = nullptr;
    int arc = PGP_OK;
    auto take = [&](double** out, size_t bytes) { if (arc == PGP_OK) arc = tmp.alloc(out, bytes); };
    take(&acc, (NSB + 1) * sizeof(double));                      // [NSB]: "a rank failed in this batch", summed with the column sums
    take(&xd, NSB * d * sizeof(double));
    take(&XcT, (size_t)dpad * ldc * sizeof(double));
    take(&scd, dpad * sizeof(double));
    take(&Ks, (size_t)np * NSB * sizeof(double));
    take(&msd, NSB * sizeof(double));
    take(&o1, NSB * sizeof(double));
    take(&V, (size_t)w * NSB * sizeof(double));
    if (m->world > 1) {
        double any = 0.0;
        CHK(comm_agree_max(m, arc != PGP_OK ? 1.0 : 0.0, &any, st));
        if (arc != PGP_OK) return arc;
        if (any != 0.0) { pgp_set_last_hip_error(hipErrorOutOfMemory, "another rank of the sharded predict ran out of device memory", __FILE__, __LINE__); return PGP_ERR_HIP; }
    } else if (arc != PGP_OK) return arc;
    HIP_TRY(hipMemcpyAsync(scd, f->cs.scale.data(), d * sizeof(double), hipMemcpyHostToDevice, st));
    CovSpec cp = f->cs;
    cp.cp.der = -1; cp.pg.der = -1;
    std::vector<double> acc_h(NSB + 1);
    // one batch of test points up to the all-reduce; a failure here must not leave the other ranks inside that collective
    auto batch = [&](long a, long nb_, int nrhs) -> int {
        HIP_TRY(hipMemcpyAsync(xd, xs + a * d, nb_ * d * sizeof(double), hipMemcpyHostToDevice, st));
        if (ms) HIP_TRY(hipMemcpyAsync(msd, ms + a, nb_ * sizeof(double), hipMemcpyHostToDevice, st));
        else HIP_TRY(hipMemsetAsync(msd, 0, nb_ * sizeof(double), st));
        CHK(scale_transpose_launch(xd, nb_, d, scd, XcT, ldc, dpad, st));
        HIP_TRY(hipMemsetAsync(Ks, 0, (size_t)np * nrhs * sizeof(double), st));
        CHK(cov_rect_launch(XcT, ldc, nb_, f->XT, np, n, dpad, cp, Ks, np, st));     // column-major (np x nrhs): column = test point
        CHK(col_dot_full_launch(Ks, np, n, nb_, f->alpha, msd, o1, st));                 // fmu = ms + Ks' alpha (every rank, O(n ns))
        for (int k = 0; k < f->nloc; ++k) {
            const int p = f->me + k * f->world;
            const long rows = (long)(p + 1) * w;
            GemmArgs g{};
            g.A = f->Bufs + (size_t)k * ldp * w + (ldp - rows); g.lda = ldp; g.a_kc = 1;     // A(m, k) = E_p(k, m)
            g.B = Ks; g.ldb = np; g.b_kc = 1;                                              // B(t, k) = Ks(k, t)
            g.C = V; g.ldc = w;
            g.M = w; g.N = nrhs; g.K = (int)rows; g.alpha = 1.0; g.beta = 0.0;
            g.kmode = KM_LT_I; g.koff = (int)((long)p * w);                                // E_p(k, m) = 0 for k > p w + m
            const long t128 = (long)(w / 128) * (nrhs / 128);
            g.tile = t128 < c->small_tile_below ? 64 : 128;
            g.flops = 2.0 * (double)w * nrhs * ((double)rows - 0.5 * w);
            CHK(gemm_prof(c, PC_GEMM_SOLVE, g, st));
            CHK(colsumsq_acc_launch(V, w, w, nb_, acc, st));
        }
        return PGP_OK;
    };
    for (long a = 0; a < ns; a += NSB) {
        const long nb_ = std::min<long>(NSB, ns - a);
        const int nrhs = (int)round_up(nb_, 128);
        HIP_TRY(hipMemsetAsync(acc, 0, (size_t)(nrhs + 1) * sizeof(double), st));
        const int brc = batch(a, nb_, nrhs);
        if (brc != PGP_OK) {
            if (m->world == 1) return brc;
            (void)hipDeviceSynchronize();
            (void)hipGetLastError();
            const double one = 1.0;
            (void)hipMemcpy(acc + nrhs, &one, sizeof(double), hipMemcpyHostToDevice);
        }
        CHK(comm_allreduce(m, acc, (size_t)nrhs + 1, 0, st));
        HIP_TRY(hipMemcpyAsync(fmu + a, o1, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(acc_h.data(), acc, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(acc_h.data() + NSB, acc + nrhs, sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (acc_h[NSB] != 0.0) {
            if (brc != PGP_OK) return brc;
            pgp_set_last_hip_error(hipErrorUnknown, "another rank of the sharded predict failed", __FILE__, __LINE__);
            return PGP_ERR_HIP;
        }
        for (long j = 0; j < nb_; ++j) fs2[a + j] = std::max(f->kss - acc_h[j] / f->sn2, 0.0);
    }
    if (c->prof) prof_collect(c);
    return PGP_OK;
}

}  // extern "C"
