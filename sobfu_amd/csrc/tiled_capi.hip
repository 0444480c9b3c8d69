// Native multi-GPU solver loop: one rank (process, GPU) per z-slab, halo exchange by RCCL send/recv over xGMI issued
// from C++ on a dedicated communication stream and overlapped with the interior compute.
//
// Same decomposition and schedule as sobfu_amd/tiled.py (which documents the invariants), without a Python round trip
// per iteration:
//     A_bnd (planes next to an interior face)  ->  event  ->  [comm stream] group{send, recv} of 4 nabla_U planes / face
//     A_int, B_int (planes whose +-3 taps are owned)            ... run while the exchange is in flight
//     wait(comm)  ->  B_bnd (remaining planes out to owned +-1)
// (both boundary regions of a pass are ONE two-range launch; thin slabs keep pass A unsplit).  When a non-negative threshold
// can fire the device-side gate needs the GLOBAL max-norm: see sobfu_hip_tiled_iterate for the ping-pong / late-gate scheme
// that keeps that all-reduce off the critical path.
//
// RCCL is not a link-time dependency: the host process (PyTorch) has already loaded librccl.so; sobfu_hip_tiled_load_rccl
// dlopen()s that same file and resolves the nine entry points used here, so libsobfu_hip.so still loads on a machine
// without RCCL and a process never holds two copies of the library.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"
#include "sobfu_launch.hpp"


// The handful of RCCL (= NCCL API) types this file needs, declared here so that neither building nor loading the library
// depends on an RCCL installation (values as in rccl.h; the entry points are resolved with dlsym at run time).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint32 = 3, ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return handle != nullptr; }
} g_rccl;

constexpr int kHalo = 4, kSlots = 256;
constexpr int kSplitAMaxPlanes = 64;  // owned planes up to which pass A is split into boundary + interior launches

#define RCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) {                                                                \
            std::fprintf(stderr, "sobfu_hip: RCCL error %d (%s) at %s:%d\n", (int) _r,            \
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?", __FILE__, __LINE__); \
            return SOBFU_E_RCCL;                                                                \
        }                                                                                       \
    } while (0)

float host_sqrt_rd(float s) {
    float r = std::sqrt(s);
    if (r > 0.f && (double) r * (double) r > (double) s) r = std::nextafterf(r, -INFINITY);
    return r;
}

}  // namespace

struct sobfu_hip_tiled {
    int X, Y, Z, world, rank;
    int z0, z1, lo, hi, Lz, own_lo, own_hi, zbase;
    sobfu_hip_solver_params p;
    float taps[7];
    ncclComm_t comm = nullptr;
    sobfu_hip_tiled_exchange_fn xfn = nullptr;    // transport of a communicator-less handle
    sobfu_hip_tiled_allreduce_fn rfn = nullptr;
    void* tctx = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_bnd = nullptr, ev_xchg = nullptr, ev_row = nullptr, ev_red[2] = {nullptr, nullptr};
    // optional second communicator + stream for the max-norm all-reduce (sobfu_hip_tiled_add_reduce_comm): it then never queues
    // behind (or in front of) a halo exchange on the main communicator
    ncclComm_t comm2 = nullptr;
    hipStream_t red_stream = nullptr;
    // compact slab state (see sobfu_hip_solver_set_compact): 12-byte psi / nabla_U, tsdf-only F / G / phi_n
    float *nU = nullptr, *c_psi = nullptr, *c_psi2 = nullptr, *c_f = nullptr, *c_f2 = nullptr, *c_g = nullptr, *c_n = nullptr;
    uint32_t* slots = nullptr;
    int slots_iters = 0;
    size_t NL, NF;
    int schedule = 0;  // 0 heuristic, 1 overlapped + pass A split, 2 overlapped + pass A whole, 3 serial (sobfu_hip_tiled_set_schedule)
    double last_enqueue_us = 0.0;  // host time per iteration the last iterate() spent issuing the loop (diagnostics)
};

extern "C" {

int sobfu_hip_tiled_load_rccl(const char* librccl_path) {
    if (g_rccl.ok()) return 0;
    SOBFU_CHECK_ARGS(librccl_path);
    void* h = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        std::fprintf(stderr, "sobfu_hip: cannot dlopen %s: %s\n", librccl_path, dlerror());
        return SOBFU_E_RCCL;
    }
#define SYM(field, name)                                         \
    *(void**) (&g_rccl.field) = dlsym(h, name);                  \
    if (!g_rccl.field) {                                         \
        std::fprintf(stderr, "sobfu_hip: %s has no %s\n", librccl_path, name); \
        return SOBFU_E_RCCL;                                     \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.handle = h;
    return 0;
}

int sobfu_hip_tiled_unique_id(char out[128]) {
    SOBFU_CHECK_ARGS(out);
    if (!g_rccl.ok()) return SOBFU_E_RCCL;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out, &id, 128);
    return 0;
}

int sobfu_hip_tiled_destroy(sobfu_hip_tiled* t) {
    if (!t) return 0;
    for (float* q : {t->nU, t->c_psi, t->c_psi2, t->c_f, t->c_f2, t->c_g, t->c_n})
        if (q) (void) hipFree(q);
    if (t->slots) (void) hipFree(t->slots);
    if (t->ev_bnd) (void) hipEventDestroy(t->ev_bnd);
    if (t->ev_xchg) (void) hipEventDestroy(t->ev_xchg);
    for (hipEvent_t e : {t->ev_red[0], t->ev_red[1], t->ev_row})
        if (e) (void) hipEventDestroy(e);
    if (t->red_stream) (void) hipStreamDestroy(t->red_stream);
    if (t->comm2 && g_rccl.ok()) (void) g_rccl.CommDestroy(t->comm2);
    if (t->comm_stream) (void) hipStreamDestroy(t->comm_stream);
    if (t->comm && g_rccl.ok()) (void) g_rccl.CommDestroy(t->comm);
    delete t;
    return 0;
}

int sobfu_hip_tiled_create(sobfu_hip_tiled** out, int X, int Y, int Z, int world, int rank, const char unique_id[128],
                           const sobfu_hip_solver_params* params) {
    SOBFU_CHECK_ARGS(out && params && unique_id && X > 1 && Y > 1 && Z > 1 && world >= 1 && rank >= 0 && rank < world);
    bool dry = true;  // an all-zero id asks for a communicator-less handle: slab layout, kernels and stream choreography of
    for (int i = 0; i < 128; ++i) dry = dry && unique_id[i] == 0;  // (world, rank); transport: sobfu_hip_tiled_set_transport
    if (!dry && !g_rccl.ok()) return SOBFU_E_RCCL;
    if (Z < world * kHalo || params->s < 7) return SOBFU_E_UNSUPPORTED;
    auto* t = new sobfu_hip_tiled();
    t->X = X; t->Y = Y; t->Z = Z; t->world = world; t->rank = rank;
    const int base = Z / world, rem = Z % world;
    t->z0 = rank * base + (rank < rem ? rank : rem);
    t->z1 = t->z0 + base + (rank < rem ? 1 : 0);
    t->lo = rank > 0 ? kHalo : 0;
    t->hi = rank < world - 1 ? kHalo : 0;
    t->Lz = (t->z1 - t->z0) + t->lo + t->hi;
    t->own_lo = t->lo;
    t->own_hi = t->lo + (t->z1 - t->z0);
    t->zbase  = t->z0 - t->lo;
    t->NL = (size_t) X * Y * t->Lz;
    t->NF = (size_t) X * Y * Z;
    t->p  = *params;
    float h[16];
    int rc = sobfu_hip_sobolev_filter(params->s, params->lambda, h);
    for (int i = 0; i < 7; ++i) t->taps[i] = h[i];
    if (rc == 0 && (t->z1 - t->z0) < kHalo && world > 1) rc = SOBFU_E_UNSUPPORTED;
    if (rc == 0) rc = (int) hipMalloc((void**) &t->nU, t->NL * 12);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_psi, t->NL * 12);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_f, t->NL * 4);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_psi2, t->NL * 12);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_f2, t->NL * 4);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_g, t->NL * 4);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_n, t->NF * 4);
    if (rc == 0) {  // max-norm slot rows for 4096 iterations up front: a solve never reallocates inside a timed region
        rc = (int) hipMalloc((void**) &t->slots, (size_t) (4096 + 1) * kSlots * 4);
        if (rc == 0) t->slots_iters = 4096;
    }
    if (rc == 0) rc = (int) hipStreamCreateWithFlags(&t->comm_stream, hipStreamNonBlocking);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_bnd, hipEventDisableTiming);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_xchg, hipEventDisableTiming);
    // the max-norm rows are written by pass B's atomics on this device, but a real RCCL all-reduce may let PEERS write the
    // reduced row straight into this buffer (direct / registered-buffer paths): every event keeps the system-scope fence
    // until the fence-less variant has been validated on >= 2 real GPUs.  SOBFU_TILED_LOCAL_EVENTS=1 opts into events
    // without the fence (an L2 write-back + invalidate less per edge) for the row hand-offs.
    const char* le = std::getenv("SOBFU_TILED_LOCAL_EVENTS");
    const unsigned local_ev = hipEventDisableTiming | ((le && le[0] == '1') ? hipEventDisableSystemFence : 0u);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_red[0], local_ev);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_red[1], local_ev);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_row, local_ev);
    if (rc == 0 && !dry) {
        ncclUniqueId id;
        std::memcpy(&id, unique_id, 128);
        ncclResult_t r = g_rccl.CommInitRank(&t->comm, world, id, rank);
        if (r != ncclSuccess) {
            std::fprintf(stderr, "sobfu_hip: ncclCommInitRank failed: %s\n", g_rccl.GetErrorString(r));
            rc = SOBFU_E_RCCL;
        }
    }
    if (rc != 0) {
        sobfu_hip_tiled_destroy(t);
        return rc;
    }
    *out = t;
    return 0;
}

int sobfu_hip_tiled_set_transport(sobfu_hip_tiled* t, sobfu_hip_tiled_exchange_fn exchange_fn, sobfu_hip_tiled_allreduce_fn allreduce_fn,
                                  void* ctx) {
    SOBFU_CHECK_ARGS(t && !t->comm);
    t->xfn = exchange_fn;
    t->rfn = allreduce_fn;
    t->tctx = ctx;
    return 0;
}

int sobfu_hip_tiled_add_reduce_comm(sobfu_hip_tiled* t, const char unique_id[128]) {
    SOBFU_CHECK_ARGS(t && unique_id && t->comm && !t->comm2);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, 128);
    RCCL_TRY(g_rccl.CommInitRank(&t->comm2, t->world, id, t->rank));
    SOBFU_HIP_TRY(hipStreamCreateWithFlags(&t->red_stream, hipStreamNonBlocking));
    return 0;
}

int sobfu_hip_tiled_set_schedule(sobfu_hip_tiled* t, int schedule) {
    SOBFU_CHECK_ARGS(t && schedule >= 0 && schedule <= 3);
    t->schedule = schedule;
    return 0;
}

double sobfu_hip_tiled_last_enqueue_us(const sobfu_hip_tiled* t) { return t ? t->last_enqueue_us : 0.0; }

int sobfu_hip_tiled_layout(const sobfu_hip_tiled* t, int* z0, int* z1, int* lo, int* hi, int* Lz, int* zbase) {
    SOBFU_CHECK_ARGS(t);
    if (z0) *z0 = t->z0;
    if (z1) *z1 = t->z1;
    if (lo) *lo = t->lo;
    if (hi) *hi = t->hi;
    if (Lz) *Lz = t->Lz;
    if (zbase) *zbase = t->zbase;
    return 0;
}

// One grouped send/recv of `planes` owned planes per interior face of a 12-byte field (floats: 3 per voxel).
static int exchange(sobfu_hip_tiled* t, float* field3, int planes, hipStream_t stream) {
    const size_t plane_f = (size_t) t->X * t->Y * 3, cnt = plane_f * planes;
    if (!t->comm) return t->xfn ? t->xfn(t->tctx, t->rank, field3, planes, (void*) stream) : 0;  // user transport / dry handle
    RCCL_TRY(g_rccl.GroupStart());
    if (t->rank > 0) {
        RCCL_TRY(g_rccl.Send(field3 + plane_f * t->own_lo, cnt, ncclFloat32, t->rank - 1, t->comm, stream));
        RCCL_TRY(g_rccl.Recv(field3 + plane_f * (t->own_lo - planes), cnt, ncclFloat32, t->rank - 1, t->comm, stream));
    }
    if (t->rank < t->world - 1) {
        RCCL_TRY(g_rccl.Send(field3 + plane_f * (t->own_hi - planes), cnt, ncclFloat32, t->rank + 1, t->comm, stream));
        RCCL_TRY(g_rccl.Recv(field3 + plane_f * t->own_hi, cnt, ncclFloat32, t->rank + 1, t->comm, stream));
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

static int allreduce_max(sobfu_hip_tiled* t, uint32_t* buf, size_t n, hipStream_t stream, bool own_comm = false) {
    if (!t->comm) return t->rfn ? t->rfn(t->tctx, t->rank, buf, n, (void*) stream) : 0;
    RCCL_TRY(g_rccl.AllReduce(buf, buf, n, ncclUint32, ncclMax, own_comm ? t->comm2 : t->comm, stream));
    return 0;
}

// Debug / bring-up: exchange `planes` planes of a caller-provided 12-byte slab field exactly as the loop does.
int sobfu_hip_tiled_exchange(sobfu_hip_tiled* t, float* d_field3, int planes, void* stream) {
    SOBFU_CHECK_ARGS(t && d_field3 && planes > 0 && planes <= kHalo);
    return exchange(t, d_field3, planes, (hipStream_t) stream);
}

// Bring-up self test usable with ONE rank: a world-1 communicator sends n floats from d_src to itself into d_dst through
// the same group{send, recv} path the loop uses (RCCL allows self send/recv inside a group).
int sobfu_hip_tiled_self_sendrecv(sobfu_hip_tiled* t, const float* d_src, float* d_dst, size_t n, void* stream) {
    SOBFU_CHECK_ARGS(t && t->comm && d_src && d_dst && n > 0);
    RCCL_TRY(g_rccl.GroupStart());
    RCCL_TRY(g_rccl.Send(d_src, n, ncclFloat32, t->rank, t->comm, (hipStream_t) stream));
    RCCL_TRY(g_rccl.Recv(d_dst, n, ncclFloat32, t->rank, t->comm, (hipStream_t) stream));
    RCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

int sobfu_hip_tiled_allreduce_max_u32(sobfu_hip_tiled* t, uint32_t* d_buf, size_t n, void* stream) {
    SOBFU_CHECK_ARGS(t && t->comm && d_buf && n > 0);
    RCCL_TRY(g_rccl.AllReduce(d_buf, d_buf, n, ncclUint32, ncclMax, t->comm, (hipStream_t) stream));
    return 0;
}

// The gradient-descent loop (reference src/sobfu/cuda/solver.cu:106-193) on this rank's slab.  API-format arguments:
// d_phi_global_local / d_phi_n_psi_local float2 (X, Y, Lz), d_phi_n_full float2 (X, Y, Z), d_psi_local float4 (X, Y, Lz)
// whose owned +-1 planes are exact on entry (identity: sobfu_hip_tile_init_identity) and on exit.  Synchronises `stream`.
//
// Convergence without a stall: psi and F = (phi_n o psi).tsdf are PING-PONGED (iteration k reads buffer (k-1)&1 and writes
// buffer k&1), and the device-side gate of iteration k looks at the max-norm row of iteration k-2.  When the threshold fires
// at iteration k, iteration k+1 has already run speculatively -- into the OTHER buffer -- every later launch is a no-op,
// and the state the reference's `break` (solver.cu:183) leaves is intact in buffer k&1.  The all-reduce that makes row k
// global therefore has a whole iteration to complete and is issued on the comm stream behind the exchange: nothing in the
// loop ever waits for a reduction that is still in flight (a same-iteration gate costs an exposed collective or a
// stream round trip per iteration: 67 vs 55 us per iteration in the N = 8 compute-side timing).
int sobfu_hip_tiled_iterate(sobfu_hip_tiled* t, const float* d_phi_global_local, const float* d_phi_n_full,
                            float* d_phi_n_psi_local, float* d_psi_local, int n_iters, sobfu_hip_solver_report* report,
                            float* per_iter_max_norm, void* stream) {
    SOBFU_CHECK_ARGS(t && d_phi_global_local && d_phi_n_full && d_phi_n_psi_local && d_psi_local && n_iters >= 0);
    hipStream_t st = (hipStream_t) stream;
    const int X = t->X, Y = t->Y, Z = t->Z, Lz = t->Lz;
    const sobfu_hip_solver_params& p = t->p;
    sobfu_hip_solver_report r{};
    r.last_max_update_norm = r.last_max_update_index = r.last_e_data = r.last_e_reg = NAN;
    float* P[2] = {t->c_psi, t->c_psi2};
    float* F[2] = {t->c_f, t->c_f2};
    // enter the compact format (includes the warp of solver.cu:106); both halves start equal so that planes no launch
    // writes (beyond owned +-1) hold the caller's values whichever half the loop ends in
    SOBFU_TRY(sobfu_hip::launch_pack_vec(d_psi_local, P[0], t->NL, st));
    SOBFU_TRY(sobfu_hip::launch_extract_tsdf(d_phi_global_local, t->c_g, t->NL, st));
    SOBFU_TRY(sobfu_hip::launch_extract_tsdf(d_phi_n_full, t->c_n, t->NF, st));
    SOBFU_TRY(sobfu_hip::launch_apply_tsdf_only(t->c_n, F[0], P[0], X, Y, Lz, st, Z));
    SOBFU_HIP_TRY(hipMemcpyAsync(P[1], P[0], t->NL * 12, hipMemcpyDeviceToDevice, st));
    SOBFU_HIP_TRY(hipMemcpyAsync(F[1], F[0], t->NL * 4, hipMemcpyDeviceToDevice, st));
    if (n_iters > t->slots_iters) {
        if (t->slots) SOBFU_HIP_TRY(hipFree(t->slots));
        t->slots = nullptr;
        SOBFU_HIP_TRY(hipMalloc((void**) &t->slots, (size_t) (n_iters + 1) * kSlots * 4));
        t->slots_iters = n_iters;
    }
    if (n_iters > 0) SOBFU_HIP_TRY(hipMemsetAsync(t->slots, 0, (size_t) (n_iters + 1) * kSlots * 4, st));
    // SOBFU_TILED_FORCE_COMM=1 runs the communication choreography (streams, events, empty exchange group, world-1
    // all-reduce) on a single rank too: bring-up / test hook for 1-GPU machines
    const char* force = std::getenv("SOBFU_TILED_FORCE_COMM");
    const bool can_converge = p.max_update_norm >= 0.f, multi = t->world > 1 || (force && force[0] == '1');
    const int lo = t->own_lo, hi = t->own_hi, H = kHalo;
    const int a_lo = t->lo ? std::min(lo + H, hi) : lo, a_hi = t->hi ? std::max(hi - H, a_lo) : hi;
    const int b_lo = t->lo ? std::min(lo + 3, hi) : lo, b_hi = t->hi ? std::max(hi - 3, b_lo) : hi;
    const int b_first = t->lo ? lo - 1 : lo, b_last = t->hi ? hi + 1 : hi;
    // pass A split into boundary + interior launches so that the exchange starts after 4 planes per face instead of after
    // the whole pass: an extra launch (+6-7 us per iteration in the compute-only timing at N = 4 and 8), worth it only where
    // the slab is so thin that the 3.1 MB face messages cannot hide behind B_int alone (N >= 4 at 256^3, if a face takes the ~65 us that ~60 GB/s per xGMI direction implies)
    // the schedule (results do not depend on it): environment (debugging) > sobfu_hip_tiled_set_schedule (autotuner) > heuristic
    const char* sa = std::getenv("SOBFU_TILED_SPLIT_A");
    const char* se = std::getenv("SOBFU_TILED_SERIAL");
    const bool want_split = sa ? sa[0] == '1' : (t->schedule == 1 ? true : (t->schedule == 2 ? false : (hi - lo) <= kSplitAMaxPlanes));
    const bool split_a = (t->lo || t->hi) && a_hi > a_lo && want_split;
    const bool serial = se ? se[0] == '1' : t->schedule == 3;
    // Where the all-reduce of a max-norm row runs (the late gate gives row j until pass B of iteration j+2):
    //   own communicator + stream (sobfu_hip_tiled_add_reduce_comm): issued right after row j's pass B, never in the way of
    //   an exchange; otherwise on the comm stream behind the next exchange (overlapped schedules) or in line (serial).
    enum { RED_NONE, RED_INLINE, RED_COMM_STREAM, RED_OWN_COMM };
    const int red_mode = (!multi || !can_converge) ? RED_NONE : (t->comm2 ? RED_OWN_COMM : (serial ? RED_INLINE : RED_COMM_STREAM));
    bool red_issued[2] = {false, false};  // an asynchronous reduce of the latest row of this parity is behind ev_red[parity]
    const auto host_t0 = std::chrono::steady_clock::now();
    for (int it = 1; it <= n_iters; ++it) {
        const float *psi_in = P[(it - 1) & 1], *f_in = F[(it - 1) & 1];
        float *psi_out = P[it & 1], *f_out = F[it & 1];
        const uint32_t* prev = (it > 2 && can_converge) ? t->slots + (size_t) (it - 2) * kSlots : nullptr;  // the late gate
        uint32_t* row        = t->slots + (size_t) it * kSlots;
        auto A = [&](int za, int zb, int za2 = 0, int zb2 = 0) {  // pass A writes scratch only: never gated
            return sobfu_hip::launch_pass_a(f_in, t->c_g, psi_in, t->nU, p.w_reg, X, Y, Lz, nullptr, 0.f, 0, st, true, za, zb, za2, zb2);
        };
        auto B = [&](int za, int zb, int za2 = 0, int zb2 = 0) {
            return sobfu_hip::launch_pass_b(t->nU, const_cast<float*>(psi_in), t->c_n, f_out, nullptr, row, t->taps, p.alpha, X, Y, Lz, prev,
                                            p.max_update_norm, 0, st, Z, lo, hi, true, za, zb, za2, zb2, psi_out, it > 3 ? 2 : 1);
        };
        auto wait_gate = [&]() -> int {  // row it-2 must be global before the first pass-B launch of this iteration
            if (prev && red_issued[it & 1]) SOBFU_HIP_TRY(hipStreamWaitEvent(st, t->ev_red[it & 1], 0));
            return 0;
        };
        auto after_b = [&]() -> int {  // row `it` is complete on `st`; it gates iteration it+2
            if (it > n_iters - 2) return 0;  // the tail rows are reduced once, after the loop
            uint32_t* r_ = t->slots + (size_t) it * kSlots;
            if (red_mode == RED_INLINE) SOBFU_TRY(allreduce_max(t, r_, kSlots, st));
            if (red_mode == RED_OWN_COMM) {
                SOBFU_HIP_TRY(hipEventRecord(t->ev_row, st));
                SOBFU_HIP_TRY(hipStreamWaitEvent(t->red_stream, t->ev_row, 0));
                SOBFU_TRY(allreduce_max(t, r_, kSlots, t->red_stream, true));
                SOBFU_HIP_TRY(hipEventRecord(t->ev_red[it & 1], t->red_stream));
                red_issued[it & 1] = true;
            }
            return 0;
        };
        if (serial) {
            // no overlap, no cross-stream events: pass A, the exchange and pass B in line on `st`.  Every event record / wait
            // between two kernels costs a few microseconds of drained pipeline (~20 us per iteration for the overlapped
            // schedule's three), which a fast exchange on a thin slab does not repay.
            SOBFU_TRY(A(lo, hi));
            if (multi) SOBFU_TRY(exchange(t, t->nU, H, st));
            SOBFU_TRY(wait_gate());
            SOBFU_TRY(B(b_first, b_last));
            SOBFU_TRY(after_b());
            continue;
        }
        // both boundary regions of a pass go out as ONE launch (two plane ranges)
        if (split_a) SOBFU_TRY(A(lo, a_lo, a_hi, hi));
        else SOBFU_TRY(A(lo, hi));
        if (multi) {
            SOBFU_HIP_TRY(hipEventRecord(t->ev_bnd, st));
            SOBFU_HIP_TRY(hipStreamWaitEvent(t->comm_stream, t->ev_bnd, 0));
            SOBFU_TRY(exchange(t, t->nU, H, t->comm_stream));
            SOBFU_HIP_TRY(hipEventRecord(t->ev_xchg, t->comm_stream));
            if (red_mode == RED_COMM_STREAM && it >= 2 && it < n_iters) {
                // row it-1 is complete (its pass B precedes this iteration's ev_bnd, which the comm stream has waited for) and
                // gates iteration it+1: reduce it behind this iteration's exchange
                SOBFU_TRY(allreduce_max(t, t->slots + (size_t) (it - 1) * kSlots, kSlots, t->comm_stream));
                SOBFU_HIP_TRY(hipEventRecord(t->ev_red[(it - 1) & 1], t->comm_stream));
                red_issued[(it - 1) & 1] = true;
            }
        }
        if (split_a && a_hi > a_lo) SOBFU_TRY(A(a_lo, a_hi));
        SOBFU_TRY(wait_gate());
        if (b_hi > b_lo) SOBFU_TRY(B(b_lo, b_hi));
        if (multi) SOBFU_HIP_TRY(hipStreamWaitEvent(st, t->ev_xchg, 0));
        if (b_lo > b_first || b_last > b_hi) SOBFU_TRY(B(b_first, b_lo, b_hi, b_last));
        SOBFU_TRY(after_b());
        // the next iteration's A_bnd overwrites nabla_U planes the exchange of THIS iteration sent: it runs on `st` after
        // the wait above, so the sends have completed by then; the next exchange's receives overwrite halo planes B_bnd
        // of THIS iteration read: the comm stream starts it only after the next ev_bnd, recorded on `st` behind B_bnd
    }
    if (n_iters > 0)
        t->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count() / n_iters;
    if (multi && n_iters > 0) {  // rows the loop has not reduced yet: all of them without a threshold, the tail otherwise
        for (int q = 0; q < 2; ++q)
            if (red_issued[q]) SOBFU_HIP_TRY(hipStreamWaitEvent(st, t->ev_red[q], 0));
        const int first = can_converge ? std::max(1, n_iters - 1) : 1;  // rows 1 .. n_iters-2 went through the comm stream
        SOBFU_TRY(allreduce_max(t, t->slots + (size_t) first * kSlots, (size_t) (n_iters - first + 1) * kSlots, st));
    }
    std::vector<uint32_t> hs((size_t) std::max(n_iters, 1) * kSlots, 0u);
    if (n_iters > 0) SOBFU_HIP_TRY(hipMemcpyAsync(hs.data(), t->slots + kSlots, (size_t) n_iters * kSlots * 4, hipMemcpyDeviceToHost, st));
    SOBFU_HIP_TRY(hipStreamSynchronize(st));
    int done = n_iters;
    for (int k = 0; k < n_iters; ++k) {
        uint32_t m = 0;
        for (int i = 0; i < kSlots; ++i) m = std::max(m, hs[(size_t) k * kSlots + i]);
        float f;
        std::memcpy(&f, &m, 4);
        const float v = host_sqrt_rd(f);
        if (per_iter_max_norm) per_iter_max_norm[k] = v;
        r.last_max_update_norm = v;
        if (can_converge && v <= p.max_update_norm) {  // solver.cu:183 -- iteration k+2 ran speculatively, later ones not at all
            done = k + 1;
            r.converged = 1;
            break;
        }
    }
    r.iterations = done;
    // leave the compact format from the half that holds the state after `done` iterations: psi.xyz back,
    // phi_n o psi = apply(phi_n, psi) (the state of solver.cu:168)
    SOBFU_TRY(sobfu_hip::launch_unpack_vec(P[done & 1], d_psi_local, t->NL, st));
    SOBFU_TRY(sobfu_hip_tile_apply(d_phi_n_full, Z, d_phi_n_psi_local, d_psi_local, X, Y, Lz, st));
    SOBFU_HIP_TRY(hipStreamSynchronize(st));
    if (report) *report = r;
    return 0;
}

}  // extern "C"
