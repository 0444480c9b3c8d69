// Native multi-GPU solver loop: one rank (process, GPU) per volume TILE -- a Px x Py x Pz grid of tiles (2 x 2 x 2 on 8 GPUs,
// BASELINE config 4; 1 x 1 x N = z-slabs).
//
// Every field of a rank is a local array = owned cells + 4 halo cells on each side that faces a neighbour.  One iteration needs
// ONE exchange (Option B): pass A produces nabla_U on the owned cells, the 4-cell faces (and the 4 x 4 edge strips:
// the one-cell shells below read nabla_U diagonally across a tile edge, never across a corner) travel to the 3 face + 3 edge
// neighbours of a 2 x 2 x 2 tile, and pass B then updates psi / phi_n o psi on owned +- 1 along each axis -- the radius-3
// convolution is exact there, so the values the next pass A reads are never exchanged.  phi_n is replicated.
//
// The exchange is PART OF PASS A's LAUNCH (solver_kernels.hip, tile_potential_gradient_kernel): the cells of every message
// are evaluated by "push boxes" -- numbered first; short marches for the y / z faces (which also store their cells at home: the
// owned block leaves them out), lane per cell for the x face and the edge strips -- that store straight into the destination.
// Transports:
//   DIRECT    the destination is the neighbour's own nabla_U array, peer-mapped over xGMI (hipIpc; sobfu_hip_tiled_connect).
//             No pack, no unpack, no communication launch: the last push workgroup raises this rank's arrival flag at its
//             neighbours, the last workgroup of the launch waits for theirs (with a deadline), so the transfers overlap the
//             owned block's compute inside ONE launch and an iteration is two launches, as on a single GPU.  nabla_U is
//             double-buffered by iteration parity: a neighbour's pass A of iteration k+1 stores into the half this rank's
//             pass B of iteration k does not read; its pass A of k+2 cannot start before this rank's flag of k+1, which this
//             rank raises after its pass B of k has retired.  The max-norm rows become global the same way (every rank stores
//             its row maximum into entry `rank` of that row at every other rank): no collective anywhere in the loop.
//   RCCL      the destination is the packed send buffer; one grouped ncclSend / ncclRecv per exchange and one scatter kernel
//             (z-slabs: whole planes in place, and the overlapped schedules of round 1).
//   CALLBACK  the same buffers handed to a user function (in-process loopback and gloo bring-up transports of the tests).
// The one-cell x / y shells of pass B are DIRECT boxes of its launch (lane per cell), the z shells extra planes of the march.
//
// Schedules of the RCCL z-slab path: serial (pass A, exchange, pass B in line), or overlapped as in tests/tiled_reference.py:
//     A_bnd (planes next to an interior face)  ->  event  ->  [comm stream] group{send, recv} of 4 nabla_U planes / face
//     A_int, B_int (planes whose +-3 taps are owned)            ... run while the exchange is in flight
//     wait(comm)  ->  B_bnd (remaining planes out to owned +-1)
// When a non-negative threshold can fire the device-side gate needs the GLOBAL max-norm: see tiled_step for the ping-pong /
// late-gate scheme that keeps that reduction off the critical path.
//
// RCCL is not a link-time dependency: the host process (PyTorch) has already loaded librccl.so; sobfu_hip_tiled_load_rccl
// dlopen()s that same file and resolves the entry points used here, so libsobfu_hip.so still loads on a machine
// without RCCL and a process never holds two copies of the library.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
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
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return handle != nullptr; }
} g_rccl;

constexpr int kHalo = 4, kSlots = 256;
// iterations one solve may run on a handle whose max-norm rows other processes map (direct transport): the rows are part of the arena
// the peers open once, so their number is fixed at creation (16 MiB)
constexpr int kRowsIters = 16384;
constexpr int kSplitAMaxPlanes = 64;  // owned planes up to which pass A is split into boundary + interior launches
constexpr int kMaxSync = sobfu_hip::kMaxSync;
// The runtime carves allocations of up to GPU_MAX_SUBALLOC_SIZE (4 MiB by default) out of shared blocks, and hipIpcGetMemHandle
// refuses a pointer that is not the base of its block ("invalid argument", one start-up in ~10 on small test grids): whatever
// is exported to other processes is allocated at least this large, i.e. as a block of its own.
constexpr size_t kOwnBlock = (size_t) 8 << 20;

// Blocks that other processes map are never handed back to the runtime while the process lives: they are parked and reused by
// the next handle.  A block that has been exported keeps its 64-byte handle -- and the mappings the peers hold -- for good;
// exporting memory again after a free / re-allocation at the same address is what hipIpcGetMemHandle refused now and then
// when a process built several handles in a row (tile-grid autotuning).  One block per size class and kind in practice.
struct PooledBlock {
    void* ptr;
    size_t bytes;
    int device;
    bool uncached, in_use, has_handle;
    hipIpcMemHandle_t handle;
};
std::vector<PooledBlock> g_pool;
std::mutex g_pool_mutex;
// The export handle is taken when a block is allocated, and a block the runtime refuses to export is parked for good and
// replaced by another: hipIpcGetMemHandle now and then answers "invalid argument" for a perfectly ordinary fresh allocation
// (seen once in ~100 allocations when processes that had used IPC themselves had just exited) -- a second block has always worked.
int pool_take(void** out, size_t bytes, bool uncached) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int) hipGetLastError();
    PooledBlock* best = nullptr;  // best fit among the parked blocks of THIS device
    for (PooledBlock& b : g_pool)
        if (!b.in_use && b.device == dev && b.uncached == uncached && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best) {
        best->in_use = true;
        *out         = best->ptr;
        return 0;
    }
    for (int attempt = 0;; ++attempt) {
        void* p = nullptr;
        if (uncached && hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void) hipGetLastError();
            p = nullptr;  // plain device memory if the runtime refuses (the flags are polled with system-scope loads either way)
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) return (int) e;
        }
        PooledBlock b{p, bytes, dev, uncached, true, false, {}};
        const hipError_t e = hipIpcGetMemHandle(&b.handle, p);
        b.has_handle = e == hipSuccess;
        if (!b.has_handle) (void) hipGetLastError();
        if (b.has_handle || attempt == 3) {  // (after four refusals the block is used as it is: only the direct transport needs the handle)
            g_pool.push_back(b);
            *out = p;
            return 0;
        }
        std::fprintf(stderr, "sobfu_hip: hipIpcGetMemHandle refused a fresh %zu-byte block (%s); parking it and taking another\n", bytes,
                     hipGetErrorString(e));
        b.uncached = !uncached;  // never matches a later request of this kind ...
        b.bytes    = 0;          // ... or of any size
        g_pool.push_back(b);
    }
}
void pool_give_back(void* p) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (PooledBlock& b : g_pool)
        if (b.ptr == p) b.in_use = false;
}

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

// The layout of ANY rank's tile (a rank also needs its neighbours': where their halo cells sit in their arrays).
struct AxisLay {
    int g0, g1, lo, hi, L, o0, o1, base;
};
struct MsgGeom {
    int peer, dir[3];
    int sb[6], rb[6];  // cells sent / halo cells received (local cells of THIS rank)
    size_t cells;
};
struct TileLay {
    int P[3], c[3];
    AxisLay a[3];
    std::vector<MsgGeom> msgs;
    bool ok = true;
};
TileLay make_layout(const int dims[3], const int P[3], int rank) {
    TileLay t;
    const int c[3] = {rank % P[0], (rank / P[0]) % P[1], rank / (P[0] * P[1])};  // x fastest
    for (int k = 0; k < 3; ++k) {
        t.P[k] = P[k];
        t.c[k] = c[k];
        const int base = dims[k] / P[k], rem = dims[k] % P[k];  // the cells of an axis are split as evenly as possible
        AxisLay& a = t.a[k];
        a.g0   = c[k] * base + std::min(c[k], rem);
        a.g1   = a.g0 + base + (c[k] < rem ? 1 : 0);
        a.lo   = c[k] > 0 ? kHalo : 0;
        a.hi   = c[k] < P[k] - 1 ? kHalo : 0;
        a.L    = (a.g1 - a.g0) + a.lo + a.hi;
        a.o0   = a.lo;
        a.o1   = a.lo + (a.g1 - a.g0);
        a.base = a.g0 - a.lo;
        if (P[k] > 1 && base < kHalo) t.ok = false;  // a tile must own at least a halo's worth of cells per split axis
    }
    // halo messages: every face neighbour (one non-zero offset) and edge neighbour (two); corners are never read.  Along an
    // axis with offset +1 the 4 owned cells next to that face are sent and the 4 halo cells beyond it received; along an
    // axis with offset 0 the owned range (the same on both sides, as the neighbour shares this coordinate).
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int dl[3] = {dx, dy, dz};
                const int nnz = (dx != 0) + (dy != 0) + (dz != 0);
                if (nnz < 1 || nnz > 2) continue;
                bool inside = true;
                for (int k = 0; k < 3; ++k) inside = inside && c[k] + dl[k] >= 0 && c[k] + dl[k] < P[k];
                if (!inside) continue;
                MsgGeom m;
                m.cells = 1;
                for (int k = 0; k < 3; ++k) {
                    const AxisLay& a = t.a[k];
                    m.dir[k] = dl[k];
                    if (dl[k] > 0) { m.sb[2 * k] = a.o1 - kHalo; m.sb[2 * k + 1] = a.o1; m.rb[2 * k] = a.o1; m.rb[2 * k + 1] = a.o1 + kHalo; }
                    else if (dl[k] < 0) { m.sb[2 * k] = a.o0; m.sb[2 * k + 1] = a.o0 + kHalo; m.rb[2 * k] = a.o0 - kHalo; m.rb[2 * k + 1] = a.o0; }
                    else { m.sb[2 * k] = m.rb[2 * k] = a.o0; m.sb[2 * k + 1] = m.rb[2 * k + 1] = a.o1; }
                    m.cells *= (size_t) (m.sb[2 * k + 1] - m.sb[2 * k]);
                }
                m.peer = (c[0] + dx) + P[0] * ((c[1] + dy) + P[1] * (c[2] + dz));
                t.msgs.push_back(m);
            }
    // the z FACES go last: on the packed (RCCL / callback) transports of a 3-D tile they travel IN PLACE as whole padded planes of the
    // array (see exchange_packed), so the messages that do use the packed buffers are a prefix of the list and of the buffers
    std::stable_partition(t.msgs.begin(), t.msgs.end(), [](const MsgGeom& m) { return !(m.dir[0] == 0 && m.dir[1] == 0 && m.dir[2] != 0); });
    return t;
}

}  // namespace

struct sobfu_hip_tiled {
    int X, Y, Z, world, rank;
    // tile grid P, this rank's tile coordinates c; per axis: owned global range [g0, g1), halo cells lo / hi, local extent L,
    // owned local range [o0, o1), global coordinate `base` of local cell 0
    int P[3], c[3], g0[3], g1[3], lo[3], hi[3], L[3], o0[3], o1[3], base[3];
    bool slab;  // Px == Py == 1: on the RCCL transport halos are whole planes and travel in place (no pack / unpack)
    int z0, z1, Lz, own_lo, own_hi, zbase;  // the z entries again, under the names the slab schedules use
    sobfu_hip_solver_params p;
    float taps[7];
    ncclComm_t comm = nullptr;
    sobfu_hip_tiled_exchange_fn xfn = nullptr;    // transport of a communicator-less handle
    sobfu_hip_tiled_allreduce_fn rfn = nullptr;
    void* tctx = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_bnd = nullptr, ev_xchg = nullptr, ev_row = nullptr, ev_red[2] = {nullptr, nullptr}, ev_first = nullptr;
    // optional second communicator + stream for the max-norm all-reduce (sobfu_hip_tiled_add_reduce_comm): it then never queues
    // behind (or in front of) a halo exchange on the main communicator
    ncclComm_t comm2 = nullptr;
    hipStream_t red_stream = nullptr;
    // compact tile state (see sobfu_hip_solver_set_compact): 12-byte psi / nabla_U, tsdf-only F / G / phi_n.  nabla_U is
    // double-buffered by iteration parity on the tile path (see the top of the file); the slab schedules use nUb[0].
    float *nUb[2] = {nullptr, nullptr}, *c_psi = nullptr, *c_psi2 = nullptr, *c_f = nullptr, *c_f2 = nullptr, *c_g = nullptr, *c_n = nullptr;
    uint32_t* slots = nullptr;
    int slots_iters = 0;
    size_t NL, NF;
    // halo messages: one per face / edge neighbour, packed one after the other (same offsets on both sides)
    std::vector<sobfu_hip_tiled_msg> msgs;
    std::vector<MsgGeom> geom;
    std::vector<int> sboxes, rboxes;  // 6 ints per message: the cells sent / the halo cells received
    float *sendbuf = nullptr, *recvbuf = nullptr;
    // packed transports of a 3-D tile: the first n_packed messages travel through the buffers, the z faces behind them in place
    // (zmsgs: offsets into the nabla_U array itself -- 4 whole padded planes out of the owned rim, 4 into the halo)
    int n_packed = 0;
    std::vector<sobfu_hip_tiled_msg> zmsgs;
    uint32_t* scatter_table = nullptr;  // device: destination cell of every cell of the packed receive buffer (the loop's scatter, precomputed)
    unsigned scatter_cells = 0;
    // direct transport (sobfu_hip_tiled_connect): push destinations in the peers, signalling state
    bool direct = false, dead = false, first_checked = false, dry_packed = false, force_comm = false;
    int debug_skip = 0;  // timing experiments only (results are wrong): see tiled_step_impl
    int wait_enabled = 1;
    sobfu_hip::TileSync* sync_d = nullptr;
    // what the peers map (direct transport): ONE arena [nabla_U half 0 | half 1 | global max-norm rows ((slots_iters + 2) x 256)] and
    // the arrival flags [kMaxSync], uncached -- both padded to kOwnBlock so that hipIpcGetMemHandle accepts them
    char* arena = nullptr;
    size_t arena_bytes = 0, nu_off[2] = {0, 0}, rows_off = 0;
    uint32_t *flags = nullptr, *grows = nullptr, *grows_own = nullptr;  // grows_own: a larger private copy once a solve outgrew the arena's
    uint32_t seq_total = 0;                        // sequence numbers used so far (every rank counts the same)
    uint64_t timeout_ticks = 0;                    // deadline of the in-kernel waits (100 MHz ticks; SOBFU_TILED_DEADLINE_S at create)
    std::vector<sobfu_hip::TileLaunchBox> a_boxes[2];  // pass A's boxes per nabla_U half: one push box per message + the owned block
    sobfu_hip::TilePassAPlan* a_plan[2] = {nullptr, nullptr};  // ... and their launch plans (box lists in device memory)
    sobfu_hip::TileLaunchBox a_own{};  // the owned block alone (launches without messages)
    int schedule = 0;  // 0 heuristic, 1 overlapped + pass A split, 2 overlapped + pass A whole, 3 serial (sobfu_hip_tiled_set_schedule)
    double last_enqueue_us = 0.0;  // host time per iteration the last iterate() spent issuing the loop (diagnostics)
    // optional timing of the serial schedule's three pieces with HIP events on the loop's stream (sobfu_hip_tiled_set_profiling)
    int prof_stride = 0, prof_pending = 0, prof_n = 0;
    std::vector<hipEvent_t> prof_ev;
    double prof_ms[3] = {0, 0, 0};  // pass A, exchange (transfer + unpack; the direct transport has none), pass B
    struct Session {  // an open solve (tiled_begin .. tiled_end)
        bool active = false;
        const float* pn = nullptr;
        float *pnp = nullptr, *psi = nullptr;
        int cap = 0, launched = 0;
        bool flushed = false;  // the direct transport's end-of-solve handshake has been issued (test harnesses issue it as a phase)
        bool red_issued[2] = {false, false};
        int red_upto = 0;  // rows 1 .. red_upto have been (or are being) all-reduced
        uint32_t seq_base = 0;
    } q;
};

namespace {

double deadline_seconds() {
    const char* e = std::getenv("SOBFU_TILED_DEADLINE_S");
    const double v = e ? std::atof(e) : 0.0;
    return v > 0.0 ? v : 30.0;
}

// (re)builds pass A's box lists: one push box per message, then the owned block.  Destinations: the peers' halo cells when
// connected (dst[half][i] != null), else the packed send buffer.
void build_a_boxes(sobfu_hip_tiled* t, float* const* dst0, float* const* dst1, const TileLay* peers) {
    // The y and z FACES are pushed by short marches whose cells the owned block would compute a second time.  Instead such a box
    // stands in for the owned block on its cells (it stores them at home too) and the owned block shrinks: by the 4 rim planes
    // along z, by a whole 8-row tile along y (the face box then marches 8 rows, of which the 4 rim rows travel).  Where face
    // boxes meet, the z box is the one that stores at home.  The x face (4 cells of a 64-lane row: lane per cell) and the edge
    // strips stay push-only: shrinking the owned block by 4 columns would not save it a single workgroup.  Measured (one box, A/B):
    // 1 x 1 x 8 slabs of 256^3 43.1 -> 41.6 us per iteration; 2 x 2 x 2 and 1 x 2 x 4 tiles unchanged (43.4 / 44.7).
    const int ny_nb = (t->lo[1] ? 1 : 0) + (t->hi[1] ? 1 : 0), nz_nb = (t->lo[2] ? 1 : 0) + (t->hi[2] ? 1 : 0);
    const bool wide = (t->o1[0] - t->o0[0]) >= 64;  // rows wide enough for the faces to be MARCHED (thin rows: lane per cell, push-only)
    // packed (RCCL / callback) transports of a 3-D tile: the z faces leave IN PLACE, straight out of the owned block's rim planes -- no
    // push box evaluates them and the owned block keeps those planes
    const bool z_inplace = dst0 == nullptr && dst1 == nullptr && !t->slab && !t->zmsgs.empty();
    const bool z_home = !z_inplace && wide && nz_nb > 0 && (t->o1[2] - t->o0[2]) > kHalo * nz_nb;
    const bool y_home = wide && ny_nb > 0 && (t->o1[1] - t->o0[1]) > 8 * ny_nb;
    const int iz0 = t->o0[2] + ((z_home && t->lo[2]) ? kHalo : 0), iz1 = t->o1[2] - ((z_home && t->hi[2]) ? kHalo : 0);  // planes the z boxes leave
    const int iy0 = t->o0[1] + ((y_home && t->lo[1]) ? 8 : 0), iy1 = t->o1[1] - ((y_home && t->hi[1]) ? 8 : 0);
    for (int h = 0; h < 2; ++h) {
        std::vector<sobfu_hip::TileLaunchBox>& v = t->a_boxes[h];
        v.clear();
        for (size_t i = 0; i < t->geom.size(); ++i) {
            const MsgGeom& m = t->geom[i];
            sobfu_hip::TileLaunchBox b{};
            // wide rows (y / z faces): a short march costs a fifth of the loads of a lane-per-cell evaluation (1 x 1 x 8 slabs: pass A
            // 26.6 -> 18.4 us); thin in x: direct
            const bool march = (m.sb[1] - m.sb[0]) >= 64;
            b.box = sobfu_hip::LaunchBox{m.sb[0], m.sb[1], m.sb[2], m.sb[3], m.sb[4], m.sb[5], !march};
            b.push_y0 = m.sb[2]; b.push_y1 = m.sb[3];
            const bool face_z = march && m.dir[0] == 0 && m.dir[1] == 0 && m.dir[2] != 0, face_y = march && m.dir[0] == 0 && m.dir[2] == 0 && m.dir[1] != 0;
            if (face_z && z_home) { b.local_z0 = m.sb[4]; b.local_z1 = m.sb[5]; }
            if (face_y && y_home) {
                if (m.dir[1] > 0) b.box.y0 = t->o1[1] - 8; else b.box.y1 = t->o0[1] + 8;
                b.local_z0 = iz0; b.local_z1 = iz1;
            }
            if (z_inplace && m.dir[0] == 0 && m.dir[1] == 0 && m.dir[2] != 0) continue;
            if ((!march && (t->debug_skip & 128)) || (march && (t->debug_skip & 256))) continue;  // timing experiments: without the thin / the marched push boxes
            if (march && (t->debug_skip & 512)) b.push_y1 = b.push_y0;  // timing experiments: the marched boxes keep their cells at home only
            float* const* dst = h ? dst1 : dst0;
            if (dst && dst[i]) {  // the matching message of the peer: direction -dir; its receive box is where these cells live there
                const TileLay& pl = peers[i];
                const MsgGeom* pm = nullptr;
                for (const MsgGeom& g : pl.msgs)
                    if (g.dir[0] == -m.dir[0] && g.dir[1] == -m.dir[1] && g.dir[2] == -m.dir[2]) pm = &g;
                b.dst = dst[i];
                b.ox = pm->rb[0] - m.sb[0]; b.oy = pm->rb[2] - m.sb[2]; b.oz = pm->rb[4] - m.sb[4];
                b.px = pl.a[0].L; b.py = pl.a[1].L;
            } else {
                b.dst = t->sendbuf + t->msgs[i].send_off;
                b.ox = -m.sb[0]; b.oy = -m.sb[2]; b.oz = -m.sb[4];
                b.px = m.sb[1] - m.sb[0]; b.py = m.sb[3] - m.sb[2];
            }
            v.push_back(b);
        }
        sobfu_hip::TileLaunchBox own{};
        own.box = sobfu_hip::LaunchBox{t->o0[0], t->o1[0], iy0, iy1, iz0, iz1, false};
        v.push_back(own);
        // the same launch without messages (a world of one; timing experiments): the whole owned block
        t->a_own = sobfu_hip::TileLaunchBox{};
        t->a_own.box = sobfu_hip::LaunchBox{t->o0[0], t->o1[0], t->o0[1], t->o1[1], t->o0[2], t->o1[2], false};
        sobfu_hip::tile_pass_a_plan_destroy(t->a_plan[h]);
        t->a_plan[h] = nullptr;
        if (sobfu_hip::tile_pass_a_plan_create(&t->a_plan[h], v.data(), (int) v.size(), t->L[0], t->L[1], t->L[2]) != 0) t->a_plan[h] = nullptr;
    }
}

void abort_comms(sobfu_hip_tiled* t) {
    if (g_rccl.ok() && g_rccl.CommAbort) {
        if (t->comm2) { (void) g_rccl.CommAbort(t->comm2); t->comm2 = nullptr; }
        if (t->comm) { (void) g_rccl.CommAbort(t->comm); t->comm = nullptr; }
    }
    t->dead = true;
}

}  // namespace

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
    SYM(CommAbort, "ncclCommAbort")
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
    for (float* q : {t->c_psi, t->c_psi2, t->c_f, t->c_f2, t->c_g, t->c_n})
        if (q) (void) hipFree(q);
    if (t->arena) pool_give_back(t->arena);
    if (t->slots) (void) hipFree(t->slots);
    if (t->grows_own) (void) hipFree(t->grows_own);
    if (t->flags) pool_give_back(t->flags);
    if (t->sync_d) (void) hipFree(t->sync_d);
    for (int h = 0; h < 2; ++h) sobfu_hip::tile_pass_a_plan_destroy(t->a_plan[h]);
    for (hipEvent_t e : t->prof_ev) (void) hipEventDestroy(e);
    if (t->sendbuf) (void) hipFree(t->sendbuf);
    if (t->recvbuf) (void) hipFree(t->recvbuf);
    if (t->scatter_table) (void) hipFree(t->scatter_table);
    if (t->ev_bnd) (void) hipEventDestroy(t->ev_bnd);
    if (t->ev_xchg) (void) hipEventDestroy(t->ev_xchg);
    for (hipEvent_t e : {t->ev_red[0], t->ev_red[1], t->ev_row, t->ev_first})
        if (e) (void) hipEventDestroy(e);
    if (t->red_stream) (void) hipStreamDestroy(t->red_stream);
    if (t->comm2 && g_rccl.ok()) (void) g_rccl.CommDestroy(t->comm2);
    if (t->comm_stream) (void) hipStreamDestroy(t->comm_stream);
    if (t->comm && g_rccl.ok()) (void) g_rccl.CommDestroy(t->comm);
    delete t;
    return 0;
}

int sobfu_hip_tiled_create3(sobfu_hip_tiled** out, int X, int Y, int Z, int Px, int Py, int Pz, int rank, const char unique_id[128],
                            const sobfu_hip_solver_params* params) {
    SOBFU_CHECK_ARGS(out && params && unique_id && X > 1 && Y > 1 && Z > 1 && Px >= 1 && Py >= 1 && Pz >= 1 && rank >= 0 &&
                     rank < Px * Py * Pz);
    bool dry = true;  // an all-zero id asks for a communicator-less handle: tile layout, kernels and stream choreography of
    for (int i = 0; i < 128; ++i) dry = dry && unique_id[i] == 0;  // (grid, rank); transport: sobfu_hip_tiled_set_transport / _connect
    if (!dry && !g_rccl.ok()) return SOBFU_E_RCCL;
    if (params->s < 7) return SOBFU_E_UNSUPPORTED;
    auto* t = new sobfu_hip_tiled();
    t->X = X; t->Y = Y; t->Z = Z; t->world = Px * Py * Pz; t->rank = rank;
    const int dims[3] = {X, Y, Z}, P[3] = {Px, Py, Pz};
    const TileLay lay = make_layout(dims, P, rank);
    int rc = lay.ok ? 0 : SOBFU_E_UNSUPPORTED;
    for (int a = 0; a < 3; ++a) {
        t->P[a] = P[a]; t->c[a] = lay.c[a];
        t->g0[a] = lay.a[a].g0; t->g1[a] = lay.a[a].g1; t->lo[a] = lay.a[a].lo; t->hi[a] = lay.a[a].hi; t->L[a] = lay.a[a].L;
        t->o0[a] = lay.a[a].o0; t->o1[a] = lay.a[a].o1; t->base[a] = lay.a[a].base;
    }
    t->slab = Px == 1 && Py == 1;
    // bring-up / timing settings of this handle, read once ("environment")
    const char* dp = std::getenv("SOBFU_TILED_DRY_PACKED");
    t->dry_packed = dp && dp[0] == '1';
    const char* fc = std::getenv("SOBFU_TILED_FORCE_COMM");
    t->force_comm = fc && fc[0] == '1';
    const char* ds = std::getenv("SOBFU_TILED_DEBUG_SKIP");
    t->debug_skip = ds ? std::atoi(ds) : 0;
    t->z0 = t->g0[2]; t->z1 = t->g1[2]; t->Lz = t->L[2]; t->own_lo = t->o0[2]; t->own_hi = t->o1[2]; t->zbase = t->base[2];
    t->NL = (size_t) t->L[0] * t->L[1] * t->L[2];
    t->NF = (size_t) X * Y * Z;
    t->p  = *params;
    float h[16];
    if (rc == 0) rc = sobfu_hip_sobolev_filter(params->s, params->lambda, h);
    for (int i = 0; i < 7; ++i) t->taps[i] = h[i];
    if (rc == 0) {
        size_t off = 0;
        t->geom = lay.msgs;
        for (const MsgGeom& g : lay.msgs) {
            sobfu_hip_tiled_msg m;
            m.peer     = g.peer;
            m.send_off = m.recv_off = off;
            m.count    = g.cells * 3;
            off += m.count;
            t->msgs.push_back(m);
            t->sboxes.insert(t->sboxes.end(), g.sb, g.sb + 6);
            t->rboxes.insert(t->rboxes.end(), g.rb, g.rb + 6);
            const bool zface = g.dir[0] == 0 && g.dir[1] == 0 && g.dir[2] != 0;
            if (zface && !t->slab) {  // the neighbour across a z face shares this tile's x / y layout: whole planes of the array match
                const size_t plane_f = (size_t) t->L[0] * t->L[1] * 3;
                t->zmsgs.push_back(sobfu_hip_tiled_msg{g.peer, plane_f * (size_t) g.sb[4], plane_f * (size_t) g.rb[4], plane_f * (size_t) kHalo});
            } else if (!zface) {
                t->n_packed += 1;  // (z faces are last in the list)
            }
        }
        if (t->slab) t->n_packed = (int) t->msgs.size();  // (slabs never take the packed path; keep the count meaningful)
        if (off > 0) {
            rc = (int) hipMalloc((void**) &t->sendbuf, off * sizeof(float));
            if (rc == 0) rc = (int) hipMalloc((void**) &t->recvbuf, off * sizeof(float));
        }
        if (rc == 0 && !t->slab && t->n_packed > 0) {  // where every cell of the packed receive buffer goes (x fastest inside a message box)
            std::vector<uint32_t> tab;
            for (int i = 0; i < t->n_packed; ++i) {
                const int* b = t->rboxes.data() + 6 * i;
                for (int z = b[4]; z < b[5]; ++z)
                    for (int y = b[2]; y < b[3]; ++y)
                        for (int x = b[0]; x < b[1]; ++x) tab.push_back((uint32_t) ((size_t) x + (size_t) t->L[0] * ((size_t) y + (size_t) t->L[1] * (size_t) z)));
            }
            t->scatter_cells = (unsigned) tab.size();
            rc = (int) hipMalloc((void**) &t->scatter_table, tab.size() * sizeof(uint32_t));
            if (rc == 0) rc = (int) hipMemcpy(t->scatter_table, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        }
    }
    if (rc == 0) {
        auto up = [](size_t v) { return (v + 4095) & ~(size_t) 4095; };
        t->nu_off[0] = 0;
        t->nu_off[1] = up(t->NL * 12);
        t->rows_off  = t->nu_off[1] + up(t->NL * 12);
        t->arena_bytes = std::max(t->rows_off + up((size_t) (kRowsIters + 2) * kSlots * 4), kOwnBlock);
        rc = pool_take((void**) &t->arena, t->arena_bytes, false);
        if (rc == 0) {
            t->nUb[0] = (float*) (t->arena + t->nu_off[0]);
            t->nUb[1] = (float*) (t->arena + t->nu_off[1]);
            t->grows  = (uint32_t*) (t->arena + t->rows_off);
            // halo cells no message fills (tile corners) stay finite; rows start at "converged nowhere" = 0 (blocking: the loop's
            // streams do not order against the null stream)
            rc = (int) hipMemset(t->arena, 0, t->arena_bytes);
        }
    }
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_psi, t->NL * 12);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_f, t->NL * 4);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_psi2, t->NL * 12);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_f2, t->NL * 4);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_g, t->NL * 4);
    if (rc == 0) rc = (int) hipMalloc((void**) &t->c_n, t->NF * 4);
    if (rc == 0) {  // max-norm slot rows for kRowsIters iterations up front: a solve never reallocates inside a timed region
        rc = (int) hipMalloc((void**) &t->slots, (size_t) (kRowsIters + 1) * kSlots * 4);
        if (rc == 0) t->slots_iters = kRowsIters;
    }
    if (rc == 0) {
        // arrival flags: written by the peers over xGMI, polled here -- uncached memory, so that neither side's L2 sits between
        // a store and the poll (plain device memory if the runtime refuses)
        rc = pool_take((void**) &t->flags, kOwnBlock, true);
        if (rc == 0) rc = (int) hipMemset(t->flags, 0, kMaxSync * sizeof(uint32_t));
    }
    if (rc == 0) rc = (int) hipMalloc((void**) &t->sync_d, sizeof(sobfu_hip::TileSync));
    if (rc == 0) {
        sobfu_hip::TileSync sy{};
        sy.my_rank = (uint32_t) rank; sy.world = (uint32_t) t->world;
        t->timeout_ticks = (uint64_t) (deadline_seconds() * 1e8);
        sy.timeout_ticks = t->timeout_ticks;
        sy.my_flags = t->flags; sy.my_grows = t->grows;
        rc = (int) hipMemcpy(t->sync_d, &sy, sizeof sy, hipMemcpyHostToDevice);
    }
    if (rc == 0) build_a_boxes(t, nullptr, nullptr, nullptr);
    if (rc == 0) rc = (int) hipStreamCreateWithFlags(&t->comm_stream, hipStreamNonBlocking);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_bnd, hipEventDisableTiming);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_xchg, hipEventDisableTiming);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_first, hipEventDisableTiming);
    // the max-norm rows are written by pass B's atomics on this device, but a real RCCL all-reduce may let PEERS write the
    // reduced row straight into this buffer (direct / registered-buffer paths): every event keeps the system-scope fence
    // (a fence-less variant -- an L2 write-back + invalidate less per edge -- would have to be validated on >= 2 real GPUs first).
    const unsigned local_ev = hipEventDisableTiming;
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_red[0], local_ev);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_red[1], local_ev);
    if (rc == 0) rc = (int) hipEventCreateWithFlags(&t->ev_row, local_ev);
    if (rc == 0 && !dry) {
        ncclUniqueId id;
        std::memcpy(&id, unique_id, 128);
        ncclResult_t r = g_rccl.CommInitRank(&t->comm, t->world, id, rank);
        if (r != ncclSuccess) {
            std::fprintf(stderr, "sobfu_hip: ncclCommInitRank failed: %s\n", g_rccl.GetErrorString(r));
            rc = SOBFU_E_RCCL;
        }
    }
    // hipMemset / hipMemcpy on device memory may return before they have run.  The arrays cleared above are about to be mapped and
    // written by OTHER processes, which no stream of this one orders against: a clear that ran late would wipe a peer's first
    // arrival flag (seen: one hang in ~10 start-ups with four ranks sharing a GPU) -- drain the device before anybody can see them.
    if (rc == 0) rc = (int) hipDeviceSynchronize();
    if (rc != 0) {
        sobfu_hip_tiled_destroy(t);
        return rc;
    }
    *out = t;
    return 0;
}

int sobfu_hip_tiled_create(sobfu_hip_tiled** out, int X, int Y, int Z, int world, int rank, const char unique_id[128],
                           const sobfu_hip_solver_params* params) {
    return sobfu_hip_tiled_create3(out, X, Y, Z, 1, 1, world, rank, unique_id, params);  // z-slabs
}

// ---- direct transport -------------------------------------------------------------------------------------------------------
int sobfu_hip_ipc_export(const void* d_ptr, char handle[64]) {
    SOBFU_CHECK_ARGS(d_ptr && handle);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (const PooledBlock& b : g_pool)  // the library's own blocks were exported when they were allocated
            if (b.ptr == d_ptr && b.has_handle) {
                std::memcpy(handle, &b.handle, 64);
                return 0;
            }
    }
    hipIpcMemHandle_t h;
    SOBFU_HIP_TRY(hipIpcGetMemHandle(&h, const_cast<void*>(d_ptr)));
    std::memcpy(handle, &h, 64);
    return 0;
}
int sobfu_hip_ipc_open(const char handle[64], void** d_ptr) {
    SOBFU_CHECK_ARGS(handle && d_ptr);
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, 64);
    SOBFU_HIP_TRY(hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
int sobfu_hip_ipc_close(void* d_ptr) {
    if (!d_ptr) return 0;
    SOBFU_HIP_TRY(hipIpcCloseMemHandle(d_ptr));
    return 0;
}

int sobfu_hip_tiled_exports_get(const sobfu_hip_tiled* t, sobfu_hip_tiled_exports* out) {
    SOBFU_CHECK_ARGS(t && out);
    out->arena = t->arena; out->nabla_u_off[0] = t->nu_off[0]; out->nabla_u_off[1] = t->nu_off[1]; out->rows_off = t->rows_off;
    out->flags = t->flags;
    return 0;
}

int sobfu_hip_tiled_connect(sobfu_hip_tiled* t, int n_peers, const int* peer_ranks, const sobfu_hip_tiled_exports* peers) {
    SOBFU_CHECK_ARGS(t && n_peers >= 0 && (n_peers == 0 || (peer_ranks && peers)) && !t->q.active && !t->comm);
    if (t->world > kMaxSync) return SOBFU_E_UNSUPPORTED;
    // Cells other GPUs stored are read at system scope, which only the pipelined march of pass B does (launch_pass_b_boxes): it needs
    // 32-bit gather offsets (phi_n below 2^30 voxels) and buffer addressing (the local arrays below 4 GiB).  Refused HERE, not by a
    // launch in the middle of a solve whose pass A has already pushed into the peers.
    if ((size_t) t->X * t->Y * t->Z >= ((size_t) 1 << 30) || (size_t) t->L[0] * t->L[1] * t->L[2] * 12 >= ((size_t) 1 << 32)) return SOBFU_E_UNSUPPORTED;
    if (t->grows_own) {  // a longer solve on the unconnected handle moved the rows to a private array: back to the exported ones
        SOBFU_HIP_TRY(hipFree(t->grows_own));
        if (t->slots) SOBFU_HIP_TRY(hipFree(t->slots));
        t->grows_own = t->slots = nullptr;
        SOBFU_HIP_TRY(hipMalloc((void**) &t->slots, (size_t) (kRowsIters + 1) * kSlots * 4));
        t->slots_iters = kRowsIters;
        t->grows       = (uint32_t*) (t->arena + t->rows_off);
    }
    const int dims[3] = {t->X, t->Y, t->Z};
    auto find = [&](int r) -> const sobfu_hip_tiled_exports* {
        for (int i = 0; i < n_peers; ++i)
            if (peer_ranks[i] == r) return &peers[i];
        return nullptr;
    };
    // push destinations: every message needs its peer
    std::vector<float*> d0(t->geom.size()), d1(t->geom.size());
    std::vector<TileLay> pl;
    for (size_t i = 0; i < t->geom.size(); ++i) {
        const sobfu_hip_tiled_exports* e = find(t->geom[i].peer);
        if (!e || !e->arena || !e->flags) return SOBFU_E_BADARG;
        d0[i] = (float*) ((char*) e->arena + e->nabla_u_off[0]);
        d1[i] = (float*) ((char*) e->arena + e->nabla_u_off[1]);
        pl.push_back(make_layout(dims, t->P, t->geom[i].peer));
    }
    // sync set: EVERY other rank -- the halo neighbours for the nabla_U cells, the rest because the max-norm rows become global by
    // every rank storing its row maximum at every other rank (world <= 64: a few 4-byte stores per iteration)
    sobfu_hip::TileSync sy{};
    sy.my_rank = (uint32_t) t->rank; sy.world = (uint32_t) t->world;
    sy.timeout_ticks = t->timeout_ticks;
    sy.my_flags = t->flags; sy.my_grows = t->grows;
    for (int r = 0; r < t->world; ++r) {
        if (r == t->rank) continue;
        const sobfu_hip_tiled_exports* e = find(r);
        if (!e || !e->flags || !e->arena) return SOBFU_E_BADARG;
        sy.sync_rank[sy.n_sync]  = r;
        sy.peer_flags[sy.n_sync] = (uint32_t*) e->flags;
        sy.peer_grows[sy.n_sync] = (uint32_t*) ((char*) e->arena + e->rows_off);
        sy.n_sync += 1;
    }
    SOBFU_HIP_TRY(hipMemcpy(t->sync_d, &sy, sizeof sy, hipMemcpyHostToDevice));
    SOBFU_HIP_TRY(hipDeviceSynchronize());
    build_a_boxes(t, d0.data(), d1.data(), pl.data());
    t->direct = true;
    return 0;
}

int sobfu_hip_tiled_max_iterations(const sobfu_hip_tiled* t) { return t ? (t->direct ? kRowsIters : 0x7fffffff) : 0; }

int sobfu_hip_p2p_info(int device, int peer_device, int out[4]) {
    SOBFU_CHECK_ARGS(out && device >= 0 && peer_device >= 0);
    int can = 0, rank = -1;
    uint32_t link = 0, hops = 0;
    out[0] = out[1] = out[2] = out[3] = -1;
    if (device == peer_device) {
        out[0] = 1; out[1] = 0; out[2] = 0; out[3] = 0;
        return 0;
    }
    if (hipDeviceCanAccessPeer(&can, device, peer_device) != hipSuccess) (void) hipGetLastError(); else out[0] = can;
    if (hipExtGetLinkTypeAndHopCount(device, peer_device, &link, &hops) != hipSuccess) (void) hipGetLastError();
    else { out[1] = (int) link; out[2] = (int) hops; }
    if (hipDeviceGetP2PAttribute(&rank, hipDevP2PAttrPerformanceRank, device, peer_device) != hipSuccess) (void) hipGetLastError(); else out[3] = rank;
    return 0;
}

int sobfu_hip_tiled_wait_stats(sobfu_hip_tiled* t, double* wait_us, int* waits, int reset) {
    SOBFU_CHECK_ARGS(t);
    sobfu_hip::TileSync sy;
    SOBFU_HIP_TRY(hipMemcpy(&sy, t->sync_d, sizeof sy, hipMemcpyDeviceToHost));
    if (wait_us) *wait_us = (double) sy.wait_ticks / 100.0;  // wall_clock64: 100 MHz
    if (waits) *waits = (int) sy.wait_count;
    if (reset) {
        const uint64_t z[2] = {0, 0};
        SOBFU_HIP_TRY(hipMemcpy(&t->sync_d->wait_ticks, z, 16, hipMemcpyHostToDevice));
    }
    return 0;
}

// diagnostics of a CONNECTED handle outside a solve; every rank of the world makes the same calls in the same order (the sequence
// numbers of the arrival flags advance in step)
int sobfu_hip_tiled_pingpong(sobfu_hip_tiled* t, int rank_a, int rank_b, int reps, void* stream) {
    SOBFU_CHECK_ARGS(t && t->direct && !t->q.active && reps > 0 && rank_a != rank_b && rank_a >= 0 && rank_b >= 0 && rank_a < t->world && rank_b < t->world);
    if (t->dead) return SOBFU_E_TIMEOUT;
    const uint32_t seq0 = t->seq_total + 1u;
    t->seq_total += 2u * (uint32_t) reps;
    if (t->rank != rank_a && t->rank != rank_b) return 0;
    const int other = t->rank == rank_a ? rank_b : rank_a;
    const int q     = other < t->rank ? other : other - 1;  // the sync set lists every other rank in rank order
    return sobfu_hip::launch_tile_pingpong(t->sync_d, q, t->rank == rank_a ? 1 : 0, seq0, reps, (hipStream_t) stream);
}

int sobfu_hip_tiled_probe_push(sobfu_hip_tiled* t, int reps, void* stream) {
    SOBFU_CHECK_ARGS(t && (t->direct || (!t->comm && !t->xfn && !t->dry_packed)) && !t->q.active && reps > 0);  // handles whose pass A signals
    if (t->dead) return SOBFU_E_TIMEOUT;
    for (int r = 0; r < reps; ++r) {
        const uint32_t seq = ++t->seq_total;
        const std::vector<sobfu_hip::TileLaunchBox>& bx = t->a_boxes[seq & 1u];
        if (bx.size() < 2) continue;  // no messages
        // pass A's push boxes alone, on whatever the compact state holds (the halo rims they fill are scratch between solves)
        SOBFU_TRY(sobfu_hip::launch_tile_pass_a(t->c_f, t->c_g, t->c_psi, t->nUb[seq & 1u], t->p.w_reg, t->L[0], t->L[1], t->L[2], bx.data(),
                                                (int) bx.size() - 1, t->sync_d, seq, t->wait_enabled, nullptr, 0, 0, (hipStream_t) stream, true));
    }
    return 0;
}

int sobfu_hip_tiled_set_wait(sobfu_hip_tiled* t, int wait) {
    SOBFU_CHECK_ARGS(t);
    t->wait_enabled = wait ? 1 : 0;
    return 0;
}

int sobfu_hip_tiled_status(sobfu_hip_tiled* t, int* missing_peer) {
    SOBFU_CHECK_ARGS(t);
    uint32_t err = 0;
    SOBFU_HIP_TRY(hipMemcpy(&err, &t->sync_d->err, 4, hipMemcpyDeviceToHost));
    if (missing_peer) *missing_peer = err ? (int) err - 1 : -1;
    return (err || t->dead) ? SOBFU_E_TIMEOUT : 0;
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

int sobfu_hip_tiled_set_profiling(sobfu_hip_tiled* t, int stride, int max_samples) {
    SOBFU_CHECK_ARGS(t && stride >= 0 && max_samples >= 0);
    t->prof_stride = stride;
    while (stride > 0 && t->prof_ev.size() < (size_t) 4 * max_samples) {  // created here, never inside a timed region
        hipEvent_t e;
        SOBFU_HIP_TRY(hipEventCreate(&e));
        t->prof_ev.push_back(e);
    }
    return 0;
}

int sobfu_hip_tiled_get_profile(sobfu_hip_tiled* t, float ms[3], int* samples, int reset) {
    SOBFU_CHECK_ARGS(t && ms);
    for (int k = 0; k < t->prof_pending; ++k) {  // the caller has synchronised the stream
        for (int j = 0; j < 3; ++j) {
            float v = 0;
            SOBFU_HIP_TRY(hipEventElapsedTime(&v, t->prof_ev[4 * k + j], t->prof_ev[4 * k + j + 1]));
            t->prof_ms[j] += v;
        }
        t->prof_n += 1;
    }
    t->prof_pending = 0;
    for (int j = 0; j < 3; ++j) ms[j] = (float) t->prof_ms[j];
    if (samples) *samples = t->prof_n;
    if (reset) {
        t->prof_ms[0] = t->prof_ms[1] = t->prof_ms[2] = 0;
        t->prof_n = 0;
    }
    return 0;
}

int sobfu_hip_tiled_layout(const sobfu_hip_tiled* t, int* z0, int* z1, int* lo, int* hi, int* Lz, int* zbase) {
    SOBFU_CHECK_ARGS(t);
    if (z0) *z0 = t->z0;
    if (z1) *z1 = t->z1;
    if (lo) *lo = t->lo[2];
    if (hi) *hi = t->hi[2];
    if (Lz) *Lz = t->Lz;
    if (zbase) *zbase = t->zbase;
    return 0;
}

int sobfu_hip_tiled_layout3(const sobfu_hip_tiled* t, int out[24]) {
    SOBFU_CHECK_ARGS(t && out);
    for (int a = 0; a < 3; ++a) {
        out[a] = t->P[a]; out[3 + a] = t->c[a]; out[6 + a] = t->g0[a]; out[9 + a] = t->g1[a]; out[12 + a] = t->lo[a];
        out[15 + a] = t->hi[a]; out[18 + a] = t->L[a]; out[21 + a] = t->base[a];
    }
    return 0;
}

int sobfu_hip_tiled_messages(const sobfu_hip_tiled* t, sobfu_hip_tiled_msg* msgs, int* send_boxes, int* recv_boxes, int max_msgs) {
    if (!t) return SOBFU_E_BADARG;
    const int n = (int) t->msgs.size();
    for (int i = 0; i < n && i < max_msgs; ++i) {
        if (msgs) msgs[i] = t->msgs[i];
        if (send_boxes) std::memcpy(send_boxes + 6 * i, t->sboxes.data() + 6 * i, 6 * sizeof(int));
        if (recv_boxes) std::memcpy(recv_boxes + 6 * i, t->rboxes.data() + 6 * i, 6 * sizeof(int));
    }
    return n;  // number of messages of an exchange
}

int sobfu_hip_tiled_messages_inplace(const sobfu_hip_tiled* t, sobfu_hip_tiled_msg* msgs, int max_msgs, int* n_packed) {
    if (!t) return SOBFU_E_BADARG;
    if (n_packed) *n_packed = t->n_packed;
    const int n = (int) t->zmsgs.size();
    for (int i = 0; i < n && i < max_msgs; ++i)
        if (msgs) msgs[i] = t->zmsgs[i];
    return n;
}

// Delivers the messages of one exchange from d_send to d_recv (RCCL grouped send/recv, or the user transport) and, in the same RCCL
// group, a second list with its own base pointers (the in-place z faces of a 3-D tile: both bases are the field array itself).
static int transfer(sobfu_hip_tiled* t, const float* d_send, float* d_recv, const sobfu_hip_tiled_msg* msgs, int n, hipStream_t stream,
                    const float* d_send2 = nullptr, float* d_recv2 = nullptr, const sobfu_hip_tiled_msg* msgs2 = nullptr, int n2 = 0) {
    if (n + n2 == 0) return 0;
    if (!t->comm) {  // user transport / dry handle: one call per list (a transport sees at most one message per peer in a call)
        if (!t->xfn) return 0;
        if (n > 0) SOBFU_TRY(t->xfn(t->tctx, t->rank, d_send, d_recv, msgs, n, (void*) stream));
        if (n2 > 0) SOBFU_TRY(t->xfn(t->tctx, t->rank, d_send2, d_recv2, msgs2, n2, (void*) stream));
        return 0;
    }
    RCCL_TRY(g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
        RCCL_TRY(g_rccl.Send(d_send + msgs[i].send_off, msgs[i].count, ncclFloat32, msgs[i].peer, t->comm, stream));
        RCCL_TRY(g_rccl.Recv(d_recv + msgs[i].recv_off, msgs[i].count, ncclFloat32, msgs[i].peer, t->comm, stream));
    }
    for (int i = 0; i < n2; ++i) {
        RCCL_TRY(g_rccl.Send(d_send2 + msgs2[i].send_off, msgs2[i].count, ncclFloat32, msgs2[i].peer, t->comm, stream));
        RCCL_TRY(g_rccl.Recv(d_recv2 + msgs2[i].recv_off, msgs2[i].count, ncclFloat32, msgs2[i].peer, t->comm, stream));
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

// z-slab schedules: `planes` owned planes per interior face of a slab field travel in place (no copy on either side)
static int exchange_planes(sobfu_hip_tiled* t, float* field3, int planes, hipStream_t stream) {
    const size_t plane_f = (size_t) t->X * t->Y * 3, cnt = plane_f * planes;
    sobfu_hip_tiled_msg m[2];
    int n = 0;
    if (t->rank > 0) m[n++] = {t->rank - 1, plane_f * t->own_lo, plane_f * (t->own_lo - planes), cnt};
    if (t->rank < t->world - 1) m[n++] = {t->rank + 1, plane_f * (t->own_hi - planes), plane_f * t->own_hi, cnt};
    return transfer(t, field3, field3, m, n, stream);
}

// tile path, RCCL / callback transports: the send buffer (filled by pass A's push boxes) -> peers -> scatter into the halo cells
// The z FACES do not go through the buffers: the neighbour across a z face has the same x / y layout, so 4 whole padded planes of the
// array (Lx x Ly x 4 cells: + 6 % bytes on a 132^2 plane) leave straight out of the owned rim and land straight in the halo planes --
// no push box packs them, the scatter kernel does not touch them.  What arrives in those planes' x / y halo columns is the sender's own
// (stale) halo content: exactly the cells of the xz / yz EDGE STRIPS, which the scatter -- behind the transfer on the same stream --
// overwrites with the diagonal neighbours' cells; the corner regions beyond are never read (every stencil is axis-aligned).
static int exchange_packed(sobfu_hip_tiled* t, float* field3, hipStream_t stream) {
    const int n = t->n_packed, nz = (int) t->zmsgs.size();
    if (n + nz == 0) return 0;
    SOBFU_TRY(transfer(t, t->sendbuf, t->recvbuf, t->msgs.data(), n, stream, field3, field3, t->zmsgs.data(), nz));
    if (t->scatter_table) return sobfu_hip::launch_msg_scatter_table(field3, t->recvbuf, t->scatter_table, t->scatter_cells, stream);
    return sobfu_hip::launch_msg_copy(false, field3, t->recvbuf, t->L[0], t->L[1], t->L[2], t->rboxes.data(), n, stream);
}

static int allreduce_max(sobfu_hip_tiled* t, uint32_t* buf, size_t n, hipStream_t stream, bool own_comm = false) {
    if (!t->comm) return t->rfn ? t->rfn(t->tctx, t->rank, buf, n, (void*) stream) : 0;
    RCCL_TRY(g_rccl.AllReduce(buf, buf, n, ncclUint32, ncclMax, own_comm ? t->comm2 : t->comm, stream));
    return 0;
}

// Debug / bring-up: one halo exchange of a caller-provided 12-byte tile field exactly as the loop's RCCL / callback transports do
// it (z-slabs: `planes` planes in place; 3-D tiles: pack, transfer, scatter).
int sobfu_hip_tiled_exchange(sobfu_hip_tiled* t, float* d_field3, int planes, void* stream) {
    SOBFU_CHECK_ARGS(t && d_field3 && planes > 0 && planes <= kHalo && (t->slab || planes == kHalo));
    if (t->slab) return exchange_planes(t, d_field3, planes, (hipStream_t) stream);
    const int n = t->n_packed;  // (the z faces travel in place: nothing to pack)
    if (n + (int) t->zmsgs.size() == 0) return 0;
    SOBFU_TRY(sobfu_hip::launch_msg_copy(true, d_field3, t->sendbuf, t->L[0], t->L[1], t->L[2], t->sboxes.data(), n, (hipStream_t) stream));
    return exchange_packed(t, d_field3, (hipStream_t) stream);
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

// The gradient-descent loop (reference src/sobfu/cuda/solver.cu:106-193) on this rank's tile, in three pieces (begin / step / end;
// sobfu_hip_tiled_iterate = all three).  API-format arguments: d_phi_global_local / d_phi_n_psi_local float2 (Lx, Ly, Lz),
// d_phi_n_full float2 (X, Y, Z), d_psi_local float4 (Lx, Ly, Lz) whose owned cells and one-cell shells are exact on entry
// (identity: sobfu_hip_tile3_init_identity) and on exit.
//
// Convergence without a stall: psi and F = (phi_n o psi).tsdf are PING-PONGED (iteration k reads buffer (k-1)&1 and writes
// buffer k&1), and the device-side gate of iteration k looks at the max-norm row of iteration k-2.  When the threshold fires
// at iteration k, iteration k+1 has already run speculatively -- into the OTHER buffer -- every later launch is a no-op,
// and the state the reference's `break` (solver.cu:183) leaves is intact in buffer k&1.  The reduction that makes row k
// global therefore has a whole iteration to complete: on the direct transport it rides on pass A of iteration k+1 (every rank
// stores its row maximum at every other rank, covered by that iteration's arrival flags); on the others it is an all-reduce
// behind the exchange.  Nothing in the loop ever waits for a reduction that is still in flight (a same-iteration gate costs an
// exposed collective or a stream round trip per iteration: 67 vs 55 us per iteration in the round-2 N = 8 compute-side timing).
// a bare dry handle (no communicator, no transport plugged in) times the direct schedule -- or, with SOBFU_TILED_DRY_PACKED=1 at
// create, the launches of the RCCL / callback transports (pass A packing into the send buffer, the scatter kernel, pass B)
static bool uses_sync(const sobfu_hip_tiled* t) { return t->direct || (!t->comm && !t->xfn && !t->dry_packed); }
static bool tile_path(const sobfu_hip_tiled* t) { return !t->slab || uses_sync(t); }

static int tiled_begin(sobfu_hip_tiled* t, const float* d_phi_global_local, const float* d_phi_n_full, float* d_phi_n_psi_local,
                       float* d_psi_local, int max_iters, hipStream_t st) {
    sobfu_hip_tiled::Session& q = t->q;
    if (q.active) return SOBFU_E_BADARG;
    if (t->dead) return SOBFU_E_TIMEOUT;
    const int X = t->X, Y = t->Y, Z = t->Z, Lx = t->L[0], Ly = t->L[1], Lz = t->L[2];
    float* P[2] = {t->c_psi, t->c_psi2};
    float* F[2] = {t->c_f, t->c_f2};
    if (max_iters > t->slots_iters) {
        if (t->direct) {  // the global rows are mapped by the peers: their number is fixed (sobfu_hip_tiled_max_iterations)
            std::fprintf(stderr, "sobfu_hip: a solve of %d iterations exceeds the %d the direct transport's peer-mapped max-norm rows hold; "
                                 "split the solve or use the RCCL transport\n", max_iters, t->slots_iters);
            return SOBFU_E_UNSUPPORTED;
        }
        if (t->slots) SOBFU_HIP_TRY(hipFree(t->slots));
        if (t->grows_own) SOBFU_HIP_TRY(hipFree(t->grows_own));
        t->slots = t->grows_own = nullptr;
        t->slots_iters = 0;
        SOBFU_HIP_TRY(hipMalloc((void**) &t->slots, (size_t) (max_iters + 1) * kSlots * 4));
        SOBFU_HIP_TRY(hipMalloc((void**) &t->grows_own, (size_t) (max_iters + 2) * kSlots * 4));
        SOBFU_HIP_TRY(hipMemset(t->grows_own, 0, (size_t) (max_iters + 2) * kSlots * 4));
        t->grows = t->grows_own;
        SOBFU_HIP_TRY(hipMemcpy(&t->sync_d->my_grows, &t->grows, sizeof(uint32_t*), hipMemcpyHostToDevice));
        t->slots_iters = max_iters;
    }
    sobfu_hip_tiled::Session n{};
    n.pn = d_phi_n_full; n.pnp = d_phi_n_psi_local; n.psi = d_psi_local;
    n.cap = max_iters;
    n.seq_base = t->seq_total;
    // enter the compact format (includes the warp of solver.cu:106); both halves start equal so that cells no launch
    // writes (beyond the one-cell shells) hold the caller's values whichever half the loop ends in
    SOBFU_TRY(sobfu_hip::launch_pack_vec(d_psi_local, P[0], t->NL, st));
    SOBFU_TRY(sobfu_hip::launch_extract_tsdf(d_phi_global_local, t->c_g, t->NL, st));
    SOBFU_TRY(sobfu_hip::launch_extract_tsdf(d_phi_n_full, t->c_n, t->NF, st));
    SOBFU_TRY(sobfu_hip::launch_apply_tsdf_only(t->c_n, F[0], P[0], Lx, Ly, Lz, st, Z, X, Y));
    SOBFU_HIP_TRY(hipMemcpyAsync(P[1], P[0], t->NL * 12, hipMemcpyDeviceToDevice, st));
    SOBFU_HIP_TRY(hipMemcpyAsync(F[1], F[0], t->NL * 4, hipMemcpyDeviceToDevice, st));
    if (max_iters > 0) SOBFU_HIP_TRY(hipMemsetAsync(t->slots, 0, (size_t) (max_iters + 1) * kSlots * 4, st));
    q        = n;  // the session is open only once everything above has been enqueued
    q.active = true;
    return 0;
}

// the first iteration of a communicator's life, under a host-side deadline: a wedged rank says what it is waiting for and
// aborts its communicators instead of sitting in a collective until somebody's process-group timeout
static int first_iteration_watchdog(sobfu_hip_tiled* t, hipStream_t st) {
    t->first_checked = true;
    SOBFU_HIP_TRY(hipEventRecord(t->ev_first, st));
    const double limit = deadline_seconds();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(t->ev_first);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return (int) e;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    std::fprintf(stderr, "sobfu_hip: rank %d: the first tiled iteration has not completed after %.0f s -- waiting for the halo exchange "
                         "(grouped ncclSend/ncclRecv with ranks", t->rank, limit);
    for (const sobfu_hip_tiled_msg& m : t->msgs) std::fprintf(stderr, " %d", m.peer);
    std::fprintf(stderr, ")%s; aborting the communicator(s)\n", t->p.max_update_norm >= 0.f ? " or the max-norm all-reduce" : "");
    abort_comms(t);
    return SOBFU_E_TIMEOUT;
}

// phases: 1 = pass A (+ the exchange of the RCCL / callback transports), 2 = pass B (+ the row reduction); 3 = a whole iteration
static int tiled_step_impl(sobfu_hip_tiled* t, int n_steps, hipStream_t st, int phases) {
    sobfu_hip_tiled::Session& q = t->q;
    if (!q.active || n_steps < 0 || q.launched + n_steps > q.cap) return SOBFU_E_BADARG;
    if (t->dead) return SOBFU_E_TIMEOUT;
    const int X = t->X, Y = t->Y, Z = t->Z, Lx = t->L[0], Ly = t->L[1], Lz = t->L[2];
    const sobfu_hip_solver_params& p = t->p;
    float* P[2] = {t->c_psi, t->c_psi2};
    float* F[2] = {t->c_f, t->c_f2};
    const int n_iters = q.cap;  // the last iteration this solve can reach (rows past cap - 2 gate nothing)
    // (SOBFU_TILED_FORCE_COMM=1 at create runs the communication choreography -- streams, events, empty exchange group, world-1
    // all-reduce -- on a single rank too: bring-up / test hook for 1-GPU machines)
    const bool can_converge = p.max_update_norm >= 0.f, multi = t->world > 1 || t->force_comm;
    const int lo = t->own_lo, hi = t->own_hi, H = kHalo;
    const int a_lo = t->lo[2] ? std::min(lo + H, hi) : lo, a_hi = t->hi[2] ? std::max(hi - H, a_lo) : hi;
    const int b_lo = t->lo[2] ? std::min(lo + 3, hi) : lo, b_hi = t->hi[2] ? std::max(hi - 3, b_lo) : hi;
    const int b_first = t->lo[2] ? lo - 1 : lo, b_last = t->hi[2] ? hi + 1 : hi;
    const int ax0 = t->o0[0], ax1 = t->o1[0], ay0 = t->o0[1], ay1 = t->o1[1];
    const int own[6] = {ax0, ax1, ay0, ay1, lo, hi};
    const bool tiles = tile_path(t), sync = uses_sync(t);
    // timing experiments only (results are wrong): bit 0 no push boxes, bit 1 no thin shells, bit 2 no pass A, bit 3 no pass B,
    // bit 4 no owned block in pass B, bit 5 no y shells, bit 6 no x shells; in build_a_boxes: bit 7 no thin push boxes (x faces, edges,
    // corners), bit 8 no marched push boxes (y / z faces), bit 9 the marched push boxes store at home only
    const bool b_direct = false;
    const int dbg = t->debug_skip;
    // pass A split into boundary + interior launches so that the exchange starts after 4 planes per face instead of after
    // the whole pass: an extra launch (+6-7 us per iteration in the compute-only timing at N = 4 and 8), worth it only where
    // the slab is so thin that the 3.1 MB face messages cannot hide behind B_int alone (N >= 4 at 256^3, if a face takes the ~65 us that ~60 GB/s per xGMI direction implies)
    // the schedule (results do not depend on it): sobfu_hip_tiled_set_schedule (autotuner, tests) > heuristic
    const bool want_split = t->schedule == 1 ? true : (t->schedule == 2 ? false : (hi - lo) <= kSplitAMaxPlanes);
    const bool split_a = (t->lo[2] || t->hi[2]) && a_hi > a_lo && want_split;
    const bool serial = t->schedule == 3;  // z-slab path only
    // Where the all-reduce of a max-norm row runs on the RCCL / callback transports (the late gate gives row j until pass B of
    // iteration j+2): in line behind pass B (serial / tile path), on the comm stream behind the next exchange (overlapped slab
    // schedules), or -- opt-in, sobfu_hip_tiled_add_reduce_comm -- on a communicator and stream of its own.
    enum { RED_NONE, RED_INLINE, RED_COMM_STREAM, RED_OWN_COMM };
    const int red_mode = (!multi || !can_converge || sync) ? RED_NONE : (t->comm2 ? RED_OWN_COMM : ((serial || tiles) ? RED_INLINE : RED_COMM_STREAM));
    bool* red_issued = q.red_issued;  // an asynchronous reduce of the latest row of this parity is behind ev_red[parity]
    const auto host_t0 = std::chrono::steady_clock::now();
    const int last = q.launched + n_steps;
    for (int it = q.launched + 1; it <= last; ++it) {
        const float *psi_in = P[(it - 1) & 1], *f_in = F[(it - 1) & 1];
        float *psi_out = P[it & 1], *f_out = F[it & 1];
        float* nu = t->nUb[tiles ? (it & 1) : 0];
        const uint32_t* rows = sync ? t->grows : t->slots;  // the rows the gate reads: global by the time it does
        const uint32_t* prev = (it > 2 && can_converge) ? rows + (size_t) (it - 2) * kSlots : nullptr;  // the late gate
        uint32_t* row        = t->slots + (size_t) it * kSlots;
        auto A = [&](int za, int zb, int za2 = 0, int zb2 = 0) {  // pass A writes scratch only: never gated
            const sobfu_hip::LaunchBox bx[2] = {{ax0, ax1, ay0, ay1, za, zb, false}, {ax0, ax1, ay0, ay1, za2, zb2, false}};
            return sobfu_hip::launch_pass_a_boxes(f_in, t->c_g, psi_in, nu, p.w_reg, Lx, Ly, Lz, bx, 2, nullptr, 0.f, 0, st, true);
        };
        auto B = [&](int za, int zb, int za2 = 0, int zb2 = 0, bool shells = false) {
            // the owned block with its z shells (extra planes of the march); the one-cell x / y shells are direct boxes
            const bool ysh = shells && !(dbg & 32), xsh = shells && !(dbg & 64);
            const sobfu_hip::LaunchBox bx[6] = {{ax0, ax1, ay0, ay1, za, (dbg & 16) ? za : zb, b_direct}, {ax0, ax1, ay0, ay1, za2, zb2, b_direct},
                                                {ax0, ax1, ay0 - 1, (ysh && t->lo[1]) ? ay0 : ay0 - 1, lo, hi, true},
                                                {ax0, ax1, ay1, (ysh && t->hi[1]) ? ay1 + 1 : ay1, lo, hi, true},
                                                {ax0 - 1, (xsh && t->lo[0]) ? ax0 : ax0 - 1, ay0, ay1, lo, hi, true},
                                                {ax1, (xsh && t->hi[0]) ? ax1 + 1 : ax1, ay0, ay1, lo, hi, true}};
            return sobfu_hip::launch_pass_b_boxes(nu, const_cast<float*>(psi_in), t->c_n, f_out, nullptr, row, t->taps, p.alpha, Lx, Ly, Lz, X, Y,
                                                  Z, own, bx, 6, prev, p.max_update_norm, 0, st, true, psi_out, it > 3 ? 2 : 1,
                                                  sync /* direct transport (and the handles that time its launches): peer-written cells are read at system scope */);
        };
        auto wait_gate = [&]() -> int {  // row it-2 must be global before the first pass-B launch of this iteration
            if (prev && red_issued[it & 1]) SOBFU_HIP_TRY(hipStreamWaitEvent(st, t->ev_red[it & 1], 0));
            return 0;
        };
        auto after_b = [&]() -> int {  // row `it` is complete on `st`; it gates iteration it+2
            if (it > n_iters - 2) return 0;  // the tail rows are reduced once, by tiled_end
            // every row after the last one reduced so far, up to this one (normally just this one; more after a change of schedule
            // in mid-solve, whose reduction points differ)
            const int first = std::min(q.red_upto + 1, it);
            uint32_t* r_    = t->slots + (size_t) first * kSlots;
            const size_t cnt = (size_t) (it - first + 1) * kSlots;
            if (red_mode == RED_INLINE) {
                SOBFU_TRY(allreduce_max(t, r_, cnt, st));
                q.red_upto = it;
            }
            if (red_mode == RED_OWN_COMM) {
                SOBFU_HIP_TRY(hipEventRecord(t->ev_row, st));
                SOBFU_HIP_TRY(hipStreamWaitEvent(t->red_stream, t->ev_row, 0));
                SOBFU_TRY(allreduce_max(t, r_, cnt, t->red_stream, true));
                SOBFU_HIP_TRY(hipEventRecord(t->ev_red[it & 1], t->red_stream));
                red_issued[it & 1] = true;
                q.red_upto = it;
            }
            return 0;
        };
        const bool ev = (tiles || serial) && phases == 3 && t->prof_stride > 0 && it % t->prof_stride == 0 && (size_t) 4 * (t->prof_pending + 1) <= t->prof_ev.size();
        hipEvent_t* e = ev ? t->prof_ev.data() + 4 * t->prof_pending : nullptr;
        if (tiles) {
            // TILE PATH -- pass A's launch carries the exchange (push boxes first).  Direct transport: that is all of it -- the
            // launch retires when the neighbours' cells have landed too.  RCCL / callback: the packed buffer travels, one
            // kernel scatters what arrived.  No cross-stream events anywhere.
            if (phases & 1) {
                if (ev) SOBFU_HIP_TRY(hipEventRecord(e[0], st));
                const std::vector<sobfu_hip::TileLaunchBox>& bx = t->a_boxes[it & 1];
                const bool pushes = (multi || sync) && !(dbg & 1);  // a world of one has no messages
                const uint32_t* row_prev = (sync && it >= 2) ? t->slots + (size_t) (it - 1) * kSlots : nullptr;
                if (dbg & 4) {
                } else if (pushes && t->a_plan[it & 1]) {  // the planned launch: box list in device memory
                    SOBFU_TRY(sobfu_hip::launch_tile_pass_a_plan(t->a_plan[it & 1], f_in, t->c_g, psi_in, nu, p.w_reg, sync ? t->sync_d : nullptr,
                                                                 q.seq_base + (uint32_t) it, t->wait_enabled, row_prev, (uint32_t) (it - 1), st));
                } else {
                    SOBFU_TRY(sobfu_hip::launch_tile_pass_a(f_in, t->c_g, psi_in, nu, p.w_reg, Lx, Ly, Lz, pushes ? bx.data() : &t->a_own,
                                                            pushes ? (int) bx.size() : 1, sync ? t->sync_d : nullptr, q.seq_base + (uint32_t) it,
                                                            t->wait_enabled, row_prev, (uint32_t) (it - 1), 0, st, true));
                }
                if (ev) SOBFU_HIP_TRY(hipEventRecord(e[1], st));
                if (multi && !sync) SOBFU_TRY(exchange_packed(t, nu, st));
            }
            if (phases & 2) {
                SOBFU_TRY(wait_gate());
                if (ev) SOBFU_HIP_TRY(hipEventRecord(e[2], st));
                if (!(dbg & 8)) SOBFU_TRY(B(b_first, b_last, 0, 0, !(dbg & 2)));
                if (ev) {
                    SOBFU_HIP_TRY(hipEventRecord(e[3], st));
                    t->prof_pending += 1;
                }
                SOBFU_TRY(after_b());
                q.launched = it;  // advanced once the iteration is fully enqueued
                if (t->comm && !t->first_checked) SOBFU_TRY(first_iteration_watchdog(t, st));
            }
            continue;
        }
        if (phases != 3) return SOBFU_E_UNSUPPORTED;
        if (serial) {
            // no overlap, no cross-stream events: pass A, the exchange and pass B in line on `st`.  Every event record / wait
            // between two kernels costs a few microseconds of drained pipeline (~20 us per iteration for the overlapped
            // schedule's three), which a fast exchange on a thin slab does not repay.
            if (ev) SOBFU_HIP_TRY(hipEventRecord(e[0], st));
            SOBFU_TRY(A(lo, hi));
            if (ev) SOBFU_HIP_TRY(hipEventRecord(e[1], st));
            if (multi) SOBFU_TRY(exchange_planes(t, nu, H, st));
            SOBFU_TRY(wait_gate());
            if (ev) SOBFU_HIP_TRY(hipEventRecord(e[2], st));
            SOBFU_TRY(B(b_first, b_last));
            if (ev) {
                SOBFU_HIP_TRY(hipEventRecord(e[3], st));
                t->prof_pending += 1;
            }
            SOBFU_TRY(after_b());
            q.launched = it;
            if (t->comm && !t->first_checked) SOBFU_TRY(first_iteration_watchdog(t, st));
            continue;
        }
        // both boundary regions of a pass go out as ONE launch (two plane ranges)
        if (split_a) SOBFU_TRY(A(lo, a_lo, a_hi, hi));
        else SOBFU_TRY(A(lo, hi));
        if (multi) {
            SOBFU_HIP_TRY(hipEventRecord(t->ev_bnd, st));
            SOBFU_HIP_TRY(hipStreamWaitEvent(t->comm_stream, t->ev_bnd, 0));
            SOBFU_TRY(exchange_planes(t, nu, H, t->comm_stream));
            SOBFU_HIP_TRY(hipEventRecord(t->ev_xchg, t->comm_stream));
            if (red_mode == RED_COMM_STREAM && it >= 2 && it < n_iters && q.red_upto < it - 1) {
                // row it-1 is complete (its pass B precedes this iteration's ev_bnd, which the comm stream has waited for) and
                // gates iteration it+1: reduce it (and any earlier row not reduced yet) behind this iteration's exchange
                const int first = q.red_upto + 1;
                SOBFU_TRY(allreduce_max(t, t->slots + (size_t) first * kSlots, (size_t) (it - first) * kSlots, t->comm_stream));
                SOBFU_HIP_TRY(hipEventRecord(t->ev_red[(it - 1) & 1], t->comm_stream));
                red_issued[(it - 1) & 1] = true;
                q.red_upto = it - 1;
            }
        }
        if (split_a && a_hi > a_lo) SOBFU_TRY(A(a_lo, a_hi));
        SOBFU_TRY(wait_gate());
        if (b_hi > b_lo) SOBFU_TRY(B(b_lo, b_hi));
        if (multi) SOBFU_HIP_TRY(hipStreamWaitEvent(st, t->ev_xchg, 0));
        if (b_lo > b_first || b_last > b_hi) SOBFU_TRY(B(b_first, b_lo, b_hi, b_last));
        SOBFU_TRY(after_b());
        q.launched = it;
        if (t->comm && !t->first_checked) SOBFU_TRY(first_iteration_watchdog(t, st));
        // the next iteration's A_bnd overwrites nabla_U planes the exchange of THIS iteration sent: it runs on `st` after
        // the wait above, so the sends have completed by then; the next exchange's receives overwrite halo planes B_bnd
        // of THIS iteration read: the comm stream starts it only after the next ev_bnd, recorded on `st` behind B_bnd
    }
    if (n_steps > 0)
        t->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count() / n_steps;
    return 0;
}

// an error inside the loop: peers may be blocked in a collective this rank never issued -- drain what was enqueued, abort the
// communicators (so that the peers' calls return instead of hanging) and give the handle up
static int tiled_step(sobfu_hip_tiled* t, int n_steps, hipStream_t st, int phases = 3) {
    const int rc = tiled_step_impl(t, n_steps, st, phases);
    if (rc != 0 && rc != SOBFU_E_BADARG && rc != SOBFU_E_UNSUPPORTED) {
        abort_comms(t);  // FIRST: the stream may hold an RCCL kernel waiting for a peer this rank will never answer; the abort releases it
        (void) hipStreamSynchronize(st);
        t->q.active = false;
    }
    return rc;
}

// direct transport, end of a solve: the last max-norm row travels now; the handshake also tells every rank that its neighbours
// have retired their last pass B, i.e. that the next solve's first pushes cannot land under a kernel still reading the halo cells
static int tiled_flush(sobfu_hip_tiled* t, hipStream_t st) {
    sobfu_hip_tiled::Session& q = t->q;
    if (q.flushed) return 0;
    const int n = q.launched;
    const uint32_t seq = q.seq_base + (uint32_t) n + 1u;
    SOBFU_TRY(sobfu_hip::launch_tile_flush(t->sync_d, seq, t->wait_enabled, n > 0 ? t->slots + (size_t) n * kSlots : nullptr, (uint32_t) n, st));
    t->seq_total = seq;
    q.flushed    = true;
    return 0;
}

static int tiled_end(sobfu_hip_tiled* t, sobfu_hip_solver_report* report, float* per_iter_max_norm, hipStream_t st) {
    sobfu_hip_tiled::Session& q = t->q;
    if (!q.active) return SOBFU_E_BADARG;
    if (t->dead) return SOBFU_E_TIMEOUT;
    const sobfu_hip_solver_params& p = t->p;
    float* P[2] = {t->c_psi, t->c_psi2};
    sobfu_hip_solver_report r{};
    r.last_max_update_norm = r.last_max_update_index = r.last_e_data = r.last_e_reg = NAN;
    const bool can_converge = p.max_update_norm >= 0.f, multi = t->world > 1 || t->force_comm;
    const bool sync = uses_sync(t);
    const int n_iters = q.launched;
    const uint32_t* rows = t->slots;
    if (sync) {
        SOBFU_TRY(tiled_flush(t, st));
        rows = t->grows;
    } else if (multi && n_iters > 0) {  // rows the loop has not reduced yet: all of them without a threshold, the tail otherwise
        for (int k = 0; k < 2; ++k)
            if (q.red_issued[k]) SOBFU_HIP_TRY(hipStreamWaitEvent(st, t->ev_red[k], 0));
        const int first = q.red_upto + 1;
        if (first <= n_iters) SOBFU_TRY(allreduce_max(t, t->slots + (size_t) first * kSlots, (size_t) (n_iters - first + 1) * kSlots, st));
    }
    std::vector<uint32_t> hs((size_t) std::max(n_iters, 1) * kSlots, 0u);
    if (n_iters > 0) SOBFU_HIP_TRY(hipMemcpyAsync(hs.data(), rows + kSlots, (size_t) n_iters * kSlots * 4, hipMemcpyDeviceToHost, st));
    SOBFU_HIP_TRY(hipStreamSynchronize(st));
    if (sync) {
        uint32_t err = 0;
        SOBFU_HIP_TRY(hipMemcpy(&err, &t->sync_d->err, 4, hipMemcpyDeviceToHost));
        if (err) {
            std::fprintf(stderr, "sobfu_hip: rank %d: the arrival flag of rank %u did not reach this GPU within the deadline (direct transport)\n",
                         t->rank, err - 1u);
            t->dead  = true;
            q.active = false;
            return SOBFU_E_TIMEOUT;
        }
    }
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
    SOBFU_TRY(sobfu_hip::launch_unpack_vec(P[done & 1], q.psi, t->NL, st));
    SOBFU_TRY(sobfu_hip_tile3_apply(q.pn, t->X, t->Y, t->Z, q.pnp, q.psi, t->L[0], t->L[1], t->L[2], st));
    SOBFU_HIP_TRY(hipStreamSynchronize(st));
    q.active = false;
    if (report) *report = r;
    return 0;
}

int sobfu_hip_tiled_begin(sobfu_hip_tiled* t, const float* d_phi_global_local, const float* d_phi_n_full, float* d_phi_n_psi_local,
                          float* d_psi_local, int max_iters, void* stream) {
    SOBFU_CHECK_ARGS(t && d_phi_global_local && d_phi_n_full && d_phi_n_psi_local && d_psi_local && max_iters >= 0);
    return tiled_begin(t, d_phi_global_local, d_phi_n_full, d_phi_n_psi_local, d_psi_local, max_iters, (hipStream_t) stream);
}

int sobfu_hip_tiled_step(sobfu_hip_tiled* t, int n_iters, void* stream) {
    SOBFU_CHECK_ARGS(t && n_iters >= 0);
    return tiled_step(t, n_iters, (hipStream_t) stream);
}

int sobfu_hip_tiled_step_phase(sobfu_hip_tiled* t, int phase, void* stream) {
    SOBFU_CHECK_ARGS(t && phase >= 0 && phase <= 2);
    if (phase == 2) return (t->q.active && uses_sync(t)) ? tiled_flush(t, (hipStream_t) stream) : SOBFU_E_BADARG;
    return tiled_step(t, 1, (hipStream_t) stream, phase == 0 ? 1 : 2);
}

int sobfu_hip_tiled_end(sobfu_hip_tiled* t, sobfu_hip_solver_report* report, float* per_iter_max_norm, void* stream) {
    SOBFU_CHECK_ARGS(t);
    return tiled_end(t, report, per_iter_max_norm, (hipStream_t) stream);
}

int sobfu_hip_tiled_iterate(sobfu_hip_tiled* t, const float* d_phi_global_local, const float* d_phi_n_full,
                            float* d_phi_n_psi_local, float* d_psi_local, int n_iters, sobfu_hip_solver_report* report,
                            float* per_iter_max_norm, void* stream) {
    SOBFU_CHECK_ARGS(t && d_phi_global_local && d_phi_n_full && d_phi_n_psi_local && d_psi_local && n_iters >= 0);
    hipStream_t st = (hipStream_t) stream;
    SOBFU_TRY(tiled_begin(t, d_phi_global_local, d_phi_n_full, d_phi_n_psi_local, d_psi_local, n_iters, st));
    SOBFU_TRY(tiled_step(t, n_iters, st));
    return tiled_end(t, report, per_iter_max_norm, st);
}

}  // extern "C"
