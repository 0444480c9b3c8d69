// Solver kernels for gfx950.
//
// (The launcher-for-launcher counterparts of include/sobfu/solver.hpp:109-136 -- potential gradient, the three 1-D Sobolev
//  convolutions, psi update -- live in launcher_kernels.hip.)
// The MI355X-native two-pass decomposition of one solver iteration (solver.cu:114-193):
//             pass A  fused_potential_gradient : grad(phi_n o psi) + (-Lap psi) + combine      -> nabla_U
//             pass B  fused_smooth_update_apply: (Sx+Sy+Sz) nabla_U, psi -= alpha*.., phi_n o psi, max||u||^2
//           Both march along z with a register pipeline for the z taps and stage each xy plane (plus a
//           radius-1 / radius-3 halo) in a double-buffered LDS tile for the x/y taps: every plane is read from
//           HBM once per tile (+ halo), 112 B/voxel/iteration algorithmic traffic instead of the reference's
//           456 B/voxel.
//
// Arithmetic is op-for-op the reference's (see sobfu_device.hpp): results are bit-identical to the launcher-for-launcher kernels.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"
#include "sobfu_launch.hpp"

using namespace sobfu_hip;

#ifndef SOBFU_SWIZZLE_B
#define SOBFU_SWIZZLE_B true  // XCD-aware tile map for pass B (see tile_of_block)
#endif

namespace {

struct Taps {
    float s[7];
};
constexpr int kMaxMsgs = 18;  // 6 face + 12 edge neighbours of a 3-D tile

// the fused two-pass iteration: device code in four included parts, one translation unit
#include "solver_iter_common.inl"
#include "solver_pass_a.inl"
#include "solver_pass_b.inl"
#include "solver_aux_kernels.inl"

}  // namespace

// Tile configuration of the fused passes (see DESIGN.md "Kernel tuning").
// tile of a workgroup: 64 lanes x 8 waves, one row per wave (rows-per-thread 2 / 4 and 4 / 16 waves were measured in rounds 1 - 2:
// profiles/LABBOOK.md; the kernels keep RPT / WY as template parameters, the launchers instantiate this one shape)
constexpr int kRPT = 1, kWY = 8;

namespace sobfu_hip {

// z-chunk length of a fused pass.  A launch has tiles * ceil(nz / zc) workgroups; `capacity` of them are resident at
// once on the chip (256 CUs x workgroups per CU allowed by VGPRs / LDS / waves).  Cost model: time ~ (1 + refill / zc)
// / utilisation, where utilisation = groups / (ceil(groups / capacity) * capacity) penalises a ragged last wave of
// workgroups (measured at 256^3, pass B: 768 groups = exactly 3 per CU: 166 us; 512 groups: 184 us; 1024: 195 us) and
// refill = planes re-read when a march starts (2 for pass A, 6 for pass B).  Small grids end up with many short
// marches, which is what the latency-bound regime wants (64^3: zc = 2 is 1.4x faster than zc = 8).
static int env_zc(const char* env) {  // tuning override (SOBFU_ZC_A / SOBFU_ZC_B): planes per march, 0 = none
    const char* e = getenv(env);
    const int v   = e ? atoi(e) : 0;
    return v > 0 ? v : 0;
}
int pick_zc(int X, int Y, int nz, int ty, int capacity, int refill, const char* env) {
    if (const int v = env_zc(env)) return v < nz ? v : nz;
    const long tiles = (long) ((X + TX - 1) / TX) * ((Y + ty - 1) / ty);
    int best_zc = nz;
    double best = 1e30;
    for (int c = 1; c <= nz; ++c) {
        const int zc = (nz + c - 1) / c;
        if (zc < 2 && nz >= 2) break;
        const long groups = tiles * ((nz + zc - 1) / zc);
        const long waves  = (groups + capacity - 1) / capacity;
        const double util = (double) groups / (double) (waves * capacity);
        const double cost = (1.0 + (double) refill / zc) / util;
        if (cost < best - 1e-9) { best = cost; best_zc = zc; }
    }
    return best_zc;
}

// Does the iteration's state (76 B per cell of the local arrays) stay in the 256 MiB Infinity Cache from one launch to the next?
// Then the streaming hints are off (they would push what the next launch reads out of the cache).  SOBFU_CACHE_CELLS overrides.
static bool cache_resident(int X, int Y, int Z) {
    const char* e = getenv("SOBFU_CACHE_CELLS");
    return (long) X * Y * Z <= (e ? atol(e) : 3300000L);  // ~250 MB / 76 B
}

// direct boxes: lanes of a wave that run along x, and the workgroups (of WY waves) the box needs
static int direct_wx(int ex) {
    int wx = 1;
    while (wx < ex && wx < 64) wx *= 2;
    return wx;
}
// A wave of a box that is thin in x touches up to 64 / wx cache lines with EVERY load (its lanes sit in different rows), and the
// address unit of a CU takes them one line per cycle: eight such waves in one workgroup -- on one CU -- queue up behind each
// other (the one-column x shell of a 2 x 2 x 2 tile: 13.9 us as 32 full workgroups).  Such boxes get few working waves per
// workgroup, i.e. many small workgroups that the dispatcher spreads over all CUs.
// Only where the launch leaves the chip room (pass B of a tile: fewer workgroups than slots) -- in pass A, whose march fills every
// slot, a thousand one-wave workgroups in front of it cost more than they save (`spread` = false: full workgroups).
static int direct_wpg(int wx, bool spread) { return spread ? std::max(1, std::min(kWY, wx / 4)) : kWY; }
static int direct_groups(const LaunchBox& s, int wx, bool spread) {
    const int wyl = 64 / wx, wpg = direct_wpg(wx, spread);
    const long waves = (long) ((s.x1 - s.x0 + wx - 1) / wx) * ((s.y1 - s.y0 + wyl - 1) / wyl) * (s.z1 - s.z0);
    return (int) ((waves + wpg - 1) / wpg);
}
static double box_cells(const LaunchBox& s) {
    return (s.x1 > s.x0 && s.y1 > s.y0 && s.z1 > s.z0) ? (double) (s.x1 - s.x0) * (s.y1 - s.y0) * (s.z1 - s.z0) : 0.0;
}
// geometry of one live box; returns its workgroups.  Marching boxes: zc_override > 0 fixes the planes per march; else, when the box's
// xy tiles fit the share of the chip it gets (`even`: launches that are one resident round -- multi-GPU tiles, cache-resident grids),
// the planes are split EVENLY over as many chunks as fill that share (chunk lengths differ by at most one plane: a launch of one
// round lasts as long as its longest march); else the cost model picks a chunk length (pick_zc).
static int finish_box(Box& b, const LaunchBox& s, int ty, int share, int refill, int zc_override, const char* env, bool spread, bool even = false) {
    b.x0 = s.x0; b.x1 = s.x1; b.y0 = s.y0; b.y1 = s.y1; b.z0 = s.z0; b.z1 = s.z1;
    b.kind = s.direct ? 1 : 0;
    b.wpg = kWY;
    b.rem = 0;
    b.pair = 0;
    if (s.direct) {
        b.zc  = direct_wx(s.x1 - s.x0);
        b.wpg = direct_wpg(b.zc, spread);
        return direct_groups(s, b.zc, spread);
    }
    const int eu = s.x1 - s.x0, ev = s.y1 - s.y0, nz = s.z1 - s.z0;
    const int tiles = ((eu + TX - 1) / TX) * ((ev + ty - 1) / ty);
    int nch = 0;
    if (zc_override <= 0 && env_zc(env) == 0 && even && tiles <= share) nch = std::max(share / tiles, 1);
    if (nch > 0 && zc_override <= 0) {
        nch   = std::min(nch, std::max(nz / 2, 1));  // a march of one plane is all prologue
        b.zc  = nz / nch;
        b.rem = nz % nch;
        return tiles * nch;
    }
    b.zc = zc_override > 0 ? std::min(zc_override, nz) : pick_zc(eu, ev, nz, ty, share, refill, env);
    return tiles * ((nz + b.zc - 1) / b.zc);
}
// Fills the launch geometry of a box list: z-chunk per marching box (cost model above, the chip's capacity shared between the
// marching boxes; direct boxes are one short round trip and take no share) and the workgroup prefix -- marching boxes first.
// Returns the workgroups.
static int finish_boxes(BoxList& L, const LaunchBox* boxes, int n, int ty, int capacity, int refill, int zc_override, const char* env,
                        bool even = false) {
    L.n = 0;
    int live = 0;
    bool thin = false;
    for (int i = 0; i < n; ++i) {
        live += (box_cells(boxes[i]) > 0 && !boxes[i].direct) ? 1 : 0;
        thin = thin || (box_cells(boxes[i]) > 0 && boxes[i].direct);
    }
    // an even split fills the marching share exactly -- then the thin boxes' workgroups would start only when a march ends, and end the
    // launch: they keep a sixteenth of the slots (2 x 2 x 2 tile of 256^3: 15 chunks -> 480 + 288 workgroups, 43.0 us; 16 -> 512 + 288, 44.1)
    if (even && thin) capacity -= capacity / 16;
    int total = 0;
    L.m0 = L.m1 = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const bool direct_pass = pass == 1;  // marching boxes first
        if (!direct_pass) L.m0 = total;
        for (int i = 0; i < n && L.n < kMaxBoxes; ++i) {
            if (box_cells(boxes[i]) == 0 || boxes[i].direct != direct_pass) continue;
            // the chip's workgroup slots are shared equally between the marching boxes (the two plane ranges of an overlapped slab
            // schedule): a thin range is latency-critical, so it gets as many short marches as the big one gets long ones
            L.first[L.n] = total;
            total += finish_box(L.b[L.n], boxes[i], ty, std::max(capacity / std::max(live, 1), 1), refill, zc_override, env, true, even);
            ++L.n;
        }
        if (!direct_pass) L.m1 = total;
    }
    for (int k = L.n; k <= kMaxBoxes; ++k) L.first[k] = total;
    return total;
}

int launch_pass_a_boxes(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z, const LaunchBox* boxes,
                        int n, const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact) {
    constexpr int TY = kRPT * kWY;
    bool direct = false;
    for (int i = 0; i < n; ++i) direct = direct || (boxes[i].direct && box_cells(boxes[i]) > 0);
    if (direct) {  // thin boxes: the tile kernel (no messages, no signalling)
        std::vector<TileLaunchBox> tb((size_t) n);
        for (int i = 0; i < n; ++i) tb[(size_t) i] = TileLaunchBox{boxes[i], nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        return launch_tile_pass_a(pnp, pg, psi, nU, w_reg, X, Y, Z, tb.data(), n, (TileSync*) nullptr, 0, 0, nullptr, 0, zc, stream, compact);
    }
    PassAArgs a{{pnp, pg, psi, nU, {X, Y, Z}, w_reg, prev_slots, max_update_norm}, {}};
    const int groups = finish_boxes(a.boxes, boxes, n, TY, 256 * 4 * 8 / kWY, 2, zc, "SOBFU_ZC_A");  // <= 52 VGPR, 22 KB LDS: 4 workgroups of 8 waves per CU
    if (groups == 0) return 0;
    const dim3 grid((unsigned) groups), block(TX, kWY);
    if (compact && cache_resident(X, Y, Z)) hipLaunchKernelGGL((fused_potential_gradient_kernel<kRPT, kWY, true, 0>), grid, block, 0, stream, a);
    else if (compact) hipLaunchKernelGGL((fused_potential_gradient_kernel<kRPT, kWY, true, kNT>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((fused_potential_gradient_kernel<kRPT, kWY, false, 0>), grid, block, 0, stream, a);
    return (int) hipGetLastError();
}

// Pass A of a multi-GPU tile (see tile_potential_gradient_kernel): the boxes with a destination (push boxes: direct, their
// result goes to `dst` only) are numbered first, then the others.  sync / seq / wait / row: the direct transport's signalling.
// the launch geometry of a tile's pass A: push boxes first; returns the workgroups (< 0: too many boxes)
static int fill_tile_boxes(TileBoxList& L, const TileLaunchBox* boxes, int n, int X, int Y, int Z, int zc) {
    constexpr int TY = kRPT * kWY;
    L.n = 0;
    int live = 0, total = 0;
    for (int i = 0; i < n; ++i) live += (box_cells(boxes[i].box) > 0 && !boxes[i].box.direct && boxes[i].dst == nullptr) ? 1 : 0;
    for (int pass = 0; pass < 2; ++pass) {  // push boxes first
        for (int i = 0; i < n; ++i) {
            const TileLaunchBox& s = boxes[i];
            if ((s.dst != nullptr) != (pass == 0) || box_cells(s.box) == 0) continue;
            if (L.n >= kMaxTileBoxes) return -1;
            TileBox& t = L.b[L.n];
            L.first[L.n] = total;
            // z-chunks: a marching push box (a face with wide rows) marches up to 8 planes; the owned block of a cache-resident
            // tile is sized for TWO workgroups per CU -- the push boxes take slots too, and at that size 8-plane marches beat the 4-plane ones that
            // filling all four slots per CU would give (2 x 2 x 2 tile of 256^3: pass A 19.8 -> 19.1 us, 1 x 2 x 4: 18.9 -> 17.2)
            const int zc_box = zc > 0 ? zc : ((s.dst != nullptr && !s.box.direct) ? std::min(8, s.box.z1 - s.box.z0) : 0);
            total += finish_box(t.b, s.box, TY, std::max(256 * (cache_resident(X, Y, Z) ? 2 : 4) * 8 / kWY / std::max(live, 1), 1), 2, zc_box,
                                "SOBFU_ZC_A", false);
            t.push.base = s.dst;  // (member by member: the list is looked up by its bytes, padding included -- the caller zeroed it)
            t.push.ox = s.ox; t.push.oy = s.oy; t.push.oz = s.oz; t.push.px = s.px; t.push.py = s.py;
            t.push.y0 = s.push_y0; t.push.y1 = s.push_y1; t.push.lz0 = s.local_z0; t.push.lz1 = s.local_z1;
            ++L.n;
        }
        if (pass == 0) L.n_push_wgs = total;
    }
    for (int k = L.n; k <= kMaxTileBoxes; ++k) L.first[k] = total;
    return total;
}

// Device copies of the box lists seen so far.  An entry belongs to the HIP DEVICE it was allocated on (two handles on different GPUs
// of one process may build byte-identical lists: each gets its own copy), is found by a hash of the list's bytes (then memcmp), and is
// uploaded with hipMemcpyAsync ON THE LAUNCH STREAM from a pinned staging copy that lives as long as the entry: no blocking
// null-stream copy inside a launch path, stream order makes the list visible to the launch that follows.  Entries are never freed
// one by one (a launch in flight may still be reading its list); when a device's entries exceed kMaxCachedLists -- lists are keyed by
// peer pointers, so a process that keeps creating handles keeps creating lists -- the device is drained and its entries are dropped.
// Nobody keeps a pointer into this cache beyond the launch it was looked up for (a TilePassAPlan owns its own copy).
struct CachedBoxes {
    int device;
    uint64_t hash;
    TileBoxList* host;  // pinned
    TileBoxList* dev;
    hipEvent_t uploaded;    // completion of the upload: a launch that finds the entry waits for it on its own stream until it is known done
    bool ready;
    uint64_t retired;       // generation in which the entry left the look-up (graveyard entries)
};
constexpr size_t kMaxCachedLists = 256;
static std::vector<CachedBoxes> g_box_cache, g_box_graveyard;
static std::mutex g_box_cache_mutex;
static uint64_t g_box_generation = 0;  // retirements so far
static uint64_t bytes_hash(const void* p, size_t n) {  // FNV-1a
    const unsigned char* b = (const unsigned char*) p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
static int device_boxes(const TileBoxList& L, TileBoxList** out, hipStream_t stream) {
    int dev = 0;
    SOBFU_HIP_TRY(hipGetDevice(&dev));
    const uint64_t h = bytes_hash(&L, sizeof L);
    std::unique_lock<std::mutex> lock(g_box_cache_mutex);
    size_t on_dev = 0;
    for (CachedBoxes& c : g_box_cache) {
        if (c.device != dev) continue;
        ++on_dev;
        if (c.hash == h && std::memcmp(c.host, &L, sizeof L) == 0) {
            if (!c.ready) {
                if (hipEventQuery(c.uploaded) == hipSuccess) c.ready = true;
                else SOBFU_HIP_TRY(hipStreamWaitEvent(stream, c.uploaded, 0));  // (also on the uploading stream: a no-op there, and a recycled stream handle cannot fool it)
            }
            *out = c.dev;
            return 0;
        }
    }
    if (on_dev >= kMaxCachedLists) {  // rare: this device's entries retire.  Two generations: what retired in an EARLIER generation is
        // freed now, after a device drain; what retires now is only taken out of the look-up -- a thread that has just looked an entry up
        // and is about to launch with it (the lock is not held across the launch) still finds it alive.  The drain runs WITHOUT the lock:
        // with in-process ranks on the direct transport a kernel in flight may be waiting for a peer whose host thread needs this cache.
        const uint64_t gen = g_box_generation;
        lock.unlock();
        SOBFU_HIP_TRY(hipDeviceSynchronize());
        lock.lock();
        for (size_t k = 0; k < g_box_graveyard.size();) {
            if (g_box_graveyard[k].device == dev && g_box_graveyard[k].retired <= gen) {  // retired before the drain began
                (void) hipFree(g_box_graveyard[k].dev);
                (void) hipHostFree(g_box_graveyard[k].host);
                (void) hipEventDestroy(g_box_graveyard[k].uploaded);
                g_box_graveyard.erase(g_box_graveyard.begin() + (long) k);
            } else ++k;
        }
        if (g_box_generation == gen) {  // nobody else retired this device's entries while the lock was open
            g_box_generation += 1;
            for (size_t k = 0; k < g_box_cache.size();) {
                if (g_box_cache[k].device == dev) {
                    g_box_cache[k].retired = g_box_generation;
                    g_box_graveyard.push_back(g_box_cache[k]);
                    g_box_cache.erase(g_box_cache.begin() + (long) k);
                } else ++k;
            }
        }
    }
    CachedBoxes c{dev, h, nullptr, nullptr, nullptr, false, 0};
    hipError_t e = hipHostMalloc((void**) &c.host, sizeof L, hipHostMallocDefault);
    if (e == hipSuccess) {
        std::memcpy(c.host, &L, sizeof L);
        e = hipMalloc((void**) &c.dev, sizeof L);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.uploaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMemcpyAsync(c.dev, c.host, sizeof L, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(c.uploaded, stream);
    if (e != hipSuccess) {
        if (c.uploaded) (void) hipEventDestroy(c.uploaded);
        if (c.dev) (void) hipFree(c.dev);
        if (c.host) (void) hipHostFree(c.host);
        return (int) e;
    }
    g_box_cache.push_back(c);
    *out = c.dev;
    return 0;
}
static int launch_tile_boxes(const TileBoxList* d_boxes, int groups, const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X,
                             int Y, int Z, TileSync* sync, uint32_t seq, int wait, const uint32_t* row, uint32_t row_index, hipStream_t stream,
                             bool compact) {
    if (groups == 0) return 0;
    TilePassAArgsP a{{pnp, pg, psi, nU, {X, Y, Z}, w_reg, nullptr, 0.f}, d_boxes, {sync, seq, wait, row, row_index}};
    const dim3 grid((unsigned) groups), block(TX, kWY);
    if (compact && cache_resident(X, Y, Z)) hipLaunchKernelGGL((tile_potential_gradient_kernel<kRPT, kWY, true, 0>), grid, block, 0, stream, a);
    else if (compact) hipLaunchKernelGGL((tile_potential_gradient_kernel<kRPT, kWY, true, kNT>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((tile_potential_gradient_kernel<kRPT, kWY, false, 0>), grid, block, 0, stream, a);
    return (int) hipGetLastError();
}

int launch_tile_pass_a(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z, const TileLaunchBox* boxes,
                       int n, TileSync* sync, uint32_t seq, int wait, const uint32_t* row, uint32_t row_index, int zc, hipStream_t stream, bool compact) {
    TileBoxList L;
    std::memset(&L, 0, sizeof L);  // (padding bytes too: the list is looked up by its bytes)
    const int total = fill_tile_boxes(L, boxes, n, X, Y, Z, zc);
    if (total < 0) return SOBFU_E_BADARG;
    if (total == 0) return 0;
    TileBoxList* d = nullptr;
    SOBFU_TRY(device_boxes(L, &d, stream));
    return launch_tile_boxes(d, total, pnp, pg, psi, nU, w_reg, X, Y, Z, sync, seq, wait, row, row_index, stream, compact);
}

// A PLANNED launch of the same pass (compact format): the geometry is worked out once per handle and half of the nabla_U ping-pong
struct TilePassAPlan {
    TileBoxList* d_boxes = nullptr;
    int groups = 0, X = 0, Y = 0, Z = 0;
};
int tile_pass_a_plan_create(TilePassAPlan** out, const TileLaunchBox* boxes, int n, int X, int Y, int Z) {
    TileBoxList L;
    std::memset(&L, 0, sizeof L);
    const int total = fill_tile_boxes(L, boxes, n, X, Y, Z, 0);
    if (total < 0) return SOBFU_E_BADARG;
    auto* p = new TilePassAPlan();
    p->groups = total; p->X = X; p->Y = Y; p->Z = Z;
    // a plan OWNS its device copy (the cache above may drop its entries; a plan lives as long as its handle): plan time is handle
    // creation, not a launch path, so a blocking copy is fine -- the list is simply there before any stream launches with it
    hipError_t e = hipSuccess;
    if (total > 0) {
        e = hipMalloc((void**) &p->d_boxes, sizeof L);
        if (e == hipSuccess) e = hipMemcpy(p->d_boxes, &L, sizeof L, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (p->d_boxes) (void) hipFree(p->d_boxes);
        delete p;
        return (int) e;
    }
    *out = p;
    return 0;
}
void tile_pass_a_plan_destroy(TilePassAPlan* p) {
    if (p == nullptr) return;
    if (p->d_boxes) (void) hipFree(p->d_boxes);  // (hipFree waits for the device: no launch is still reading the list)
    delete p;
}
int launch_tile_pass_a_plan(const TilePassAPlan* p, const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, TileSync* sync,
                            uint32_t seq, int wait, const uint32_t* row, uint32_t row_index, hipStream_t stream) {
    return launch_tile_boxes(p->d_boxes, p->groups, pnp, pg, psi, nU, w_reg, p->X, p->Y, p->Z, sync, seq, wait, row, row_index, stream, true);
}

int launch_tile_pingpong(TileSync* sync, int q, int first, uint32_t seq0, int reps, hipStream_t stream) {
    hipLaunchKernelGGL(tile_pingpong_kernel, dim3(1), dim3(64), 0, stream, sync, q, first, seq0, reps);
    return (int) hipGetLastError();
}

int launch_tile_flush(TileSync* sync, uint32_t seq, int wait, const uint32_t* row, uint32_t row_index, hipStream_t stream) {
    hipLaunchKernelGGL(tile_flush_kernel, dim3(1), dim3(64), 0, stream, sync, seq, wait, row, row_index);
    return (int) hipGetLastError();
}

int launch_pass_b_boxes(const float* nU, float* psi, const float* phi_n, float* pnp, float* updates, uint32_t* slots, const float taps[7],
                        float alpha, int X, int Y, int Z, int pX, int pY, int pZ, const int own[6], const LaunchBox* boxes, int n,
                        const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact, float* psi_out, int prev_rows,
                        bool sys_acquire) {
    constexpr int TY = kRPT * kWY;
    PassBArgs a{nU, psi, phi_n, pnp, (float4*) updates, slots, {X, Y, Z}, {}, alpha, {}, prev_slots, max_update_norm, {pX, pY, pZ},
                {own[0], own[1], own[2], own[3], own[4], own[5]}, prev_rows, psi_out ? psi_out : psi, sys_acquire ? 1 : 0};
    for (int i = 0; i < 7; ++i) a.S.s[i] = taps[i];
    if ((size_t) X * Y * 16 >= ((size_t) 1 << 32)) return SOBFU_E_UNSUPPORTED;  // in-plane byte offsets are 32-bit
    if (sys_acquire && (size_t) X * Y * Z * 12 >= ((size_t) 1 << 32)) return SOBFU_E_UNSUPPORTED;  // scope-carrying loads are buffer loads: arrays below 4 GiB
    const bool idx32 = (size_t) pX * pY * pZ < ((size_t) 1 << 30);  // tsdf-only phi_n below 4 GiB: 32-bit byte offsets for the corner gather
    // the solver's own format (compact, 32-bit gather offsets, no `updates`): streaming hints only for grids beyond the Infinity
    // Cache; the pipelined march where the launch is latency-bound (cache-resident sizes; SOBFU_PIPE_B=0/1 overrides)
    const bool resident = cache_resident(X, Y, Z);
    const char* pipe_e = getenv("SOBFU_PIPE_B");
    // (buffer addressing: arrays below 4 GiB.  Connected handles ALWAYS take the pipelined march, whatever SOBFU_PIPE_B says: it is the
    // march whose loads carry the system scope for the halo cells other GPUs stored)
    const bool pipe = compact && idx32 && !updates && (size_t) X * Y * Z * 12 < ((size_t) 1 << 32) && (sys_acquire || (pipe_e ? atoi(pipe_e) != 0 : resident));
    if (sys_acquire && !pipe) return SOBFU_E_UNSUPPORTED;  // never fall back to a march that reads peer-written cells with ordinary loads
    // workgroups a CU holds: <= 80 VGPR (launch bounds) and 32 - 48 KB LDS: 3 of 8 waves; the pipelined march (<= 128 VGPR): 2
    // cache-resident launches are ONE resident round of workgroups, which lasts as long as its longest march: the planes are
    // split evenly over as many z-chunks as fill the marching workgroups' share of the chip
    const int groups = finish_boxes(a.boxes, boxes, n, TY, 256 * (pipe ? 2 : 3) * 8 / kWY, 6, zc, "SOBFU_ZC_B", resident && pipe);
    if (groups == 0) return 0;
    bool direct = false;
    int zc_max = 0;
    for (int i = 0; i < a.boxes.n; ++i) {
        direct = direct || a.boxes.b[i].kind != 0;
        if (a.boxes.b[i].kind == 0) zc_max = std::max(zc_max, a.boxes.b[i].zc + (a.boxes.b[i].rem > 0 ? 1 : 0));  // the first `rem` chunks march one plane more
        if (a.boxes.b[i].kind == 0 && pipe && resident && SOBFU_PAIR_B) a.boxes.b[i].pair = 1;  // neighbouring z-chunks march towards / away from each other
    }
    const dim3 grid((unsigned) groups), block(TX, kWY);
#define SOBFU_LAUNCH_B(UPD, CMP, DIR) \
    hipLaunchKernelGGL((fused_smooth_update_apply_kernel<kRPT, kWY, UPD, CMP, DIR>), grid, block, 0, stream, a)
#define SOBFU_LAUNCH_BX(DIR, HLV, NTV, PIP) \
    hipLaunchKernelGGL((fused_smooth_update_apply_kernel<kRPT, kWY, false, true, DIR, true, HLV, NTV, PIP>), grid, block, 0, stream, a)
    const bool ntbuf = (size_t) X * Y * Z * 12 < ((size_t) 1 << 32);  // the plain march's 12-byte psi load / store as buffer instructions (their cache-policy operand carries the streaming hint): arrays below 4 GiB
    if (direct) {
        if (updates && compact) SOBFU_LAUNCH_B(true, true, true);
        else if (updates) SOBFU_LAUNCH_B(true, false, true);
        else if (compact && idx32) {
            if (resident && pipe) SOBFU_LAUNCH_BX(true, 0, 0, true);
            else if (resident) SOBFU_LAUNCH_BX(true, 0, 0, false);
            else if (pipe) SOBFU_LAUNCH_BX(true, 0, kNT, true);
            else if (ntbuf) hipLaunchKernelGGL((fused_smooth_update_apply_kernel<kRPT, kWY, false, true, true, true, 0, kNT, false, true>), grid, block, 0, stream, a);
            else SOBFU_LAUNCH_BX(true, 0, kNT, false);
        }
        else if (compact) SOBFU_LAUNCH_B(false, true, true);
        else SOBFU_LAUNCH_B(false, false, true);
    } else {
        if (updates && compact) SOBFU_LAUNCH_B(true, true, false);
        else if (updates) SOBFU_LAUNCH_B(true, false, false);
        else if (compact && idx32) {
            // long marches (big grids): halo requests run SOBFU_HLEAD planes ahead; short ones (small grids, multi-GPU tiles) skip
            // the extra prologue round trip
            const bool lead = SOBFU_HLEAD > 0 && zc_max >= SOBFU_HLEAD_MIN_ZC && !resident && !pipe;
            if (lead && ntbuf) hipLaunchKernelGGL((fused_smooth_update_apply_kernel<kRPT, kWY, false, true, false, true, SOBFU_HLEAD, kNT, false, true>), grid, block, 0, stream, a);
            else if (lead) SOBFU_LAUNCH_BX(false, SOBFU_HLEAD, kNT, false);
            else if (resident && pipe) SOBFU_LAUNCH_BX(false, 0, 0, true);
            else if (resident) SOBFU_LAUNCH_BX(false, 0, 0, false);
            else if (pipe) SOBFU_LAUNCH_BX(false, 0, kNT, true);
            else if (ntbuf) hipLaunchKernelGGL((fused_smooth_update_apply_kernel<kRPT, kWY, false, true, false, true, 0, kNT, false, true>), grid, block, 0, stream, a);
            else SOBFU_LAUNCH_BX(false, 0, kNT, false);
        }
        else if (compact) SOBFU_LAUNCH_B(false, true, false);
        else SOBFU_LAUNCH_B(false, false, false);
    }
#undef SOBFU_LAUNCH_BX
#undef SOBFU_LAUNCH_B
    return (int) hipGetLastError();
}

// z-range forms (whole x-y planes of the array): the single-GPU solver and the z-slab loop
int launch_pass_a(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z,
                  const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact, int z_lo, int z_hi,
                  int z_lo2, int z_hi2) {
    if (z_hi <= 0 && z_hi2 <= z_lo2) { z_lo = 0; z_hi = Z; }  // no range given: the whole grid
    const LaunchBox b[2] = {{0, X, 0, Y, z_lo, z_hi, false}, {0, X, 0, Y, z_lo2, z_hi2, false}};
    return launch_pass_a_boxes(pnp, pg, psi, nU, w_reg, X, Y, Z, b, 2, prev_slots, max_update_norm, zc, stream, compact);
}

int launch_pass_b(const float* nU, float* psi, const float* phi_n, float* pnp, float* updates, uint32_t* slots,
                  const float taps[7], float alpha, int X, int Y, int Z, const uint32_t* prev_slots,
                  float max_update_norm, int zc, hipStream_t stream, int phi_Z, int own_lo, int own_hi, bool compact, int z_lo,
                  int z_hi, int z_lo2, int z_hi2, float* psi_out, int prev_rows) {
    if (phi_Z <= 0) { phi_Z = Z; own_lo = 0; own_hi = Z; }
    if (z_hi <= 0 && z_hi2 <= z_lo2) { z_lo = 0; z_hi = Z; }  // no range given: the whole grid
    const LaunchBox b[2] = {{0, X, 0, Y, z_lo, z_hi, false}, {0, X, 0, Y, z_lo2, z_hi2, false}};
    const int own[6] = {0, X, 0, Y, own_lo, own_hi};
    return launch_pass_b_boxes(nU, psi, phi_n, pnp, updates, slots, taps, alpha, X, Y, Z, X, Y, phi_Z, own, b, 2, prev_slots, max_update_norm, zc,
                               stream, compact, psi_out, prev_rows);
}

#define SOBFU_LIN(N) dim3((unsigned) (((N) + 255) / 256)), dim3(256), 0, stream
int launch_pack_vec(const float* src4, float* dst3, size_t N, hipStream_t stream) {
    hipLaunchKernelGGL(pack_vec_kernel, SOBFU_LIN(N), (const float4*) src4, (P3*) dst3, N);
    return (int) hipGetLastError();
}
int launch_unpack_vec(const float* src3, float* dst4, size_t N, hipStream_t stream) {
    hipLaunchKernelGGL(unpack_vec_kernel, SOBFU_LIN(N), (const P3*) src3, (float4*) dst4, N);
    return (int) hipGetLastError();
}
int launch_extract_tsdf(const float* src2, float* dst1, size_t N, hipStream_t stream) {
    hipLaunchKernelGGL(extract_tsdf_kernel, SOBFU_LIN(N), (const float2*) src2, dst1, N);
    return (int) hipGetLastError();
}
int launch_apply_tsdf_only(const float* phi1, float* out1, const float* psi3, int X, int Y, int Z, hipStream_t stream, int phi_Z, int phi_X,
                           int phi_Y) {
    hipLaunchKernelGGL(apply_tsdf_only_kernel, voxel_grid(X, Y, Z), voxel_block(), 0, stream, phi1, out1, (const P3*) psi3, Dims{X, Y, Z},
                       Dims{phi_X > 0 ? phi_X : X, phi_Y > 0 ? phi_Y : Y, phi_Z > 0 ? phi_Z : Z});
    return (int) hipGetLastError();
}
// n boxes of 6 ints (x0, x1, y0, y1, z0, z1) of a 12-byte (Lx, Ly, Lz) field <-> consecutive buffer segments (x fastest)
int launch_msg_copy(bool pack, float* field3, float* buf, int Lx, int Ly, int Lz, const int* boxes, int n, hipStream_t stream) {
    if (n <= 0) return 0;
    if (n > kMaxMsgs) return SOBFU_E_BADARG;
    MsgBoxes m{};
    m.n = n;
    unsigned total = 0;
    for (int i = 0; i < n; ++i) {
        const int* b = boxes + 6 * i;
        if (!(b[0] >= 0 && b[1] > b[0] && b[1] <= Lx && b[2] >= 0 && b[3] > b[2] && b[3] <= Ly && b[4] >= 0 && b[5] > b[4] && b[5] <= Lz)) return SOBFU_E_BADARG;
        m.x0[i] = b[0]; m.y0[i] = b[2]; m.z0[i] = b[4];
        m.nx[i] = b[1] - b[0]; m.ny[i] = b[3] - b[2];
        m.first[i] = total;
        total += (unsigned) (b[1] - b[0]) * (unsigned) (b[3] - b[2]) * (unsigned) (b[5] - b[4]);
    }
    for (int k = n; k <= kMaxMsgs; ++k) m.first[k] = total;
    const dim3 grid((total + 255u) / 256u), block(256);
    if (pack) hipLaunchKernelGGL(msg_copy_kernel<true>, grid, block, 0, stream, field3, buf, Dims{Lx, Ly, Lz}, m);
    else hipLaunchKernelGGL(msg_copy_kernel<false>, grid, block, 0, stream, field3, buf, Dims{Lx, Ly, Lz}, m);
    return (int) hipGetLastError();
}
int launch_msg_scatter_table(float* field3, const float* buf, const uint32_t* d_table, unsigned n_cells, hipStream_t stream) {
    if (n_cells == 0) return 0;
    hipLaunchKernelGGL(msg_scatter_table_kernel, dim3((n_cells + 255u) / 256u), dim3(256), 0, stream, field3, buf, d_table, n_cells);
    return (int) hipGetLastError();
}
#undef SOBFU_LIN
int launch_compact_enter(const float* psi4, const float* pg2, const float* pn2, float* c_psi, float* c_g, float* c_n, float* c_f, int X, int Y,
                         int Z, hipStream_t stream) {
    hipLaunchKernelGGL(compact_enter_kernel, voxel_grid(X, Y, Z), voxel_block(), 0, stream, (const float4*) psi4, (const float2*) pg2,
                       (const float2*) pn2, (P3*) c_psi, c_g, c_n, c_f, Dims{X, Y, Z});
    return (int) hipGetLastError();
}
int launch_compact_leave(const float* c_psi, const float* pn2, float* psi4, float* pnp2, int X, int Y, int Z, hipStream_t stream) {
    hipLaunchKernelGGL(compact_leave_kernel, voxel_grid(X, Y, Z), voxel_block(), 0, stream, (const P3*) c_psi, (const float2*) pn2,
                       (float4*) psi4, (float2*) pnp2, Dims{X, Y, Z});
    return (int) hipGetLastError();
}

}  // namespace sobfu_hip

extern "C" {

int sobfu_hip_fused_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_psi,
                                       float* d_nabla_U, float w_reg, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_phi_n_psi && d_phi_global && d_psi && d_nabla_U && X > 1 && Y > 1 && Z > 1);
    if ((size_t) X * Y * Z > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;
    return sobfu_hip::launch_pass_a(d_phi_n_psi, d_phi_global, d_psi, d_nabla_U, w_reg, X, Y, Z, nullptr, 0.f, 0, (hipStream_t) stream, false, 0, 0);
}

int sobfu_hip_fused_smooth_update_apply(const float* d_nabla_U, float* d_psi, const float* d_phi_n, float* d_phi_n_psi,
                                        float* d_updates, uint32_t* d_max_sq_slots, const float taps[7], float alpha,
                                        int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_nabla_U && d_psi && d_phi_n && d_phi_n_psi && d_max_sq_slots && taps && X > 0 && Y > 0 && Z > 0);
    if ((size_t) X * Y * Z > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;
    return sobfu_hip::launch_pass_b(d_nabla_U, d_psi, d_phi_n, d_phi_n_psi, d_updates, d_max_sq_slots, taps, alpha, X, Y, Z,
                                    nullptr, 0.f, 0, (hipStream_t) stream, 0, 0, 0, false, 0, 0);
}

int sobfu_hip_pack_vec3(const float* d_src4, float* d_dst3, size_t n, void* stream) {
    SOBFU_CHECK_ARGS(d_src4 && d_dst3 && n > 0);
    return sobfu_hip::launch_pack_vec(d_src4, d_dst3, n, (hipStream_t) stream);
}
int sobfu_hip_unpack_vec3(const float* d_src3, float* d_dst4, size_t n, void* stream) {
    SOBFU_CHECK_ARGS(d_src3 && d_dst4 && n > 0);
    return sobfu_hip::launch_unpack_vec(d_src3, d_dst4, n, (hipStream_t) stream);
}
int sobfu_hip_extract_tsdf(const float* d_src2, float* d_dst1, size_t n, void* stream) {
    SOBFU_CHECK_ARGS(d_src2 && d_dst1 && n > 0);
    return sobfu_hip::launch_extract_tsdf(d_src2, d_dst1, n, (hipStream_t) stream);
}
// ---- 3-D tiles (sobfu_hip_tile3_*): local arrays (Lx, Ly, Lz) with halo cells on every side that faces a neighbour ----
static bool box_ok(const int b[6], int Lx, int Ly, int Lz) {
    return b[0] >= 0 && b[0] <= b[1] && b[1] <= Lx && b[2] >= 0 && b[2] <= b[3] && b[3] <= Ly && b[4] >= 0 && b[4] <= b[5] && b[5] <= Lz;
}

int sobfu_hip_tile3_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_psi, float* d_nabla_U, float w_reg,
                                       int Lx, int Ly, int Lz, const int box[6], int thin, const uint32_t* d_prev_slots,
                                       float max_update_norm, int compact, void* stream) {
    SOBFU_CHECK_ARGS(d_phi_n_psi && d_phi_global && d_psi && d_nabla_U && Lx > 1 && Ly > 1 && Lz > 1 && box && box_ok(box, Lx, Ly, Lz));
    if ((size_t) Lx * Ly * Lz > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;
    const sobfu_hip::LaunchBox b{box[0], box[1], box[2], box[3], box[4], box[5], thin != 0};
    return sobfu_hip::launch_pass_a_boxes(d_phi_n_psi, d_phi_global, d_psi, d_nabla_U, w_reg, Lx, Ly, Lz, &b, 1, d_prev_slots, max_update_norm, 0,
                                          (hipStream_t) stream, compact != 0);
}

int sobfu_hip_tile3_smooth_update_apply(const float* d_nabla_U, float* d_psi, const float* d_phi_n, float* d_phi_n_psi, float* d_updates,
                                        uint32_t* d_max_sq_slots, const float taps[7], float alpha, int Lx, int Ly, int Lz, int Xg, int Yg,
                                        int Zg, const int own[6], const int box[6], int thin, const uint32_t* d_prev_slots,
                                        float max_update_norm, int compact, void* stream) {
    SOBFU_CHECK_ARGS(d_nabla_U && d_psi && d_phi_n && d_phi_n_psi && d_max_sq_slots && taps && Lx > 0 && Ly > 0 && Lz > 0 && Xg > 0 && Yg > 0 &&
                     Zg > 0 && own && box && box_ok(own, Lx, Ly, Lz) && box_ok(box, Lx, Ly, Lz));
    if ((size_t) Lx * Ly * Lz > (size_t) 0x7fffffff || (size_t) Xg * Yg * Zg > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;
    const sobfu_hip::LaunchBox b{box[0], box[1], box[2], box[3], box[4], box[5], thin != 0};
    return sobfu_hip::launch_pass_b_boxes(d_nabla_U, d_psi, d_phi_n, d_phi_n_psi, d_updates, d_max_sq_slots, taps, alpha, Lx, Ly, Lz, Xg, Yg, Zg,
                                          own, &b, 1, d_prev_slots, max_update_norm, 0, (hipStream_t) stream, compact != 0);
}

int sobfu_hip_tile3_apply_tsdf_only(const float* d_phi1, int Xg, int Yg, int Zg, float* d_out1, const float* d_psi3, int Lx, int Ly, int Lz,
                                    void* stream) {
    SOBFU_CHECK_ARGS(d_phi1 && d_out1 && d_psi3 && Lx > 0 && Ly > 0 && Lz > 0 && Xg > 0 && Yg > 0 && Zg > 0);
    return sobfu_hip::launch_apply_tsdf_only(d_phi1, d_out1, d_psi3, Lx, Ly, Lz, (hipStream_t) stream, Zg, Xg, Yg);
}

int sobfu_hip_tile3_pack(const float* d_field3, int Lx, int Ly, int Lz, float* d_buf, const int* boxes, int n_boxes, void* stream) {
    SOBFU_CHECK_ARGS(d_field3 && d_buf && boxes && n_boxes >= 0 && Lx > 0 && Ly > 0 && Lz > 0);
    return sobfu_hip::launch_msg_copy(true, const_cast<float*>(d_field3), d_buf, Lx, Ly, Lz, boxes, n_boxes, (hipStream_t) stream);
}

int sobfu_hip_tile3_unpack(float* d_field3, int Lx, int Ly, int Lz, const float* d_buf, const int* boxes, int n_boxes, void* stream) {
    SOBFU_CHECK_ARGS(d_field3 && d_buf && boxes && n_boxes >= 0 && Lx > 0 && Ly > 0 && Lz > 0);
    return sobfu_hip::launch_msg_copy(false, d_field3, const_cast<float*>(d_buf), Lx, Ly, Lz, boxes, n_boxes, (hipStream_t) stream);
}

}  // extern "C"
