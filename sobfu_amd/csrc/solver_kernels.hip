// Solver kernels for gfx950.
//
// Part 1 -- launcher-for-launcher counterparts of include/sobfu/solver.hpp:109-136 (potential gradient, the
//           three 1-D Sobolev convolutions, psi update), one lane per voxel, 3-D grids.
// Part 2 -- the MI355X-native two-pass decomposition of one solver iteration (solver.cu:114-193):
//             pass A  fused_potential_gradient : grad(phi_n o psi) + (-Lap psi) + combine      -> nabla_U
//             pass B  fused_smooth_update_apply: (Sx+Sy+Sz) nabla_U, psi -= alpha*.., phi_n o psi, max||u||^2
//           Both march along z with a register pipeline for the z taps and stage each xy plane (plus a
//           radius-1 / radius-3 halo) in a double-buffered LDS tile for the x/y taps: every plane is read from
//           HBM once per tile (+ halo), 112 B/voxel/iteration algorithmic traffic instead of the reference's
//           456 B/voxel.
//
// Arithmetic is op-for-op the reference's (see sobfu_device.hpp): results are bit-identical to Part 1.
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

// ============================================================================================================
// Part 1: reference-shaped launchers
// ============================================================================================================

// calculate_potential_gradient_kernel -- solver.cu:15-33
__global__ void __launch_bounds__(256) potential_gradient_kernel(const float2* __restrict__ pnp, const float2* __restrict__ pg,
                                                                 const float4* __restrict__ grad, const float4* __restrict__ L,
                                                                 float4* __restrict__ nU, float w_reg, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float d = pnp[i].x - pg[i].x;
    nU[i]   = add4(mul4(grad[i], d), mul4(L[i], w_reg));
}

// convolution_{rows,columns,depth}_kernel -- solver.cu:237-446: sum = 0; for j=-3..3: sum += S[3-j]*src(clamp(i+j))
template <int AXIS>
__global__ void __launch_bounds__(256) conv1d_kernel(float4* __restrict__ dst, const float4* __restrict__ src, Taps S, Dims d) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z;
    if (x >= d.x || y >= d.y) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        int xx = x, yy = y, zz = z;
        if (AXIS == 0) xx = min(max(x + j, 0), d.x - 1);
        if (AXIS == 1) yy = min(max(y + j, 0), d.y - 1);
        if (AXIS == 2) zz = min(max(z + j, 0), d.z - 1);
        float4 v = src[vidx(d, xx, yy, zz)];
        float s  = S.s[3 - j];
        sx += v.x * s;
        sy += v.y * s;
        sz += v.z * s;
    }
    float4* o = dst + vidx(d, x, y, z);
    if (AXIS == 0) {
        *o = f4(sx, sy, sz);  // rows assign (solver.cu:290)
    } else {                  // columns / depth accumulate, w untouched (solver.cu:366,443; utils.hpp:253-258)
        float4 c = *o;
        c.x += sx;
        c.y += sy;
        c.z += sz;
        *o = c;
    }
}

// update_psi_kernel -- solver.cu:53-69
__global__ void __launch_bounds__(256) update_psi_kernel(float4* __restrict__ psi, const float4* __restrict__ nUS,
                                                         float4* __restrict__ updates, float alpha, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 u   = mul4(nUS[i], alpha);
    updates[i] = u;
    float4 p   = psi[i];
    p.x -= u.x;
    p.y -= u.y;
    p.z -= u.z;
    psi[i] = p;
}

// ============================================================================================================
// Part 2: fused two-pass iteration
// ============================================================================================================

constexpr int TX = 64;  // tile width in lanes: one wave per tile row (32-wide tiles measured slower: profiles/LABBOOK.md, round 5)

// --- storage formats ------------------------------------------------------------------------------------------------
// API format (COMPACT = false): the reference's layouts -- psi / nabla_U float4 (w == 0), TSDF volumes float2.
// Compact format (COMPACT = true), private to the solver handle while it iterates: psi / nabla_U as packed 12-byte
// xyz triples (the w lane is a constant 0 that would cost 25 % of their traffic) and tsdf-only 4-byte copies of
// phi_global, phi_n and phi_n o psi (the weight lane is not read by the iteration; it is rebuilt once after the
// loop).  Same arithmetic on the same values => bit-identical results, 76 instead of 112 bytes per voxel-iteration.
struct P3 {  // 12-byte element of the compact fields (only used for pointer arithmetic / sizeof)
    float x, y, z;
};
typedef float v3f __attribute__((ext_vector_type(3)));
typedef v3f __attribute__((aligned(4))) v3f_u;  // 12-byte access, 4-byte aligned -> global_load/store_dwordx3
template <bool C>
SOBFU_DEV float4 ldv(const void* base, size_t i) {
    if (C) {
        v3f v = *(const v3f_u*) ((const float*) base + 3 * i);
        return make_float4(v.x, v.y, v.z, 0.f);
    }
    return ((const float4*) base)[i];
}
template <bool C>
SOBFU_DEV void stv(void* base, size_t i, const float4& v) {
    if (C) {
        v3f o = {v.x, v.y, v.z};
        *(v3f_u*) ((float*) base + 3 * i) = o;
    } else {
        ((float4*) base)[i] = v;
    }
}
// streaming variants (nontemporal hint) for data a launch touches exactly once: they should not displace the lines that
// neighbouring workgroups re-read from the XCD's L2 (nabla_U halo rows, phi_n corners)
// kNT = 3: pass B's stores of psi / phi_n o psi, its load of psi, and pass A's load of phi_global carry the hint on grids beyond the
// Infinity Cache (template NTL = kNT; 0 on cache-resident grids).  Hinting the nabla_U store / pass A's inner rows as well was
// within run-to-run noise (profiles/LABBOOK.md, round 5) and is gone.
// NB (found in the ISA in round 5): hipcc keeps the hint of __builtin_nontemporal_load / _store on 4-byte accesses (`global_store_dword
// ... nt`: phi_n o psi, phi_global, F) but DROPS it on the 4-byte-aligned 12-byte vector type -- the 12-byte variants below compile to
// plain `global_load/store_dwordx3`.  Where the hint on a 12-byte access matters it goes through a buffer instruction, whose cache-policy
// operand carries it (buf_ld3 / buf_st3: the pipelined march, and the plain march's psi load / store: template NTBUF, + 3.4 %
// iterations/s at 256^3).
constexpr int kNT = 3;
// The same accesses as (uniform plane pointer) + (32-bit byte offset of the lane's cell in the plane): the address is a scalar base
// plus one 32-bit lane register (global_load ... v_off, s[base]) instead of a 64-bit lane address per stream.
template <bool C>
SOBFU_DEV float4 ldvb(const char* plane_ptr, uint32_t byte_off, bool nt = false) {
    const char* p = plane_ptr + (size_t) byte_off;
    if (C) {
        v3f v = nt ? __builtin_nontemporal_load((const v3f_u*) p) : *(const v3f_u*) p;
        return make_float4(v.x, v.y, v.z, 0.f);
    }
    typedef float v4f __attribute__((ext_vector_type(4)));
    if (nt) {
        v4f v = __builtin_nontemporal_load((const v4f*) p);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *(const float4*) p;
}
template <bool C>
SOBFU_DEV void stvb(char* plane_ptr, uint32_t byte_off, const float4& v, bool nt = false) {
    char* p = plane_ptr + (size_t) byte_off;
    if (C) {
        v3f o = {v.x, v.y, v.z};
        if (nt) __builtin_nontemporal_store(o, (v3f_u*) p);
        else *(v3f_u*) p = o;
    } else {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f o = {v.x, v.y, v.z, v.w};
        if (nt) __builtin_nontemporal_store(o, (v4f*) p);
        else *(v4f*) p = o;
    }
}
template <bool C>
SOBFU_DEV float ldt(const void* base, size_t i) {  // tsdf of voxel i
    return C ? ((const float*) base)[i] : ((const float2*) base)[i].x;
}
// interpolate_tsdf on a tsdf-only volume (utils.hpp:50-86 without the weight fetch)
SOBFU_DEV float interp_tsdf_only(const float* __restrict__ v, const Dims& d, float px, float py, float pz) {
    Tri a = tri_setup(px, d.x), b = tri_setup(py, d.y), c = tri_setup(pz, d.z);
    const size_t sy = (size_t) d.x, sz = (size_t) d.x * d.y;
    const float* pg = v + (size_t) b.g * sy + (size_t) c.g * sz;
    const size_t dy = (size_t) (b.h - b.g) * sy, dz = (size_t) (c.h - c.g) * sz;
    float hhh = pg[a.h + dy + dz], hhg = pg[a.h + dy], hgh = pg[a.h + dz], hgg = pg[a.h];
    float ghh = pg[a.g + dy + dz], ghg = pg[a.g + dy], ggh = pg[a.g + dz], ggg = pg[a.g];
    return lerp1(lerp1(lerp1(hhh, hhg, c.t), lerp1(hgh, hgg, c.t), b.t), lerp1(lerp1(ghh, ghg, c.t), lerp1(ggh, ggg, c.t), b.t), a.t);
}

// The same sampler with 32-bit BYTE offsets from the (uniform) volume base: one scalar base + a 32-bit lane offset per corner
// (global_load_dword v, v_off, s[base]) instead of eight 64-bit lane addresses -- ~20 VALU fewer per voxel.  Valid while the
// tsdf-only volume is < 4 GiB (< 2^30 voxels); same loads, same lerp chain, same bits.
SOBFU_DEV float interp_tsdf_only32(const float* __restrict__ v, const Dims& d, float px, float py, float pz) {
    const Tri a = tri_setup(px, d.x), b = tri_setup(py, d.y), c = tri_setup(pz, d.z);
    const uint32_t sy = 4u * (uint32_t) d.x, sz = sy * (uint32_t) d.y;
    const uint32_t o  = 4u * (uint32_t) a.g + sy * (uint32_t) b.g + sz * (uint32_t) c.g;
    const uint32_t ox = a.h != a.g ? 4u : 0u, oy = b.h != b.g ? sy : 0u, oz = c.h != c.g ? sz : 0u;
    const char* base = (const char*) v;
    auto at = [&](uint32_t off) { return *(const float*) (base + (size_t) off); };
    const float hhh = at(o + ox + oy + oz), hhg = at(o + ox + oy), hgh = at(o + ox + oz), hgg = at(o + ox);
    const float ghh = at(o + oy + oz), ghg = at(o + oy), ggh = at(o + oz), ggg = at(o);
    return lerp1(lerp1(lerp1(hhh, hhg, c.t), lerp1(hgh, hgg, c.t), b.t), lerp1(lerp1(ghh, ghg, c.t), lerp1(ggh, ggg, c.t), b.t), a.t);
}

// interp_tsdf_only32 in two halves, for the software-pipelined pass B: the eight corner loads are ISSUED when a plane's psi is
// known and CONSUMED one plane later (the gather's round trip then overlaps the next plane's barrier and taps instead of ending
// every plane's dependent chain).  Same loads, same lerp chain, same bits.
struct Gather8 {
    float c[8];  // hhh, hhg, hgh, hgg, ghh, ghg, ggh, ggg
    float ta, tb, tc;
};
SOBFU_DEV Gather8 gather_issue32(const float* __restrict__ v, const Dims& d, float px, float py, float pz) {
    const Tri a = tri_setup(px, d.x), b = tri_setup(py, d.y), c = tri_setup(pz, d.z);
    const uint32_t sy = 4u * (uint32_t) d.x, sz = sy * (uint32_t) d.y;
    const uint32_t o  = 4u * (uint32_t) a.g + sy * (uint32_t) b.g + sz * (uint32_t) c.g;
    const uint32_t ox = a.h != a.g ? 4u : 0u, oy = b.h != b.g ? sy : 0u, oz = c.h != c.g ? sz : 0u;
    const char* base = (const char*) v;
    auto at = [&](uint32_t off) { return *(const float*) (base + (size_t) off); };
    Gather8 g;
    g.c[0] = at(o + ox + oy + oz); g.c[1] = at(o + ox + oy); g.c[2] = at(o + ox + oz); g.c[3] = at(o + ox);
    g.c[4] = at(o + oy + oz); g.c[5] = at(o + oy); g.c[6] = at(o + oz); g.c[7] = at(o);
    g.ta = a.t; g.tb = b.t; g.tc = c.t;
    return g;
}
SOBFU_DEV float gather_finish(const Gather8& g) {
    return lerp1(lerp1(lerp1(g.c[0], g.c[1], g.tc), lerp1(g.c[2], g.c[3], g.tc), g.tb), lerp1(lerp1(g.c[4], g.c[5], g.tc), lerp1(g.c[6], g.c[7], g.tc), g.tb), g.ta);
}

// --- workgroup -> tile map ---------------------------------------------------------------------------------------
// A launch produces up to kMaxBoxes BOXES of cells of the (local) array: the whole volume on a single GPU; on a multi-GPU
// tile, the owned cells (plus the one-cell shells pass B refreshes) or the boundary / interior regions of an overlapped
// schedule.  Workgroups are numbered box after box.
//   MARCHING box (kind 0): x-tile (64 lanes) fastest, then y-tile, then z-chunk; a workgroup marches its z-chunk with the
//       register / LDS pipeline described above.
//   DIRECT box (kind 1): one lane per cell, every tap read straight from the L1 / L2 -- for THIN regions (the one-cell x / y
//       shells of a tile, the 4-cell faces and 4 x 4 edge strips that travel to the neighbours), where a march would either
//       leave 63 of 64 lanes idle (regions thin in x), waste most of an 8-row tile (thin in y) or be all prologue (thin in
//       z).  A wave covers a (wx x 64/wx) patch of an x-y plane, wx = min(64, pow2ceil(x extent)): coalesced along x as far
//       as the box allows; no LDS, no barrier, one round trip.  Such boxes hold a few per cent of the cells, so the ~20
//       cached loads a cell costs this way do not matter; the arithmetic is op for op the marching path's.
//
// With the XCD swizzle the linear id is first remapped so that each XCD (workgroup b runs on XCD b % 8 -- observed, used
// for speed only) owns a contiguous run of tiles and serves neighbour-tile halos from its own L2.  PMC (256^3): fabric
// bytes per launch drop 1.013 -> 0.821 GB for pass A and 1.406 -> 1.286 GB for pass B.
constexpr int kMaxBoxes = 6;
struct Box {
    int x0, x1, y0, y1, z0, z1;  // cells [x0, x1) x [y0, y1) x [z0, z1)
    int zc;                      // marching: planes per march (z-chunk); direct: wx, the lanes of a wave that run along x
    int kind;                    // 0 marching, 1 direct
    int wpg;                     // direct: waves of a workgroup that take cells (the others leave at once) -- see direct_wpg()
    int rem;                     // marching: the first `rem` z-chunks march zc + 1 planes (an even split of the planes over a chosen NUMBER of chunks)
    int pair;                    // marching, pass B: z-chunks march in alternating directions (even chunks top-down, odd ones bottom-up), so that
                                 // two neighbours start at -- or arrive at -- their common boundary TOGETHER: the 6 planes either side of it,
                                 // which both read, are fetched once where the two share an XCD (and its L2) instead of a march apart
};
struct BoxList {
    int n;
    int m0, m1;   // workgroups [m0, m1) belong to marching boxes, the rest to direct boxes
    Box b[kMaxBoxes];
    int first[kMaxBoxes + 1];  // first workgroup of box i; first[n] = workgroups in the launch
};
struct TileGeom {
    int u0, v0, zb, ze;  // tile origin along x / y, planes [zb, ze) of this march
    int u_hi, v_hi;      // cells with x >= u_hi or y >= v_hi are outside the box (computed, not stored)
    int DU, DV;          // array extents along x / y
    bool down;           // the march runs from plane ze - 1 down to zb (Box::pair)
};
SOBFU_DEV unsigned xcd_swizzle(unsigned t, unsigned nb) {
    const unsigned q = nb / 8u, rem = nb % 8u, xcd = t % 8u, slot = t / 8u;
    return xcd * q + min(xcd, rem) + slot;  // bijective for any nb
}
// Position of workgroup t inside a box of n workgroups numbered [first, first + n) in DISPATCH order -- which hands consecutive
// workgroups to consecutive XCDs -- such that every XCD gets a CONTIGUOUS run of the box's own order (x fastest, then y, then z):
// the workgroups of a thin box that share cache lines (the same rows one plane up or down, the rows next door) then share an L2
// too, while the box as a whole stays spread over all eight XCDs and over time exactly as before.
SOBFU_DEV unsigned box_xcd_order(unsigned t, unsigned first, unsigned n) {
    const unsigned d = (t - first) % 8u, slot = (t - first) / 8u;  // d: which of the box's eight interleaved streams; same XCD <=> same d
    unsigned pre = 0;
#pragma unroll
    for (unsigned k = 0; k < 7u; ++k)
        if (k < d) pre += n > k ? (n - k + 7u) / 8u : 0u;  // members of stream k
    return pre + slot;
}
#ifndef SOBFU_BOX_XCD
#define SOBFU_BOX_XCD 3  // bit 0: push / direct boxes of a tile's pass A, bit 1: direct boxes of pass B take the XCD-contiguous order
#endif
// marching geometry of workgroup t inside box b whose first workgroup is `first` (all scalar)
SOBFU_DEV TileGeom geom_in_box(const Box& b, unsigned t, int first, const Dims& d, int ty) {
    t -= (unsigned) first;
    TileGeom g;
    g.u_hi = b.x1;
    g.v_hi = b.y1;
    g.DU   = d.x;
    g.DV   = d.y;
    const unsigned ntu = (unsigned) ((b.x1 - b.x0 + TX - 1) / TX), ntv = (unsigned) ((b.y1 - b.y0 + ty - 1) / ty);
    g.u0 = b.x0 + (int) (t % ntu) * TX;
    g.v0 = b.y0 + (int) ((t / ntu) % ntv) * ty;
    const int ck = (int) (t / (ntu * ntv));
    g.zb = b.z0 + ck * b.zc + min(ck, b.rem);
    g.ze = min(g.zb + b.zc + (ck < b.rem ? 1 : 0), b.z1);
    g.down = b.pair != 0 && (ck & 1) == 0;
    return g;
}
// the cell of this lane in a DIRECT box; false: the lane has none
SOBFU_DEV bool direct_cell(const Box& b, unsigned t, int first, int& x, int& y, int& z) {
    const int wx = b.zc, wyl = 64 / wx;
    const unsigned ntx = (unsigned) ((b.x1 - b.x0 + wx - 1) / wx), nty = (unsigned) ((b.y1 - b.y0 + wyl - 1) / wyl);
    const unsigned wv = (unsigned) __builtin_amdgcn_readfirstlane((int) threadIdx.y);
    if (wv >= (unsigned) b.wpg) return false;
    const unsigned w = (t - (unsigned) first) * (unsigned) b.wpg + wv;  // wave of the box
    const int lane = threadIdx.x;
    x = b.x0 + (int) (w % ntx) * wx + (lane & (wx - 1));
    y = b.y0 + (int) ((w / ntx) % nty) * wyl + lane / wx;
    z = b.z0 + (int) (w / (ntx * nty));
    return z < b.z1 && x < b.x1 && y < b.y1;
}
// box of workgroup t (constant indices only: a dynamically indexed by-value argument would be copied to scratch)
SOBFU_DEV Box find_box(const BoxList& L, unsigned t, int& first, int* count = nullptr) {
    Box b = L.b[0];
    first = 0;
    int next = L.first[1];
#pragma unroll
    for (int k = 1; k < kMaxBoxes; ++k)
        if (k < L.n && (int) t >= L.first[k]) {
            b     = L.b[k];
            first = L.first[k];
            next  = L.first[k + 1];
        }
    if (count) *count = next - first;
    return b;
}

// --- convergence gate ----------------------------------------------------------------------------------------
// Pass B folds max ||u||^2 of iteration k into 256 uint32 slots (non-negative floats order like their bit
// patterns).  A kernel of iteration k+1 receives the slots of iteration k and returns immediately when
// sqrt_rd(max) <= max_update_norm -- the reference's `break` (solver.cu:183) without a host round trip.
//
// prev_rows = 2 (native tiled loop, late gate): the gate is true when the row at prev_slots OR the row before it says
// "converged".  That loop gates iteration j on row j-2, runs iteration k+1 speculatively after the threshold fired at k, and
// a gated launch leaves its own row at the all-zero (= converged) state it was cleared to -- looking at rows j-2 and j-3
// makes the stop sticky for both parities whatever the speculative row k+1 holds.
SOBFU_DEV bool solver_converged(const uint32_t* __restrict__ prev_slots, float max_update_norm, int prev_rows = 1) {
    if (prev_slots == nullptr) return false;
    __shared__ int s_flag;
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    if (tid < 64) {
        bool conv = false;
        for (int r = 0; r < prev_rows; ++r) {
            const uint32_t* row = prev_slots - (size_t) r * 256;
            uint32_t m = max(max(row[tid], row[tid + 64]), max(row[tid + 128], row[tid + 192]));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
            conv = conv || sqrt_rd(__uint_as_float(m)) <= max_update_norm;
        }
        if (tid == 0) s_flag = conv ? 1 : 0;
    }
    __syncthreads();
    return s_flag != 0;
}

// The same gate in two halves for the marching kernels: the slot loads are issued FIRST, the z-pipeline prologue loads
// behind them, and the verdict is formed after that -- the gate's L2 round trip overlaps the prologue's instead of
// preceding it (the marching loop itself is untouched).
struct GateRegs {
    uint32_t v[8];
};
SOBFU_DEV GateRegs gate_load(const uint32_t* __restrict__ prev_slots, int prev_rows, bool sys = false /* the rows hold entries other GPUs stored */) {
    GateRegs g;
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
#pragma unroll
    for (int k = 0; k < 8; ++k) g.v[k] = 0xffffffffu;  // "not converged" filler for rows that are not looked at
    if (prev_slots != nullptr && tid < 64) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < prev_rows) {
                const uint32_t* row = prev_slots - (size_t) r * 256;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    g.v[4 * r + k] = sys ? __hip_atomic_load(row + tid + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : row[tid + 64 * k];
            }
    }
    return g;
}
SOBFU_DEV bool gate_decide(const GateRegs& g, const uint32_t* __restrict__ prev_slots, float max_update_norm) {
    if (prev_slots == nullptr) return false;
    __shared__ int s_flag;
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    if (tid < 64) {
        bool conv = false;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            uint32_t m = max(max(g.v[4 * r], g.v[4 * r + 1]), max(g.v[4 * r + 2], g.v[4 * r + 3]));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
            conv = conv || sqrt_rd(__uint_as_float(m)) <= max_update_norm;  // 0xffffffff is a NaN pattern: never <=
        }
        if (tid == 0) s_flag = conv ? 1 : 0;
    }
    __syncthreads();
    return s_flag != 0;
}

// --- pass A ----------------------------------------------------------------------------------------------------
struct PassACore {
    const void* pnp;  // phi_n o psi
    const void* pg;   // phi_global
    const void* psi;
    void* nU;
    Dims d;  // extents of the (local) arrays
    float w_reg;
    const uint32_t* prev_slots;
    float max_update_norm;
};
struct PassAArgs {
    PassACore c;
    BoxList boxes;  // the cells this launch produces
};

// Pass B's marching loops store under a per-lane "this cell is mine" test, and the compiler sinks everything that consumes the step's
// loads into that branch with the stores.  On the path around the branch it must then assume the loads still in flight, so at
// the join -- the pipeline shift, the next step's halo staging -- it waits for vmcnt(0), which on the path that DID store also
// waits for the stores' acknowledgement: once per plane per wave, on the critical chain.  Pinning the value about to be stored
// in front of the branch makes the wait for its loads unconditional (same place: behind the arithmetic), the join then knows
// that every load has landed, and the stores drain behind the next plane's work.  (Pass A gains nothing from the same pin: measured,
// profiles/LABBOOK.md round 4.)
SOBFU_DEV void pin3(const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z)); }

// --- buffer addressing (cache-resident launches) -------------------------------------------------------------------------------
// A 128-bit buffer resource in SGPRs (base, bytes) + a 32-bit lane byte offset + a scalar byte offset (the plane): an address costs
// no vector instruction and no 64-bit lane register pair.  Arrays below 4 GiB only (checked at launch).
typedef unsigned v3u __attribute__((ext_vector_type(3)));
SOBFU_DEV __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int) bytes, 0x00020000);
}
// nt: the streaming (nontemporal) hint, bit 1 of the cache-policy operand on gfx94x / gfx950
SOBFU_DEV float4 buf_ld3(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, bool nt = false) {
    const v3u t = nt ? __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 2) : __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), 0.f);
}
// the same load at SYSTEM scope (sc0 sc1: bits 0 and 4 of the cache-policy operand) when `sys` (wave-uniform) says so: cells another
// GPU stored -- the halo rims of nabla_U on the direct transport
SOBFU_DEV float4 buf_ld3_scope(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, bool sys) {
    const v3u t = sys ? __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 17) : __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), 0.f);
}
SOBFU_DEV float buf_ld1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int) voff, (int) soff, 0));
}
SOBFU_DEV void buf_st3(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, const float4& v, bool nt = false) {
    const v3u t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
    if (nt) __builtin_amdgcn_raw_buffer_store_b96(t, r, (int) voff, (int) soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b96(t, r, (int) voff, (int) soff, 0);
}
SOBFU_DEV void buf_st1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float v, bool nt = false) {
    if (nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int) voff, (int) soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int) voff, (int) soff, 0);
}

// One cell of pass A from its centre c = psi, fc = (phi_n o psi).tsdf, bg = phi_global.tsdf and the six RAW neighbours of psi
// (p**) and of F (f**) along x (l), y (r), z -- loaded with clamped indices; the boundary rules are applied here.  Shared by
// the marching and the direct path: the same operations in the same order.
SOBFU_DEV float4 potential_gradient_cell(const float4& c, float fc, float bg, float4 plp, float4 plm, float4 prp, float4 prm, float4 pzp,
                                         float4 pzm, float flp, float flm, float frp, float frm, float fzp, float fzm, bool ulo, bool uhi,
                                         bool vlo, bool vhi, bool zlo, bool zhi, float w_reg) {
    // TsdfDifferentiator boundary rule (vector_fields.cu:165-191): mirror the missing neighbour
    const float gl1 = uhi ? flm : flp, gl2 = ulo ? flp : flm;
    const float gr1 = vhi ? frm : frp, gr2 = vlo ? frp : frm;
    const float gz1 = zhi ? fzm : fzp, gz2 = zlo ? fzp : fzm;
    const float4 g = f4((gl1 - gl2) / 2.f, (gr1 - gr2) / 2.f, (gz1 - gz2) / 2.f);
    // SecondOrderDifferentiator boundary rule (vector_fields.cu:299-331): both neighbours <- centre
    if (ulo || uhi) { plp = c; plm = c; }
    if (vlo || vhi) { prp = c; prm = c; }
    if (zlo || zhi) { pzp = c; pzm = c; }
    // the reference adds x+, x-, y+, y-, z+, z- in that order
    float4 vv = mul4(c, -6.f);
    vv = add4(vv, plp);
    vv = add4(vv, plm);
    vv = add4(vv, prp);
    vv = add4(vv, prm);
    vv = add4(vv, pzp);
    vv = add4(vv, pzm);
    const float4 L = mul4(vv, -1.f);
    // calculate_potential_gradient_kernel (solver.cu:28-31)
    const float diff = fc - bg;
    return add4(mul4(g, diff), mul4(L, w_reg));
}

// DIRECT evaluation of one cell of pass A (thin boxes): 7 psi + 7 F + 1 G loads, all but a few of them cache hits
template <bool COMPACT>
SOBFU_DEV float4 pass_a_direct_cell(const PassACore& a, int x, int y, int z) {
    const Dims d = a.d;
    const int xm = max(x - 1, 0), xp = min(x + 1, d.x - 1), ym = max(y - 1, 0), yp = min(y + 1, d.y - 1), zm = max(z - 1, 0), zp = min(z + 1, d.z - 1);
    const size_t i = vidx(d, x, y, z), ixm = vidx(d, xm, y, z), ixp = vidx(d, xp, y, z), iym = vidx(d, x, ym, z), iyp = vidx(d, x, yp, z),
                 izm = vidx(d, x, y, zm), izp = vidx(d, x, y, zp);
    const float4 c = ldv<COMPACT>(a.psi, i);
    const float4 plp = ldv<COMPACT>(a.psi, ixp), plm = ldv<COMPACT>(a.psi, ixm), prp = ldv<COMPACT>(a.psi, iyp), prm = ldv<COMPACT>(a.psi, iym),
                 pzp = ldv<COMPACT>(a.psi, izp), pzm = ldv<COMPACT>(a.psi, izm);
    const float fc = ldt<COMPACT>(a.pnp, i), flp = ldt<COMPACT>(a.pnp, ixp), flm = ldt<COMPACT>(a.pnp, ixm), frp = ldt<COMPACT>(a.pnp, iyp),
                frm = ldt<COMPACT>(a.pnp, iym), fzp = ldt<COMPACT>(a.pnp, izp), fzm = ldt<COMPACT>(a.pnp, izm);
    const float bg = ldt<COMPACT>(a.pg, i);
    return potential_gradient_cell(c, fc, bg, plp, plm, prp, prm, pzp, pzm, flp, flm, frp, frm, fzp, fzm, x == 0, x == d.x - 1, y == 0,
                                   y == d.y - 1, z == 0, z == d.z - 1, a.w_reg);
}

// the MARCHING path of pass A for the tile tg (a z-chunk of a 64 x TY tile)
// where the cells of a PUSH box go (pass A of a multi-GPU tile: see tile_potential_gradient_kernel)
struct PushDst {
    float* base;             // null: the box is stored locally
    int ox, oy, oz, px, py;  // cell (x, y, z) -> base + 3 * ((x + ox) + px * ((y + oy) + py * (z + oz)))
    int y0, y1, lz0, lz1;    // marching push boxes: rows [y0, y1) travel; planes [lz0, lz1) are stored locally as well
};
SOBFU_DEV void st3_system(float* p, const float4& v);

// NTL: streaming (nontemporal) hints, kNT or 0 -- for grids whose state exceeds the 256 MiB Infinity Cache; 0 for cache-resident
// ones (multi-GPU tiles, small grids), where the hints keep the data the NEXT launch reads out of the cache (2 x 2 x 2 tile of
// 256^3: 54.4 -> 48.8 us per iteration without them)
template <int RPT, int WY, bool COMPACT, int NTL, bool PUSHABLE = false>
SOBFU_DEV void pass_a_march(const PassACore& a, const TileGeom& tg, const GateRegs& gate, const PushDst* pd = nullptr) {
    constexpr int TY = RPT * WY, LW = TX + 2, LH = TY + 2;
    constexpr int NXH = (2 * TY + TX - 1) / TX;  // row-tasks for the two lane-halo columns
    constexpr int NTASK = 2 + NXH, TPW = (NTASK + WY - 1) / WY;
    __shared__ float4 t_psi[2][LH][LW + 2];  // {psi.xyz, F = (phi_n o psi).tsdf} -- psi.w is never read

    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    const int u0 = tg.u0, v0 = tg.v0, zb = tg.zb, ze = tg.ze;
    const int u = u0 + lx, uc = min(u, tg.DU - 1);
    const size_t plane = (size_t) d.x * d.y, sv = (size_t) d.x;

    size_t off[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) off[r] = (size_t) uc + sv * (size_t) min(v0 + wy * RPT + r, tg.DV - 1);
    // halo tasks: task 0 = row above the tile, task 1 = row below, tasks 2.. = lane-halo cells (col -1 / col TX)
    int h_lr[TPW], h_lc[TPW];  // LDS cell
    size_t h_off[TPW];
    bool h_on[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        int task = wy + k * WY;
        h_on[k]  = task < NTASK;
        int lr = 0, lc = 0;
        if (task == 0) { lr = 0; lc = lx + 1; }
        else if (task == 1) { lr = LH - 1; lc = lx + 1; }
        else {
            int e = (task - 2) * TX + lx;  // 0 .. 2*TY-1
            h_on[k] = h_on[k] && e < 2 * TY;
            lr = 1 + (e >> 1);
            lc = (e & 1) ? LW - 1 : 0;
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        int gu = min(max(u0 - 1 + lc, 0), tg.DU - 1), gv = min(max(v0 - 1 + lr, 0), tg.DV - 1);
        h_off[k] = (size_t) gu + (size_t) gv * sv;
    }

    // z register pipeline: m = z-1, c = z, n = z+1 (clamped loads; boundary rules applied at use)
    float4 pm[RPT], pc[RPT], pn[RPT];
    float fm[RPT], fc[RPT], fn[RPT];
    float4 hp[TPW];
    float hf[TPW];
    float bg[RPT], bgn[RPT];  // phi_global of plane z, requested one step ahead like everything else (no same-step round trip)
    auto ld_bg = [&](size_t i) { return (NTL >= 3 && COMPACT) ? __builtin_nontemporal_load((const float*) a.pg + i) : ldt<COMPACT>(a.pg, i); };
    {
        const size_t zm = (size_t) max(zb - 1, 0) * plane, zc0 = (size_t) zb * plane;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pm[r] = ldv<COMPACT>(a.psi, zm + off[r]);
            fm[r] = ldt<COMPACT>(a.pnp, zm + off[r]);
            pc[r] = ldv<COMPACT>(a.psi, zc0 + off[r]);
            fc[r] = ldt<COMPACT>(a.pnp, zc0 + off[r]);
            bg[r] = ld_bg(zc0 + off[r]);
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
                hp[k] = ldv<COMPACT>(a.psi, zc0 + h_off[k]);
                hf[k] = ldt<COMPACT>(a.pnp, zc0 + h_off[k]);
            }
    }
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;

    const bool ulo = (u == 0), uhi = (u == tg.DU - 1);
    for (int z = zb; z < ze; ++z) {
        const int buf = (z - zb) & 1;
        // stage plane z
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            t_psi[buf][wy * RPT + r + 1][lx + 1] = make_float4(pc[r].x, pc[r].y, pc[r].z, fc[r]);
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) t_psi[buf][h_lr[k]][h_lc[k]] = make_float4(hp[k].x, hp[k].y, hp[k].z, hf[k]);
        // prefetch plane z+1 (main) and the halo of plane z+1
        const size_t zn = (size_t) min(z + 1, d.z - 1) * plane, zcur = (size_t) z * plane;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pn[r] = ldv<COMPACT>(a.psi, zn + off[r]);
            fn[r] = ldt<COMPACT>(a.pnp, zn + off[r]);
            if (z + 1 < ze) bgn[r] = ld_bg(zn + off[r]);
        }
        if (z + 1 < ze) {
#pragma unroll
            for (int k = 0; k < TPW; ++k)
                if (h_on[k]) {
                    hp[k] = ldv<COMPACT>(a.psi, zn + h_off[k]);
                    hf[k] = ldt<COMPACT>(a.pnp, zn + h_off[k]);
                }
        }
        __syncthreads();

        const bool zlo = (z == 0), zhi = (z == d.z - 1);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int v  = v0 + wy * RPT + r;
            const int lr = wy * RPT + r + 1;
            const bool vlo = (v == 0), vhi = (v == tg.DV - 1);
            // raw neighbours along x (l*) and y (r*)
            const float4 plp = t_psi[buf][lr][lx + 2], plm = t_psi[buf][lr][lx];
            float4 prp, prm;
            float frp, frm;
            if (r + 1 < RPT) { prp = pc[r + 1 < RPT ? r + 1 : r]; frp = fc[r + 1 < RPT ? r + 1 : r]; }
            else { prp = t_psi[buf][lr + 1][lx + 1]; frp = prp.w; }
            if (r > 0) { prm = pc[r > 0 ? r - 1 : r]; frm = fc[r > 0 ? r - 1 : r]; }
            else { prm = t_psi[buf][lr - 1][lx + 1]; frm = prm.w; }
            const float4 o = potential_gradient_cell(pc[r], fc[r], bg[r], plp, plm, prp, prm, pn[r], pm[r], plp.w, plm.w, frp, frm, fn[r], fm[r], ulo,
                                                     uhi, vlo, vhi, zlo, zhi, a.w_reg);
            if (u < tg.u_hi && v < tg.v_hi) {
                const size_t i = zcur + off[r];  // inside the box no clamp was active: off[r] is the cell itself
                if (PUSHABLE && pd->base != nullptr) {  // a marching PUSH box: the rows of the message go to their destination ...
                    if (v >= pd->y0 && v < pd->y1) {
                        const size_t j = (size_t) (u + pd->ox) + (size_t) pd->px * ((size_t) (v + pd->oy) + (size_t) pd->py * (size_t) (z + pd->oz));
                        st3_system(pd->base + 3 * j, o);
                    }
                    if (z >= pd->lz0 && z < pd->lz1) stv<COMPACT>(a.nU, i, o);  // ... and where the box stands in for the owned block, home too
                } else stv<COMPACT>(a.nU, i, o);
            }
        }
        // shift the z pipeline
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pm[r] = pc[r];
            pc[r] = pn[r];
            fm[r] = fc[r];
            fc[r] = fn[r];
            bg[r] = bgn[r];
        }
    }
}


template <int RPT, int WY, bool COMPACT, int NTL>
__global__ void __launch_bounds__(TX* WY) fused_potential_gradient_kernel(PassAArgs a) {
    const GateRegs gate = gate_load(a.c.prev_slots, 1);
    const unsigned t    = xcd_swizzle(blockIdx.x, (unsigned) a.boxes.first[a.boxes.n]);
    int first;
    const Box b = find_box(a.boxes, t, first);
    pass_a_march<RPT, WY, COMPACT, NTL>(a.c, geom_in_box(b, t, first, a.c.d, RPT * WY), gate);
}

// ---- pass A of a multi-GPU TILE: the halo exchange is part of the launch ---------------------------------------------------
// Boxes of a tile launch, in workgroup order:
//   PUSH boxes: the cells of one halo message (a 4-cell face or a 4 x 4 edge strip of the owned block) are evaluated a second
//       time -- lane per cell where the box is thin in x, by a short march where its rows are wide -- and stored STRAIGHT INTO
//       THE DESTINATION -- the halo cells of the neighbour's nabla_U
//       array, peer-mapped over xGMI (direct transport), or this rank's packed send buffer (RCCL / callback transports): no
//       pack kernel, no unpack kernel and, with the direct transport, no communication launch at all.  They are numbered
//       first, so they leave while the owned block is still being computed.
//   the owned block (marching), stored locally.
// Synchronisation of the direct transport, in the kernel's tail (TileSync): every workgroup that pushed waits for its stores'
// acknowledgements and takes a ticket; the LAST of them writes this rank's arrival flag (= the iteration's sequence number) at
// every rank of the sync set, then waits until the flags of all those ranks have reached the sequence number (with a
// deadline): a launch retires when all its workgroups have, so when this one does every halo cell of the iteration has landed
// and pass B -- a separate launch, whose start invalidates the caches -- reads it.  nabla_U is
// double-buffered by iteration parity, which orders a neighbour's stores of iteration k+1 behind this rank's reads of
// iteration k without a second handshake (see tiled_capi.hip).
constexpr int kMaxTileBoxes = 20;  // 18 messages + the owned block + one spare
struct TileBox {
    Box b;
    PushDst push;
};
struct TileBoxList {
    int n, n_push_wgs;  // workgroups [0, n_push_wgs) belong to push boxes
    TileBox b[kMaxTileBoxes];
    int first[kMaxTileBoxes + 1];
};
// ---- stores that leave the GPU ----------------------------------------------------------------------------------------------
// What travels to a peer (message cells, row maxima, flags) is stored WRITE-THROUGH at system scope (sc0 sc1): it never sits
// dirty in this GPU's write-back L2, so "everything I sent has arrived" is `s_waitcnt vmcnt(0)` -- the stores' acknowledgements --
// and not the L2 write-back a system-scope release fence would do (pass A is filling that L2 with nabla_U at the time: one such
// fence per push workgroup cost 4x the whole iteration).  The flag goes out after the wait, so it cannot overtake the data.
SOBFU_DEV void st3_system(float* p, const float4& v) {
    const v3f o = {v.x, v.y, v.z};
    asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" ::"v"(p), "v"(o) : "memory");
}
SOBFU_DEV void st1_system(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
SOBFU_DEV uint32_t ld1_system(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
SOBFU_DEV void stores_acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wave 0 of one workgroup: the maximum of this rank's slot row -> entry `my_rank` of that row at every rank of the sync set (and
// here); the next signal covers these stores
SOBFU_DEV void tile_row_push(const TileSync* sy, const uint32_t* row, uint32_t row_index) {
    const int l = threadIdx.x;
    uint32_t m = max(max(row[l], row[l + 64]), max(row[l + 128], row[l + 192]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
    const size_t e = (size_t) row_index * 256u + sy->my_rank;
    if (l == 0) sy->my_grows[e] = m;
    for (int q = l; q < (int) sy->n_sync; q += 64) st1_system(sy->peer_grows[q] + e, m);
}
// one lane: raise this rank's arrival flag at every rank of the sync set (after stores_acknowledged() on everything it covers)
SOBFU_DEV void tile_signal(const TileSync* sy, uint32_t seq) {
    for (uint32_t q = 0; q < sy->n_sync; ++q) st1_system(sy->peer_flags[q] + sy->my_rank, seq);
}
// one lane: wait until every rank of the sync set has raised its flag to `seq` -- with a deadline: a missing peer is recorded
// (err = 1 + its rank) and every later wait returns at once, so a wedged neighbour never hangs this GPU.  The cells the flags
// announce are read by the NEXT launch (whose start invalidates the caches), never by this one.
SOBFU_DEV void tile_wait(TileSync* sy, uint32_t seq) {
    if (__hip_atomic_load(&sy->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const uint64_t t0 = wall_clock64();
    for (uint32_t q = 0; q < sy->n_sync; ++q) {
        const uint32_t* f = sy->my_flags + sy->sync_rank[q];
        while ((int32_t) (ld1_system(f) - seq) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > sy->timeout_ticks) {
                __hip_atomic_store(&sy->err, 1u + (uint32_t) sy->sync_rank[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
    // diagnostics (one lane per launch gets here): how long the launch sat waiting for its peers -- what of the exchange was NOT hidden
    sy->wait_ticks += wall_clock64() - t0;
    sy->wait_count += 1u;
}
// diagnostics: flag round trips with ONE peer (sync-set member q), `reps` of them inside one launch: the side that serves stores seq,
// the other answers seq + 1, ...; one lane each.  Both sides observe the same deadline as every other wait.
__global__ void __launch_bounds__(64) tile_pingpong_kernel(TileSync* sy, int q, int first, uint32_t seq0, int reps) {
    if (threadIdx.x != 0) return;
    uint32_t* theirs = sy->peer_flags[q] + sy->my_rank;
    const uint32_t* mine = sy->my_flags + sy->sync_rank[q];
    const uint64_t t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        const uint32_t s_ping = seq0 + 2u * (uint32_t) r, s_pong = s_ping + 1u;
        if (first) st1_system(theirs, s_ping);
        while ((int32_t) (ld1_system(mine) - (first ? s_pong : s_ping)) < 0) {
            if (wall_clock64() - t0 > sy->timeout_ticks) {
                __hip_atomic_store(&sy->err, 1u + (uint32_t) sy->sync_rank[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        if (!first) st1_system(theirs, s_pong);
    }
}
__global__ void __launch_bounds__(64) tile_flush_kernel(TileSync* sy, uint32_t seq, int wait, const uint32_t* row, uint32_t row_index) {
    if (row != nullptr) tile_row_push(sy, row, row_index);
    stores_acknowledged();
    if (threadIdx.x != 0) return;
    tile_signal(sy, seq);
    if (wait) tile_wait(sy, seq);
}

// the signalling part of a tile launch's arguments
struct TileSignal {
    TileSync* sync;       // null: no signalling (single-box launches, RCCL / callback transports)
    uint32_t seq;         // sequence number of this iteration
    int wait;             // the last workgroup waits for the peers' flags
    const uint32_t* row;  // this rank's max-norm slot row of the PREVIOUS iteration (null: none) ...
    uint32_t row_index;   // ... which is row `row_index` of the global rows
};
// The box list of a launch lives in DEVICE memory (a list is fixed for the life of a handle: uploaded once -- launch_tile_pass_a keeps
// every distinct list it has seen -- and read through the scalar cache), not in the kernel-argument segment: a 1.5 KB argument block
// costs a launch 0.6 us (tools/calib/launch_cost.hip: 2.9 -> 3.5 us back to back), and handed on by reference it ended up copied to
// 1.8 KB of scratch per lane (pass A 17 -> 129 us: found with SOBFU_TILED_DEBUG_SKIP=1)
struct TilePassAArgsP {
    PassACore c;
    const TileBoxList* boxes;
    TileSignal s;
};
template <int RPT, int WY, bool COMPACT, int NTL>
SOBFU_DEV void tile_potential_gradient_body(const PassACore& core, const TileBoxList& L, const TileSignal& sg) {
    const unsigned nb = (unsigned) L.first[L.n];
    // push workgroups keep their launch order (they go out first); the others are XCD-swizzled among themselves
    unsigned t = blockIdx.x;
    const bool push_wg = (int) t < L.n_push_wgs;
    if (!push_wg) t = (unsigned) L.n_push_wgs + xcd_swizzle(t - (unsigned) L.n_push_wgs, nb - (unsigned) L.n_push_wgs);
    Box b     = L.b[0].b;
    PushDst pd = L.b[0].push;
    int first = 0, next = L.first[1];
#pragma unroll
    for (int k = 1; k < kMaxTileBoxes; ++k)
        if (k < L.n && (int) t >= L.first[k]) {
            b     = L.b[k].b;
            pd    = L.b[k].push;
            first = L.first[k];
            next  = L.first[k + 1];
        }
    // inside a push box every XCD takes a contiguous run of the box's cells (see box_xcd_order)
    if (push_wg && (SOBFU_BOX_XCD & 1)) t = (unsigned) first + box_xcd_order(t, (unsigned) first, (unsigned) (next - first));
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    // the max-norm of the previous iteration, made global without a collective (workgroup 0 is a push workgroup: the signal
    // below covers these stores)
    if (sg.sync != nullptr && sg.row != nullptr && blockIdx.x == 0 && threadIdx.y == 0) tile_row_push(sg.sync, sg.row, sg.row_index);
    if (b.kind != 0) {
        int x, y, z;
        if (direct_cell(b, t, first, x, y, z)) {
            const float4 o = pass_a_direct_cell<COMPACT>(core, x, y, z);
            if (pd.base != nullptr) {
                const size_t i = (size_t) (x + pd.ox) + (size_t) pd.px * ((size_t) (y + pd.oy) + (size_t) pd.py * (size_t) (z + pd.oz));
                st3_system(pd.base + 3 * i, o);  // messages are always 12-byte cells
            } else {
                stv<COMPACT>(core.nU, vidx(core.d, x, y, z), o);
            }
        }
    } else {
        GateRegs gate;
#pragma unroll
        for (int k = 0; k < 8; ++k) gate.v[k] = 0xffffffffu;  // pass A of a tile writes scratch only: never gated
        pass_a_march<RPT, WY, COMPACT, NTL, true>(core, geom_in_box(b, t, first, core.d, RPT * WY), gate, &pd);
    }
    if (sg.sync == nullptr || !push_wg) return;
    // the push workgroups count themselves out; the LAST one raises this rank's flag at its peers and then waits for theirs: a
    // launch retires when all its workgroups have, so pass B cannot start before every neighbour's cells have landed -- while the
    // owned block's workgroups never touch the synchronisation at all
    TileSync* sy = sg.sync;
    stores_acknowledged();  // every lane: what it stored at the peers has arrived ...
    __syncthreads();        // ... before lane 0 takes the workgroup's ticket
    if (tid != 0) return;
    const uint32_t k = __hip_atomic_fetch_add(&sy->ticket_push, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == (uint32_t) L.n_push_wgs - 1u) {
        __hip_atomic_store(&sy->ticket_push, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tile_signal(sy, sg.seq);
        if (sg.wait) tile_wait(sy, sg.seq);
    }
}

template <int RPT, int WY, bool COMPACT, int NTL>
__global__ void __launch_bounds__(TX* WY) tile_potential_gradient_kernel(TilePassAArgsP a) {
    tile_potential_gradient_body<RPT, WY, COMPACT, NTL>(a.c, *a.boxes, a.s);
}

// --- pass B ----------------------------------------------------------------------------------------------------
struct PassBArgs {
    const void* nU;
    void* psi;
    const void* phi_n;
    void* pnp;        // phi_n o psi (output)
    float4* updates;  // may be null (always float4)
    uint32_t* slots;  // 256 x uint32, atomic max of ||u||^2 bit patterns
    Dims d;
    Taps S;
    float alpha;
    BoxList boxes;  // the cells this launch produces
    const uint32_t* prev_slots;
    float max_update_norm;
    // multi-GPU tiles: the fields are local arrays (d) that carry halo cells, phi_n is the whole volume (pd); only the
    // cells of `own` belong to this rank and enter the max-norm.  Single GPU: pd == d, own = everything.
    Dims pd;
    int own[6];     // x0, x1, y0, y1, z0, z1
    int prev_rows;  // rows the gate looks at (see solver_converged)
    void* psi_out;  // where the updated psi goes: == psi (in place) or the other half of a ping-pong pair (native tiled loop)
    int sys_acquire;  // direct transport: halo cells and max-norm entries of this launch's inputs were stored by OTHER GPUs (see the kernel's entry)
};

#ifndef SOBFU_PAIR_B
#define SOBFU_PAIR_B 1  // cache-resident launches of the pipelined pass B: z-chunks march in alternating directions (Box::pair)
#endif
#ifndef SOBFU_HLEAD
#define SOBFU_HLEAD 3  // planes the halo requests of pass B run ahead on long marches (0: never; one plane ahead, straight from registers)
#endif
#ifndef SOBFU_HLEAD_MIN_ZC
#define SOBFU_HLEAD_MIN_ZC 24  // shortest march (planes) that uses the halo lead
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef SOBFU_MINW_B
#define SOBFU_MINW_B 6  // waves/SIMD the register allocator must leave room for: <= 80 VGPR -> 3 workgroups of 8 waves per CU
#endif
#ifndef SOBFU_MINW_PIPE
#define SOBFU_MINW_PIPE 4  // the pipelined march: <= 128 VGPR -> 2 workgroups of 8 waves per CU
#endif
// HL: planes the halo requests run ahead of the plane they are staged for (0: one plane ahead, straight from registers -- short
// marches, where the extra prologue round trip costs more than the re-fetched halo lines).
// max ||u||^2 over the voxels a workgroup owns -> one atomicMax on one of 256 slots
template <int WY>
SOBFU_DEV void maxnorm_tail(float msq, uint32_t* slots, uint32_t* s_max) {
    const int lx = threadIdx.x, wy = threadIdx.y;
    uint32_t m = __float_as_uint(msq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
    if (lx == 0) s_max[wy] = m;
    __syncthreads();
    if (lx == 0 && wy == 0) {
#pragma unroll
        for (int w = 1; w < WY; ++w) m = max(m, s_max[w]);
        atomicMax(slots + (blockIdx.x & 255u), m);
    }
}

// DIRECT evaluation of one cell of pass B (thin boxes: the one-cell x / y shells of a tile): 19 nabla_U loads + psi + the
// phi_n gather, op for op the marching path's arithmetic (sum = 0; taps ascending j; (Sx + Sy) + Sz)
template <bool COMPACT, bool SYS>
SOBFU_DEV void direct_taps(const PassBArgs& a, int x, int y, int z, float& slx, float& sly, float& slz, float& srx, float& sry, float& srz, float& szx,
                           float& szy, float& szz) {
    const Dims d = a.d;
    const __amdgpu_buffer_rsrc_t r_nu = buf_rsrc(a.nU, SYS ? (uint32_t) ((size_t) d.x * d.y * d.z * 12) : 0u);
    auto ld_nu = [&](size_t i) { return SYS ? buf_ld3_scope(r_nu, (uint32_t) (i * 12), 0u, true) : ldv<COMPACT>(a.nU, i); };
    slx = sly = slz = srx = sry = srz = szx = szy = szz = 0.f;
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        const float s = a.S.s[3 - j];
        const float4 vl = ld_nu(vidx(d, min(max(x + j, 0), d.x - 1), y, z));
        slx += vl.x * s;
        sly += vl.y * s;
        slz += vl.z * s;
    }
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        const float s = a.S.s[3 - j];
        const float4 vr = ld_nu(vidx(d, x, min(max(y + j, 0), d.y - 1), z));
        srx += vr.x * s;
        sry += vr.y * s;
        srz += vr.z * s;
    }
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        const float s = a.S.s[3 - j];
        const float4 vz = ld_nu(vidx(d, x, y, min(max(z + j, 0), d.z - 1)));
        szx += vz.x * s;
        szy += vz.y * s;
        szz += vz.z * s;
    }
}
template <bool WRITE_UPDATES, bool COMPACT, bool IDX32>
SOBFU_DEV float pass_b_direct_cell(const PassBArgs& a, int x, int y, int z) {
    const Dims d = a.d;
    float slx, sly, slz, srx, sry, srz, szx, szy, szz;
    // nabla_U cells of the halo rims may have been stored by other GPUs (direct transport): the taps are read at system scope then
    if (COMPACT && a.sys_acquire != 0) direct_taps<COMPACT, COMPACT>(a, x, y, z, slx, sly, slz, srx, sry, srz, szx, szy, szz);
    else direct_taps<COMPACT, false>(a, x, y, z, slx, sly, slz, srx, sry, srz, szx, szy, szz);
    const float tx = (slx + srx) + szx, ty = (sly + sry) + szy, tz = (slz + srz) + szz;
    const size_t i = vidx(d, x, y, z);
    const float4 uu = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
    float4 p = ldv<COMPACT>(a.psi, i);
    p.x -= uu.x;
    p.y -= uu.y;
    p.z -= uu.z;
    stv<COMPACT>(a.psi_out, i, p);
    if (WRITE_UPDATES) a.updates[i] = uu;
    if (COMPACT) ((float*) a.pnp)[i] = IDX32 ? interp_tsdf_only32((const float*) a.phi_n, a.pd, p.x, p.y, p.z) : interp_tsdf_only((const float*) a.phi_n, a.pd, p.x, p.y, p.z);
    else ((float2*) a.pnp)[i] = interp_tsdf((const float2*) a.phi_n, a.pd, p.x, p.y, p.z);
    const bool owned = x >= a.own[0] && x < a.own[1] && y >= a.own[2] && y < a.own[3] && z >= a.own[4] && z < a.own[5];
    return owned ? norm_sq4(uu) : 0.f;
}

// The SOFTWARE-PIPELINED march of pass B (compact solver format, one row per lane).  In the plain march a plane's dependent
// chain ends in a memory round trip nothing hides when few workgroups share a CU (multi-GPU tiles, small grids: fewer workgroups
// than the chip has slots for): the eight phi_n corners are gathered at the coordinates the psi update has just produced, and the
// step waits for them.  Here the gather of plane z-1 is ISSUED AT THE TOP of step z, together with the step's other requests
// (psi(z), nabla_U plane z+4, the halo of plane z+1), and everything is awaited once, behind the barrier and the 63 taps:
//     requests | barrier | taps of plane z (LDS + the 7 register planes) | -- await --
//     fold the corners into phi_n o psi(z-1), psi(z) -= alpha * t, store both, shift the z pipeline, stage plane z+1 (centre +
//     halo) into the OTHER LDS buffer.
// No request is in flight across the loop's back edge (the compiler would wait for all of them there anyway, to copy the
// loop-carried registers), one barrier per plane as before (a wave writes buffer b^1 only behind the barrier that followed the
// last reads of b^1).  Same arithmetic, same bits.
template <int WY, int NTL, bool DOWN>
SOBFU_DEV void pass_b_march_pipe(const PassBArgs& a, const TileGeom& tg, const GateRegs& gate, float4 (*tile)[WY + 6][TX + 8], uint32_t* s_max) {
    constexpr int R = 3, TY = WY;
    constexpr int NXH = (2 * R * TY + TX - 1) / TX, NTASK = 2 * R + NXH, TPW = (NTASK + WY - 1) / WY;
    constexpr uint32_t VB = 12u, TB = 4u;
    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    const int u0 = tg.u0, v0 = tg.v0, zb = tg.zb, ze = tg.ze;
    const int u = u0 + lx, uc = min(u, tg.DU - 1), v = v0 + wy;
    const size_t plane = (size_t) d.x * d.y, sv = (size_t) d.x;
    const uint32_t cell = (uint32_t) ((size_t) uc + sv * (size_t) min(v, tg.DV - 1));
    const uint32_t off = cell * VB, offT = cell * TB;
    const bool mine = u < tg.u_hi && v < tg.v_hi;
    const bool owned = u >= a.own[0] && u < a.own[1] && v >= a.own[2] && v < a.own[3];
    int h_lr[TPW], h_lc[TPW];
    uint32_t h_off[TPW];
    bool h_on[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int task = wy + k * WY;
        h_on[k] = task < NTASK;
        int lr = 0, lc = 0;
        if (task < R) { lr = task; lc = lx + R; }
        else if (task < 2 * R) { lr = TY + task; lc = lx + R; }
        else {
            const int e = (task - 2 * R) * TX + lx;
            h_on[k] = h_on[k] && e < 2 * R * TY;
            const int row = e / (2 * R), c = e % (2 * R);
            lr = R + row;
            lc = c < R ? c : TX + c;
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        const int gu = min(max(u0 - R + lc, 0), tg.DU - 1), gv = min(max(v0 - R + lr, 0), tg.DV - 1);
        h_off[k] = (uint32_t) ((size_t) gu + (size_t) gv * sv) * VB;
    }
    // buffer addressing: a resource per array (SGPRs), the lane's 32-bit byte offset in the plane, the plane as a scalar byte offset
    const uint32_t plane4 = (uint32_t) plane * TB, cells = (uint32_t) plane * (uint32_t) d.z;
    const __amdgpu_buffer_rsrc_t r_nu = buf_rsrc(a.nU, cells * VB), r_psi = buf_rsrc(a.psi, cells * VB), r_out = buf_rsrc(a.psi_out, cells * VB),
                                 r_f = buf_rsrc(a.pnp, cells * TB);
    auto nU_plane = [&](int z) { return 3u * (uint32_t) min(max(z, 0), d.z - 1) * plane4; };
    // direct transport: cells of nabla_U's halo rims were stored by other GPUs -> this launch reads nabla_U at system scope (measured on
    // one GPU with every load of the march so marked: pass B 24.7 -> 24.2 us, the same fabric bytes: free)
    const bool sys = a.sys_acquire != 0;
    // The z taps live in EIGHT register slots that rotate (the loop is unrolled eight times: slot indices are constants, nothing is
    // shifted -- the seven-plane shift of the plain march is 18 register moves per plane, 9 % of the loop's vector instructions).
    // Slot m % 8 holds the m-th plane of the march's window: m = st .. st + 6 at step st, i.e. planes z - 3 .. z + 3 going up
    // (plane = z_first - 3 + m) or z + 3 .. z - 3 going DOWN (plane = z_first + 3 - m, see Box::pair); the plane requested at step st,
    // m = st + 7, takes the slot the window left a step ago.
    constexpr int DZ = DOWN ? -1 : 1;
    const int z_first = DOWN ? ze - 1 : zb, n_steps = ze - zb;
    float4 Q[8], hq[TPW];
#pragma unroll
    for (int m = 0; m < 7; ++m) Q[m] = buf_ld3_scope(r_nu, off, nU_plane(z_first + DZ * (m - 3)), sys);
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        if (h_on[k]) hq[k] = buf_ld3_scope(r_nu, h_off[k], nU_plane(z_first), sys);
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;
    tile[0][wy + R][lx + R] = Q[3];
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        if (h_on[k]) tile[0][h_lr[k]][h_lc[k]] = hq[k];
    float4 p_prev = make_float4(0.f, 0.f, 0.f, 0.f);
    float msq = 0.f;
    for (int s0 = 0; s0 < n_steps; s0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int st = s0 + i;
            if (st >= n_steps) break;
            const int z = z_first + DZ * st;
            const int buf = st & 1;
            const uint32_t zcur4 = (uint32_t) z * plane4;
            const float4 qc = Q[(i + 3) % 8];  // plane z
            // this step's requests
            const float4 pv = buf_ld3(r_psi, off, 3u * zcur4, NTL >= 2);
            if (st + 1 < n_steps) {
                Q[(i + 7) % 8] = buf_ld3_scope(r_nu, off, nU_plane(z + 4 * DZ), sys);
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (h_on[k]) hq[k] = buf_ld3_scope(r_nu, h_off[k], nU_plane(z + DZ), sys);
            }
            Gather8 g;
            if (mine && st > 0) g = gather_issue32((const float*) a.phi_n, a.pd, p_prev.x, p_prev.y, p_prev.z);
            __syncthreads();
            // the taps of plane z: x and y from the LDS tile, z from the register planes (sum = 0; ascending j; products not contracted)
            v2f l01 = {0.f, 0.f}, l23 = {0.f, 0.f}, r01 = {0.f, 0.f}, r23 = {0.f, 0.f}, z01 = {0.f, 0.f}, z23 = {0.f, 0.f};
#pragma unroll
            for (int j = -R; j <= R; ++j) {
                const v2f s2 = {a.S.s[R - j], a.S.s[R - j]};
                const float4 vl = (j == 0) ? qc : tile[buf][wy + R][lx + R + j];
                l01 += v2f{vl.x, vl.y} * s2;
                l23 += v2f{vl.z, vl.w} * s2;
                const float4 vr = (j == 0) ? qc : tile[buf][wy + R + j][lx + R];
                r01 += v2f{vr.x, vr.y} * s2;
                r23 += v2f{vr.z, vr.w} * s2;
                const float4 vz = Q[(i + 3 + DZ * j + 8) % 8];  // plane z + j
                z01 += v2f{vz.x, vz.y} * s2;
                z23 += v2f{vz.z, vz.w} * s2;
            }
            const v2f t01 = (l01 + r01) + z01;
            const float tx = t01.x, ty = t01.y, tz = (l23.x + r23.x) + z23.x;
            // update_psi_kernel (solver.cu:64-67)
            const float4 uu = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
            float4 p = pv;
            p.x -= uu.x;
            p.y -= uu.y;
            p.z -= uu.z;
            pin3(p);
            if (mine) {
                if (owned && z >= a.own[4] && z < a.own[5]) msq = fmaxf(msq, norm_sq4(uu));
                if (st > 0) {  // apply_kernel (vector_fields.cu:95-98) of the plane of the step before
                    buf_st1(r_f, offT, DOWN ? zcur4 + plane4 : zcur4 - plane4, gather_finish(g), NTL >= 1);
                }
                buf_st3(r_out, off, 3u * zcur4, p, NTL >= 1);
            }
            p_prev = p;
            if (st + 1 < n_steps) {  // stage the next plane of the march into the other buffer
                tile[buf ^ 1][wy + R][lx + R] = Q[(i + 4) % 8];
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (h_on[k]) tile[buf ^ 1][h_lr[k]][h_lc[k]] = hq[k];
            }
        }
    }
    if (ze > zb && mine) {  // the last plane's warp
        buf_st1(r_f, offT, (uint32_t) (DOWN ? zb : ze - 1) * plane4, interp_tsdf_only32((const float*) a.phi_n, a.pd, p_prev.x, p_prev.y, p_prev.z), NTL >= 1);
    }
    maxnorm_tail<WY>(msq, a.slots, s_max);
}

// DIRECT_OK: the launch may hold direct boxes (multi-GPU tiles)
// NTL: streaming hints (see pass_a_march).  PIPE: the software-pipelined march (pass_b_march_pipe).
// NTBUF (plain march, compact format, arrays below 4 GiB -- the launcher checks): the streaming hint of the 12-byte psi load / store is
// REAL.  hipcc drops the nontemporal flag of __builtin_nontemporal_load / _store on the 4-byte-aligned 12-byte vector type (found in the
// ISA in round 5: `global_load_dwordx3 ... off` without `nt`, while the 4-byte phi_n o psi store carries it); the buffer instructions
// take the hint as an operand.
template <int RPT, int WY, bool WRITE_UPDATES, bool COMPACT, bool DIRECT_OK, bool IDX32 = false, int HL = 0, int NTL = kNT, bool PIPE = false, bool NTBUF = false>
// (the API-format instantiations -- 16-byte psi / nabla_U, 8-byte volumes: the launcher-level entry point and set_compact(0) -- get the
// 128-VGPR budget: at 80 they spilled 12 - 28 B/lane to scratch)
__global__ void __launch_bounds__(TX* WY, (PIPE || !COMPACT) ? SOBFU_MINW_PIPE : SOBFU_MINW_B) fused_smooth_update_apply_kernel(PassBArgs a) {
    static_assert(!NTBUF || (COMPACT && !PIPE && NTL >= 1), "NTBUF: the plain march of the compact format with streaming hints");
    static_assert(!PIPE || (RPT == 1 && COMPACT && IDX32 && !WRITE_UPDATES && HL == 0), "the pipelined march exists for the compact solver format");
    constexpr int R = 3, TY = RPT * WY, LW = TX + 2 * R, LH = TY + 2 * R;
    static_assert(HL == 0 || HL >= 2, "the halo-lead FIFO needs a lead of >= 2 planes (a lead of 1 is the register path, HL = 0)");
    constexpr int NXH = (2 * R * TY + TX - 1) / TX;  // row-tasks for the 2R lane-halo columns
    constexpr int NTASK = 2 * R + NXH, TPW = (NTASK + WY - 1) / WY;
    __shared__ float4 tile[2][LH][LW + 2];
    __shared__ uint32_t s_max[WY];
    __shared__ P3 hfifo[HL > 0 ? HL : 1][HL > 0 ? NTASK * TX : 1];  // 12-byte entries: with the 32 KB tile, 3 workgroups still fit a CU's 160 KB

    // Direct transport (a.sys_acquire): the 4-cell halo rims of nabla_U were stored into this GPU's memory by kernels of OTHER GPUs
    // (write-through at system scope, acknowledged before their arrival flag went out; the flag was seen by this rank's pass A before
    // it retired: DESIGN.md section 6.2).  An invalidate at this kernel's entry (`buffer_inv sc0 sc1` by every wave) was built and
    // measured: + 39 us per launch on a 128^3 tile -- waves start at different times and every late invalidate throws away what the
    // early waves had fetched.  Instead the pipelined march reads nabla_U at system scope on such handles (buf_ld3_scope).
    const GateRegs gate = gate_load(a.prev_slots, a.prev_rows, a.sys_acquire != 0);

    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    // marching workgroups are XCD-swizzled among themselves; direct ones (numbered behind them) keep the dispatch order, which
    // spreads them over all XCDs -- a thin box concentrated on one XCD's 32 CUs is bound by their address units
    const bool marching_wg = (int) blockIdx.x >= a.boxes.m0 && (int) blockIdx.x < a.boxes.m1;
    const unsigned wg = (SOBFU_SWIZZLE_B && marching_wg)
                            ? (unsigned) a.boxes.m0 + xcd_swizzle(blockIdx.x - (unsigned) a.boxes.m0, (unsigned) (a.boxes.m1 - a.boxes.m0)) : blockIdx.x;
    int first_wg, count_wg;
    const Box box = find_box(a.boxes, wg, first_wg, &count_wg);
    if (DIRECT_OK && box.kind != 0) {  // a thin box: one lane per cell
        if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;
        int x, y, z;
        float msq = 0.f;
        const unsigned wd = (SOBFU_BOX_XCD & 2) ? (unsigned) first_wg + box_xcd_order(wg, (unsigned) first_wg, (unsigned) count_wg) : wg;
        if (direct_cell(box, wd, first_wg, x, y, z)) msq = pass_b_direct_cell<WRITE_UPDATES, COMPACT, IDX32>(a, x, y, z);
        maxnorm_tail<WY>(msq, a.slots, s_max);
        return;
    }
    const TileGeom tg = geom_in_box(box, wg, first_wg, d, TY);
    if constexpr (PIPE) {
        if (tg.down) pass_b_march_pipe<WY, NTL, true>(a, tg, gate, tile, s_max);
        else pass_b_march_pipe<WY, NTL, false>(a, tg, gate, tile, s_max);
        return;
    }
    const int u0 = tg.u0, v0 = tg.v0, zb = tg.zb, ze = tg.ze;
    const int u = u0 + lx, uc = min(u, tg.DU - 1);
    const size_t plane = (size_t) d.x * d.y, sv = (size_t) d.x;

    // in-plane BYTE offsets of the lane's cells (a plane of a vector field is < 4 GiB: checked at launch); every plane base is a
    // uniform 64-bit value, so an address costs one scalar pair + one lane register
    constexpr uint32_t VB = COMPACT ? 12u : 16u, TB = COMPACT ? 4u : 8u;  // bytes per cell of a vector field / a TSDF volume
    uint32_t off[RPT], offT[RPT];  // ... in a vector field / in a TSDF volume
    bool mine[RPT];  // the cell is stored by this launch / belongs to this rank (x, y part of the test)
    bool owned[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int v = v0 + wy * RPT + r;
        const uint32_t cell = (uint32_t) ((size_t) uc + sv * (size_t) min(v, tg.DV - 1));
        off[r]      = cell * VB;
        offT[r]     = cell * TB;
        mine[r]     = u < tg.u_hi && v < tg.v_hi;
        owned[r]    = u >= a.own[0] && u < a.own[1] && v >= a.own[2] && v < a.own[3];
    }

    // halo tasks: 0..R-1 rows above, R..2R-1 rows below, then lane-halo cells (2R per tile row)
    int h_lr[TPW], h_lc[TPW];
    uint32_t h_off[TPW];
    bool h_on[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        int task = wy + k * WY;
        h_on[k]  = task < NTASK;
        int lr = 0, lc = 0;
        if (task < R) { lr = task; lc = lx + R; }
        else if (task < 2 * R) { lr = TY + task; lc = lx + R; }  // TY + R + (task - R)
        else {
            int e = (task - 2 * R) * TX + lx;  // 0 .. 2R*TY-1
            h_on[k] = h_on[k] && e < 2 * R * TY;
            int row = e / (2 * R), c = e % (2 * R);
            lr = R + row;
            lc = c < R ? c : TX + c;  // R..2R-1 -> TX+R .. TX+2R-1
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        int gu = min(max(u0 - R + lc, 0), tg.DU - 1), gv = min(max(v0 - R + lr, 0), tg.DV - 1);
        h_off[k] = (uint32_t) ((size_t) gu + (size_t) gv * sv) * VB;
    }

    float4 hq[TPW];
    // Halo cells run HL planes ahead of the plane they are staged for, like the main cells of the z pipeline (which must be 4
    // ahead): a neighbour tile's halo request then meets the owner's own request for the same lines in the L2 instead of coming
    // 3 plane-steps (~5 MB of traffic through a 4 MB L2) later -- 86 of the 111 MB pass B read beyond its minimum at 256^3 were
    // halo lines fetched twice (PMC attribution).  In between a cell waits in a per-lane LDS FIFO (only its own lane
    // ever touches an entry: no barrier involved).  Planes zb+1 .. of the first steps are requested -- and parked -- before the
    // z pipeline's seven planes are, so that their registers are free again by then.
    if (HL > 0) {
        float4 hpre[TPW][HL > 1 ? HL - 1 : 1];
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
#pragma unroll
                for (int p = 1; p < HL; ++p) hpre[k][p - 1] = ldvb<COMPACT>((const char*) a.nU + (size_t) min(zb + p, d.z - 1) * plane * VB, h_off[k]);
            }
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
#pragma unroll
                for (int p = 1; p < HL; ++p) {
                    P3& e = hfifo[(zb + p) % (HL > 0 ? HL : 1)][(wy + k * WY) * TX + lx];
                    e.x = hpre[k][p - 1].x; e.y = hpre[k][p - 1].y; e.z = hpre[k][p - 1].z;
                }
            }
    }
    // z register pipeline q[r][0..6] = planes clamp(z-3 .. z+3)  (clamp-to-edge, solver.cu:396-424)
    float4 q[RPT][7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const char* nUz = (const char*) a.nU + (size_t) min(max(zb - 3 + k, 0), d.z - 1) * plane * VB;
#pragma unroll
        for (int r = 0; r < RPT; ++r) q[r][k] = ldvb<COMPACT>(nUz, off[r]);
    }
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        if (h_on[k]) hq[k] = ldvb<COMPACT>((const char*) a.nU + (size_t) zb * plane * VB, h_off[k]);
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;
    int hslot = HL > 0 ? zb % (HL > 0 ? HL : 1) : 0;  // FIFO slot of plane z

    float msq = 0.f;
    for (int z = zb; z < ze; ++z) {
        const int buf = (z - zb) & 1;
#pragma unroll
        for (int r = 0; r < RPT; ++r) tile[buf][wy * RPT + r + R][lx + R] = q[r][3];
        if (HL > 0) {
        const int hprev = hslot == 0 ? HL - 1 : hslot - 1;  // slot of plane z-1 == slot of plane z-1+HL
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
                const int hi = (wy + k * WY) * TX + lx;
                if (z == zb) {  // plane zb's halo came straight from the prologue's request
                    tile[buf][h_lr[k]][h_lc[k]] = hq[k];
                } else {
                    const P3 e = hfifo[hslot][hi];
                    tile[buf][h_lr[k]][h_lc[k]] = make_float4(e.x, e.y, e.z, 0.f);
                    if (z - 1 + HL < ze) {  // the cell requested during the previous step (plane z-1+HL) takes the slot plane z-1 left
                        P3& w = hfifo[hprev][hi];
                        w.x = hq[k].x; w.y = hq[k].y; w.z = hq[k].z;
                    }
                }
            }
        hslot = hslot + 1 == HL ? 0 : hslot + 1;
        } else {
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) tile[buf][h_lr[k]][h_lc[k]] = hq[k];
        }

        const size_t zcur = (size_t) z * plane;
        float4 pv[RPT], nq[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if constexpr (NTBUF && NTL >= 2) pv[r] = buf_ld3(buf_rsrc(a.psi, (uint32_t) (plane * (size_t) d.z) * VB), off[r], (uint32_t) zcur * VB, true);
            else pv[r] = ldvb<COMPACT>((const char*) a.psi + zcur * VB, off[r], NTL >= 2);
        }
        if (z + 1 < ze) {
            const char* nU4 = (const char*) a.nU + (size_t) min(z + 4, d.z - 1) * plane * VB;
#pragma unroll
            for (int r = 0; r < RPT; ++r) nq[r] = ldvb<COMPACT>(nU4, off[r]);
            if (HL == 0) {
                const char* nU1 = (const char*) a.nU + (size_t) (z + 1) * plane * VB;
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (h_on[k]) hq[k] = ldvb<COMPACT>(nU1, h_off[k]);
            }
        }
        if (HL > 0 && z + HL < ze) {
            const char* nUh = (const char*) a.nU + (size_t) (z + HL) * plane * VB;
#pragma unroll
            for (int k = 0; k < TPW; ++k)
                if (h_on[k]) hq[k] = ldvb<COMPACT>(nUh, h_off[k]);
        }
        __syncthreads();
        // row-axis taps outside this lane's strip
        float4 yt[R], yb[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            yt[j] = tile[buf][wy * RPT + j][lx + R];                 // strip rows -3, -2, -1
            yb[j] = tile[buf][wy * RPT + RPT + R + j][lx + R];       // strip rows RPT, RPT+1, RPT+2
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            // the lane-axis sum (l*) and the row-axis sum (r*) are the x and y convolutions
            // packed fp32 math: the {x, y} and {z, w} halves of a cell are adjacent register pairs (ds_read_b128), so each tap is
            // 2 v_pk_mul_f32 + 2 v_pk_add_f32 instead of 3 + 3 scalar ops (the w lane rides along; products are not contracted)
            v2f l01 = {0.f, 0.f}, l23 = {0.f, 0.f}, r01 = {0.f, 0.f}, r23 = {0.f, 0.f}, z01 = {0.f, 0.f}, z23 = {0.f, 0.f};
#pragma unroll
            for (int j = -R; j <= R; ++j) {
                const v2f s2 = {a.S.s[R - j], a.S.s[R - j]};
                const float4 vl = (j == 0) ? q[r][3] : tile[buf][wy * RPT + r + R][lx + R + j];
                l01 += v2f{vl.x, vl.y} * s2;
                l23 += v2f{vl.z, vl.w} * s2;
                const int rr = r + j;
                const float4 vr = rr < 0 ? yt[rr + R < 0 ? 0 : (rr + R > R - 1 ? R - 1 : rr + R)]
                                         : (rr >= RPT ? yb[rr - RPT > R - 1 ? R - 1 : (rr - RPT < 0 ? 0 : rr - RPT)]
                                                      : q[rr < 0 ? 0 : (rr >= RPT ? RPT - 1 : rr)][3]);
                r01 += v2f{vr.x, vr.y} * s2;
                r23 += v2f{vr.z, vr.w} * s2;
                const float4 vz = q[r][3 + j];
                z01 += v2f{vz.x, vz.y} * s2;
                z23 += v2f{vz.z, vz.w} * s2;
            }
            const v2f t01 = (l01 + r01) + z01;
            float tx = t01.x, ty = t01.y, tz = (l23.x + r23.x) + z23.x;
            // = ((Sx*src) + (Sy*src)) + (Sz*src)  (rows assign, columns +=, depth +=)
            // update_psi_kernel (solver.cu:64-67)
            float4 uu = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
            float4 p  = pv[r];
            p.x -= uu.x;
            p.y -= uu.y;
            p.z -= uu.z;
            pin3(p);
            if (mine[r]) {
                if (owned[r] && z >= a.own[4] && z < a.own[5]) msq = fmaxf(msq, norm_sq4(uu));
                // inside the box no clamp was active: off[r] is the cell itself
                if constexpr (NTBUF) buf_st3(buf_rsrc(a.psi_out, (uint32_t) (plane * (size_t) d.z) * VB), off[r], (uint32_t) zcur * VB, p, true);
                else stvb<COMPACT>((char*) a.psi_out + zcur * VB, off[r], p, NTL >= 1);
                if (WRITE_UPDATES) *(float4*) ((char*) a.updates + zcur * 16 + (size_t) (offT[r] / TB * 16u)) = uu;
                // apply_kernel (vector_fields.cu:95-98)
                if (COMPACT) {
                    const float f = IDX32 ? interp_tsdf_only32((const float*) a.phi_n, a.pd, p.x, p.y, p.z)
                                          : interp_tsdf_only((const float*) a.phi_n, a.pd, p.x, p.y, p.z);
                    float* fo = (float*) ((char*) a.pnp + zcur * TB + (size_t) offT[r]);
                    if (NTL >= 1) __builtin_nontemporal_store(f, fo);
                    else *fo = f;
                }
                else *(float2*) ((char*) a.pnp + zcur * TB + (size_t) offT[r]) = interp_tsdf((const float2*) a.phi_n, a.pd, p.x, p.y, p.z);
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                q[r][k] = q[r][k + 1];
                // keep the shift as plain register moves (hipcc otherwise SLP-vectorises the 7-deep shift of the compact
                // variant into a <7 x float> shuffle that it lowers through 64 B of scratch per lane)
                asm volatile("" : "+v"(q[r][k].x), "+v"(q[r][k].y), "+v"(q[r][k].z));
            }
            q[r][6] = nq[r];
        }
    }
    maxnorm_tail<WY>(msq, a.slots, s_max);  // max ||u||^2 over the voxels this workgroup owns
}

// --- compact-format conversions (once per solve, not per iteration) ----------------------------------------------
__global__ void __launch_bounds__(256) pack_vec_kernel(const float4* __restrict__ src, P3* __restrict__ dst, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    stv<true>(dst, i, src[i]);
}
// writes xyz back into the API float4 field; w is left untouched, as update_psi_kernel leaves it (utils.hpp:260-265)
__global__ void __launch_bounds__(256) unpack_vec_kernel(const P3* __restrict__ src, float4* __restrict__ dst, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 v = ldv<true>(src, i);
    *(v3f_u*) ((float*) (dst + i)) = v3f{v.x, v.y, v.z};
}
__global__ void __launch_bounds__(256) extract_tsdf_kernel(const float2* __restrict__ src, float* __restrict__ dst, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) dst[i] = src[i].x;
}
__global__ void __launch_bounds__(256) apply_tsdf_only_kernel(const float* __restrict__ phi, float* __restrict__ out,
                                                              const P3* __restrict__ psi, Dims d, Dims pd) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z;
    if (x >= d.x || y >= d.y) return;
    size_t i = vidx(d, x, y, z);
    float4 p = ldv<true>(psi, i);
    out[i]   = interp_tsdf_only(phi, pd, p.x, p.y, p.z);
}

// Entering / leaving the compact format in ONE pass each (the solver handle's whole-volume case; the slab loop keeps the
// separate kernels because its phi_n is a different, larger array than its slab fields):
//   enter: psi float4 -> 12-byte psi, tsdf channels of phi_global / phi_n, F = interpolate_tsdf(phi_n, psi).tsdf (solver.cu:106)
//   leave: 12-byte psi -> psi.xyz (w untouched), phi_n o psi = interpolate_tsdf(phi_n, psi) (the state solver.cu:168 leaves)
__global__ void __launch_bounds__(256) compact_enter_kernel(const float4* __restrict__ psi4, const float2* __restrict__ pg2,
                                                            const float2* __restrict__ pn2, P3* __restrict__ c_psi, float* __restrict__ c_g,
                                                            float* __restrict__ c_n, float* __restrict__ c_f, Dims d) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z;
    if (x >= d.x || y >= d.y) return;
    const size_t i = vidx(d, x, y, z);
    const float4 p = psi4[i];
    stv<true>(c_psi, i, p);
    c_g[i] = pg2[i].x;
    c_n[i] = pn2[i].x;
    c_f[i] = interp_tsdf(pn2, d, p.x, p.y, p.z).x;  // same lerp chain on the same tsdf values as interp_tsdf_only on c_n
}
__global__ void __launch_bounds__(256) compact_leave_kernel(const P3* __restrict__ c_psi, const float2* __restrict__ pn2,
                                                            float4* __restrict__ psi4, float2* __restrict__ pnp2, Dims d) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z;
    if (x >= d.x || y >= d.y) return;
    const size_t i = vidx(d, x, y, z);
    const float4 p = ldv<true>(c_psi, i);
    *(v3f_u*) ((float*) (psi4 + i)) = v3f{p.x, p.y, p.z};
    pnp2[i] = interp_tsdf(pn2, d, p.x, p.y, p.z);
}

// --- halo messages of a 3-D tile ------------------------------------------------------------------------------------
// A message is a box of cells of a 12-byte field, laid out x fastest in a contiguous buffer segment.  One launch packs (or
// unpacks) all messages of an exchange: one thread per cell, the message found by a scan of <= 18 prefix entries.
struct MsgBoxes {
    int n;
    int x0[kMaxMsgs], y0[kMaxMsgs], z0[kMaxMsgs], nx[kMaxMsgs], ny[kMaxMsgs];
    unsigned first[kMaxMsgs + 1];  // first cell of message i in the buffer; first[n] = cells in all messages
};
template <bool PACK>
__global__ void __launch_bounds__(256) msg_copy_kernel(float* __restrict__ field3, float* __restrict__ buf, Dims d, MsgBoxes m) {
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= m.first[m.n]) return;
    int x0 = m.x0[0], y0 = m.y0[0], z0 = m.z0[0], nx = m.nx[0], ny = m.ny[0];
    unsigned first = 0;
#pragma unroll
    for (int k = 1; k < kMaxMsgs; ++k)
        if (k < m.n && c >= m.first[k]) {
            x0 = m.x0[k]; y0 = m.y0[k]; z0 = m.z0[k]; nx = m.nx[k]; ny = m.ny[k];
            first = m.first[k];
        }
    const unsigned e = c - first;
    const int ix = (int) (e % (unsigned) nx), iy = (int) ((e / (unsigned) nx) % (unsigned) ny), iz = (int) (e / ((unsigned) nx * (unsigned) ny));
    const size_t i = vidx(d, x0 + ix, y0 + iy, z0 + iz);
    if (PACK) stv<true>(buf, c, ldv<true>(field3, i));
    else stv<true>(field3, i, ldv<true>(buf, c));
}

// The scatter of an exchange's packed messages by a precomputed TABLE: cell c of the receive buffer goes to cell table[c] of the field.
// The loop issues the same scatter every iteration, so the message scan, the three integer divisions per cell and the 500-byte argument
// block of msg_copy_kernel are paid once, at handle creation: what is left is two independent loads and a store per cell.
__global__ void __launch_bounds__(256) msg_scatter_table_kernel(float* __restrict__ field3, const float* __restrict__ buf, const uint32_t* __restrict__ table,
                                                                unsigned n) {
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n) return;
    stv<true>(field3, table[c], ldv<true>(buf, c));
}

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

int sobfu_hip_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_grad, const float* d_L,
                                 float* d_nabla_U, float w_reg, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_phi_n_psi && d_phi_global && d_grad && d_L && d_nabla_U && X > 0 && Y > 0 && Z > 0);
    size_t N = (size_t) X * Y * Z;
    hipLaunchKernelGGL(potential_gradient_kernel, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       (const float2*) d_phi_n_psi, (const float2*) d_phi_global, (const float4*) d_grad, (const float4*) d_L,
                       (float4*) d_nabla_U, w_reg, N);
    return (int) hipGetLastError();
}

#define CONV_IMPL(name, AXIS)                                                                                       \
    int name(float* d_dst, const float* d_src, const float taps[7], int w, int h, int d, void* stream) {            \
        SOBFU_CHECK_ARGS(d_dst && d_src && taps && w > 0 && h > 0 && d > 0 && d_dst != d_src);                       \
        Taps S;                                                                                                     \
        for (int i = 0; i < 7; ++i) S.s[i] = taps[i];                                                               \
        hipLaunchKernelGGL(conv1d_kernel<AXIS>, voxel_grid(w, h, d), voxel_block(), 0, (hipStream_t) stream,         \
                           (float4*) d_dst, (const float4*) d_src, S, Dims{w, h, d});                               \
        return (int) hipGetLastError();                                                                             \
    }
CONV_IMPL(sobfu_hip_convolution_rows, 0)
CONV_IMPL(sobfu_hip_convolution_columns, 1)
CONV_IMPL(sobfu_hip_convolution_depth, 2)

int sobfu_hip_update_psi(float* d_psi, const float* d_nabla_U_S, float* d_updates, float alpha, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_nabla_U_S && d_updates && X > 0 && Y > 0 && Z > 0);
    size_t N = (size_t) X * Y * Z;
    hipLaunchKernelGGL(update_psi_kernel, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       (float4*) d_psi, (const float4*) d_nabla_U_S, (float4*) d_updates, alpha, N);
    return (int) hipGetLastError();
}

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
