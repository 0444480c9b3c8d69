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

constexpr int TX = 64;  // tile width = one wave of consecutive x

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
#ifndef SOBFU_NT
#define SOBFU_NT 3  // 1 = pass B stores of psi / phi_n o psi, 2 = + pass B load of psi, 3 = + pass A load of phi_global (4: + nabla_U store, 5: + pass A inner rows: A slower, B faster, no net gain)
#endif
template <bool C>
SOBFU_DEV float4 ldv_nt(const void* base, size_t i) {
    if (C) {
        v3f v = __builtin_nontemporal_load((const v3f_u*) ((const float*) base + 3 * i));
        return make_float4(v.x, v.y, v.z, 0.f);
    }
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = __builtin_nontemporal_load((const v4f*) base + i);
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool C>
SOBFU_DEV void stv_nt(void* base, size_t i, const float4& v) {
    if (C) {
        v3f o = {v.x, v.y, v.z};
        __builtin_nontemporal_store(o, (v3f_u*) ((float*) base + 3 * i));
    } else {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f o = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(o, (v4f*) base + i);
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

// --- workgroup -> tile map ---------------------------------------------------------------------------------------
// Linear workgroup id -> (x tile fastest, then y, then z-chunk).  With SOBFU_XCD_SWIZZLE the id is first remapped so
// that each XCD (workgroup b runs on XCD b % 8 -- observed, used for speed only) owns a contiguous run of tiles and
// serves neighbour-tile halos from its own L2.  PMC (256^3): fabric bytes per launch drop 1.013 -> 0.821 GB for pass
// A and 1.406 -> 1.286 GB for pass B; interleaved A/B wall time: pass A -3 %, pass B +1 % (not fabric-bound) -> the
// map is enabled for pass A only.
struct TileId {
    int tx, ty, tz;
};
template <bool XCD_SWIZZLE>
SOBFU_DEV TileId tile_of_block(int ntx, int nty, int ntz) {
    unsigned t = blockIdx.x;
    if (XCD_SWIZZLE) {
        const unsigned nb = (unsigned) ntx * nty * ntz, q = nb / 8u, rem = nb % 8u, xcd = t % 8u, slot = t / 8u;
        t = xcd * q + min(xcd, rem) + slot;  // bijective for any nb
    }
    TileId r;
    r.tx = (int) (t % ntx);
    r.ty = (int) ((t / ntx) % nty);
    r.tz = (int) (t / ((unsigned) ntx * nty));
    return r;
}

// z-chunk -> plane range of a launch that covers [z_lo, z_hi) and, optionally, [z_lo2, z_hi2)
SOBFU_DEV int z_chunks(int z_lo, int z_hi, int zc) { return z_hi > z_lo ? (z_hi - z_lo + zc - 1) / zc : 0; }
SOBFU_DEV void chunk_range(int tz, int zc, int z_lo, int z_hi, int z_lo2, int z_hi2, int& zb, int& ze) {
    const int n1 = z_chunks(z_lo, z_hi, zc);
    const bool second = tz >= n1;
    zb = second ? z_lo2 + (tz - n1) * zc : z_lo + tz * zc;
    ze = min(zb + zc, second ? z_hi2 : z_hi);
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
SOBFU_DEV GateRegs gate_load(const uint32_t* __restrict__ prev_slots, int prev_rows) {
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
                for (int k = 0; k < 4; ++k) g.v[4 * r + k] = row[tid + 64 * k];
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
struct PassAArgs {
    const void* pnp;  // phi_n o psi
    const void* pg;   // phi_global
    const void* psi;
    void* nU;
    Dims d;
    float w_reg;
    int zc;  // slices per workgroup
    int z_lo, z_hi;  // planes [z_lo, z_hi) are produced by this launch (the whole volume, or a sub-range of a slab)
    int z_lo2, z_hi2;  // optional second range (both boundary regions of a slab in ONE launch); empty when z_hi2 <= z_lo2
    const uint32_t* prev_slots;
    float max_update_norm;
};

template <int RPT, int WY, bool COMPACT>
__global__ void __launch_bounds__(TX* WY) fused_potential_gradient_kernel(PassAArgs a) {
    constexpr int TY = RPT * WY, LW = TX + 2, LH = TY + 2;
    constexpr int NXH = (2 * TY + 63) / 64;  // wave-tasks for the two x-halo columns
    constexpr int NTASK = 2 + NXH, TPW = (NTASK + WY - 1) / WY;
    __shared__ float4 t_psi[2][LH][LW + 2];  // {psi.xyz, F = (phi_n o psi).tsdf} -- psi.w is never read

    const GateRegs gate = gate_load(a.prev_slots, 1);

    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    const TileId tid3 = tile_of_block<true>((d.x + TX - 1) / TX, (d.y + TY - 1) / TY, z_chunks(a.z_lo, a.z_hi, a.zc) + z_chunks(a.z_lo2, a.z_hi2, a.zc));
    const int x0 = tid3.tx * TX, y0 = tid3.ty * TY;
    int zb, ze;
    chunk_range(tid3.tz, a.zc, a.z_lo, a.z_hi, a.z_lo2, a.z_hi2, zb, ze);
    const int x = x0 + lx, xc = min(x, d.x - 1);
    const size_t plane = (size_t) d.x * d.y;

    int yr[RPT];  // clamped global rows of this lane's strip
    size_t off[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        yr[r]  = min(y0 + wy * RPT + r, d.y - 1);
        off[r] = (size_t) xc + (size_t) d.x * yr[r];
    }
    // halo tasks: task 0 = row above the tile, task 1 = row below, tasks 2.. = x-halo cells (col -1 / col TX)
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
            int e = (task - 2) * 64 + lx;  // 0 .. 2*TY-1
            h_on[k] = h_on[k] && e < 2 * TY;
            lr = 1 + (e >> 1);
            lc = (e & 1) ? LW - 1 : 0;
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        int gx = min(max(x0 - 1 + lc, 0), d.x - 1), gy = min(max(y0 - 1 + lr, 0), d.y - 1);
        h_off[k] = (size_t) gx + (size_t) d.x * gy;
    }

    // z register pipeline: m = z-1, c = z, n = z+1 (clamped loads; boundary rules applied at use)
    float4 pm[RPT], pc[RPT], pn[RPT];
    float fm[RPT], fc[RPT], fn[RPT];
    float4 hp[TPW];
    float hf[TPW];
    {
        const size_t zm = (size_t) max(zb - 1, 0) * plane, zc0 = (size_t) zb * plane;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pm[r] = ldv<COMPACT>(a.psi, zm + off[r]);
            fm[r] = ldt<COMPACT>(a.pnp, zm + off[r]);
            pc[r] = ldv<COMPACT>(a.psi, zc0 + off[r]);
            fc[r] = ldt<COMPACT>(a.pnp, zc0 + off[r]);
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
                hp[k] = ldv<COMPACT>(a.psi, zc0 + h_off[k]);
                hf[k] = ldt<COMPACT>(a.pnp, zc0 + h_off[k]);
            }
    }
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;

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
        float bg[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (SOBFU_NT >= 5 && COMPACT && RPT == 1 && wy > 0 && wy < WY - 1) {  // rows no y-neighbour tile re-reads
                pn[r] = ldv_nt<COMPACT>(a.psi, zn + off[r]);
                fn[r] = __builtin_nontemporal_load((const float*) a.pnp + zn + off[r]);
            } else {
                pn[r] = ldv<COMPACT>(a.psi, zn + off[r]);
                fn[r] = ldt<COMPACT>(a.pnp, zn + off[r]);
            }
            bg[r] = (SOBFU_NT >= 3 && COMPACT) ? __builtin_nontemporal_load((const float*) a.pg + zcur + off[r]) : ldt<COMPACT>(a.pg, zcur + off[r]);
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
        const bool xlo = (x == 0), xhi = (x == d.x - 1);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int y = y0 + wy * RPT + r;
            const int lr = wy * RPT + r + 1;
            const bool ylo = (y == 0), yhi = (y == d.y - 1);
            // raw neighbours
            float4 pxp = t_psi[buf][lr][lx + 2], pxm = t_psi[buf][lr][lx];
            float fxp = pxp.w, fxm = pxm.w;
            float4 pyp, pym;
            float fyp, fym;
            if (r + 1 < RPT) { pyp = pc[r + 1 < RPT ? r + 1 : r]; fyp = fc[r + 1 < RPT ? r + 1 : r]; }
            else { pyp = t_psi[buf][lr + 1][lx + 1]; fyp = pyp.w; }
            if (r > 0) { pym = pc[r > 0 ? r - 1 : r]; fym = fc[r > 0 ? r - 1 : r]; }
            else { pym = t_psi[buf][lr - 1][lx + 1]; fym = pym.w; }
            float4 pzp = pn[r], pzm = pm[r];
            float fzp = fn[r], fzm = fm[r];
            const float4 c = pc[r];
            // TsdfDifferentiator boundary rule (vector_fields.cu:165-191): mirror the missing neighbour
            float gx1 = xhi ? fxm : fxp, gx2 = xlo ? fxp : fxm;
            float gy1 = yhi ? fym : fyp, gy2 = ylo ? fyp : fym;
            float gz1 = zhi ? fzm : fzp, gz2 = zlo ? fzp : fzm;
            float4 g = f4((gx1 - gx2) / 2.f, (gy1 - gy2) / 2.f, (gz1 - gz2) / 2.f);
            // SecondOrderDifferentiator boundary rule (vector_fields.cu:299-331): both neighbours <- centre
            if (xlo || xhi) { pxp = c; pxm = c; }
            if (ylo || yhi) { pyp = c; pym = c; }
            if (zlo || zhi) { pzp = c; pzm = c; }
            float4 v = mul4(c, -6.f);
            v = add4(v, pxp);
            v = add4(v, pxm);
            v = add4(v, pyp);
            v = add4(v, pym);
            v = add4(v, pzp);
            v = add4(v, pzm);
            float4 L = mul4(v, -1.f);
            // calculate_potential_gradient_kernel (solver.cu:28-31)
            float diff = fc[r] - bg[r];
            float4 o   = add4(mul4(g, diff), mul4(L, a.w_reg));
            if (x < d.x && y < d.y) {
                if (SOBFU_NT >= 4) stv_nt<COMPACT>(a.nU, zcur + (size_t) x + (size_t) d.x * y, o);
                else stv<COMPACT>(a.nU, zcur + (size_t) x + (size_t) d.x * y, o);
            }
        }
        // shift the z pipeline
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pm[r] = pc[r];
            pc[r] = pn[r];
            fm[r] = fc[r];
            fc[r] = fn[r];
        }
    }
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
    int zc;
    int z_lo, z_hi;  // planes produced by this launch
    int z_lo2, z_hi2;  // optional second range (see PassAArgs)
    const uint32_t* prev_slots;
    float max_update_norm;
    // multi-GPU slab tiles: the fields are local slabs (d) that carry halo planes, phi_n is the whole volume (pd);
    // only planes [own_lo, own_hi) are owned by this rank and enter the max-norm.  Single GPU: pd == d, [0, d.z).
    Dims pd;
    int own_lo, own_hi;
    int prev_rows;  // rows the gate looks at (see solver_converged)
    void* psi_out;  // where the updated psi goes: == psi (in place) or the other half of a ping-pong pair (native tiled loop)
};

#ifndef SOBFU_MINW_B
#define SOBFU_MINW_B 6  // waves/SIMD the register allocator must leave room for: <= 80 VGPR -> 3 workgroups of 8 waves per CU
#endif
template <int RPT, int WY, bool WRITE_UPDATES, bool COMPACT>
__global__ void __launch_bounds__(TX* WY, SOBFU_MINW_B) fused_smooth_update_apply_kernel(PassBArgs a) {
    constexpr int R = 3, TY = RPT * WY, LW = TX + 2 * R, LH = TY + 2 * R;
    constexpr int NXH = (2 * R * TY + 63) / 64;  // wave-tasks for the 2R x-halo columns
    constexpr int NTASK = 2 * R + NXH, TPW = (NTASK + WY - 1) / WY;
    __shared__ float4 tile[2][LH][LW + 2];
    __shared__ uint32_t s_max[WY];

    const GateRegs gate = gate_load(a.prev_slots, a.prev_rows);

    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    const TileId tid3 = tile_of_block<SOBFU_SWIZZLE_B>((d.x + TX - 1) / TX, (d.y + TY - 1) / TY, z_chunks(a.z_lo, a.z_hi, a.zc) + z_chunks(a.z_lo2, a.z_hi2, a.zc));
    const int x0 = tid3.tx * TX, y0 = tid3.ty * TY;
    int zb, ze;
    chunk_range(tid3.tz, a.zc, a.z_lo, a.z_hi, a.z_lo2, a.z_hi2, zb, ze);
    const int x = x0 + lx, xc = min(x, d.x - 1);
    const size_t plane = (size_t) d.x * d.y;

    size_t off[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) off[r] = (size_t) xc + (size_t) d.x * min(y0 + wy * RPT + r, d.y - 1);

    // halo tasks: 0..R-1 rows above, R..2R-1 rows below, then x-halo cells (2R per tile row)
    int h_lr[TPW], h_lc[TPW];
    size_t h_off[TPW];
    bool h_on[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        int task = wy + k * WY;
        h_on[k]  = task < NTASK;
        int lr = 0, lc = 0;
        if (task < R) { lr = task; lc = lx + R; }
        else if (task < 2 * R) { lr = TY + task; lc = lx + R; }  // TY + R + (task - R)
        else {
            int e = (task - 2 * R) * 64 + lx;  // 0 .. 2R*TY-1
            h_on[k] = h_on[k] && e < 2 * R * TY;
            int row = e / (2 * R), c = e % (2 * R);
            lr = R + row;
            lc = c < R ? c : TX + c;  // R..2R-1 -> TX+R .. TX+2R-1
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        int gx = min(max(x0 - R + lc, 0), d.x - 1), gy = min(max(y0 - R + lr, 0), d.y - 1);
        h_off[k] = (size_t) gx + (size_t) d.x * gy;
    }

    // z register pipeline q[r][0..6] = planes clamp(z-3 .. z+3)  (clamp-to-edge, solver.cu:396-424)
    float4 q[RPT][7];
    float4 hq[TPW];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const size_t zo = (size_t) min(max(zb - 3 + k, 0), d.z - 1) * plane;
#pragma unroll
        for (int r = 0; r < RPT; ++r) q[r][k] = ldv<COMPACT>(a.nU, zo + off[r]);
    }
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        if (h_on[k]) hq[k] = ldv<COMPACT>(a.nU, (size_t) zb * plane + h_off[k]);
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;

    float msq = 0.f;
    for (int z = zb; z < ze; ++z) {
        const int buf = (z - zb) & 1;
#pragma unroll
        for (int r = 0; r < RPT; ++r) tile[buf][wy * RPT + r + R][lx + R] = q[r][3];
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) tile[buf][h_lr[k]][h_lc[k]] = hq[k];

        const size_t zcur = (size_t) z * plane;
        float4 pv[RPT], nq[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) pv[r] = SOBFU_NT >= 2 ? ldv_nt<COMPACT>(a.psi, zcur + off[r]) : ldv<COMPACT>(a.psi, zcur + off[r]);
        if (z + 1 < ze) {
            const size_t z4 = (size_t) min(z + 4, d.z - 1) * plane, z1 = (size_t) (z + 1) * plane;
#pragma unroll
            for (int r = 0; r < RPT; ++r) nq[r] = ldv<COMPACT>(a.nU, z4 + off[r]);
#pragma unroll
            for (int k = 0; k < TPW; ++k)
                if (h_on[k]) hq[k] = ldv<COMPACT>(a.nU, z1 + h_off[k]);
        }
        __syncthreads();

        // y taps outside this lane's strip
        float4 yt[R], yb[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            yt[j] = tile[buf][wy * RPT + j][lx + R];                 // strip rows -3, -2, -1
            yb[j] = tile[buf][wy * RPT + RPT + R + j][lx + R];       // strip rows RPT, RPT+1, RPT+2
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int y = y0 + wy * RPT + r;
            float sxx = 0.f, sxy = 0.f, sxz = 0.f, syx = 0.f, syy = 0.f, syz = 0.f, szx = 0.f, szy = 0.f, szz = 0.f;
#pragma unroll
            for (int j = -R; j <= R; ++j) {
                const float s = a.S.s[R - j];
                float4 vx = (j == 0) ? q[r][3] : tile[buf][wy * RPT + r + R][lx + R + j];
                sxx += vx.x * s;
                sxy += vx.y * s;
                sxz += vx.z * s;
                const int rr = r + j;
                float4 vy = rr < 0 ? yt[rr + R < 0 ? 0 : (rr + R > R - 1 ? R - 1 : rr + R)]
                                   : (rr >= RPT ? yb[rr - RPT > R - 1 ? R - 1 : (rr - RPT < 0 ? 0 : rr - RPT)]
                                                : q[rr < 0 ? 0 : (rr >= RPT ? RPT - 1 : rr)][3]);
                syx += vy.x * s;
                syy += vy.y * s;
                syz += vy.z * s;
                float4 vz = q[r][3 + j];
                szx += vz.x * s;
                szy += vz.y * s;
                szz += vz.z * s;
            }
            // ((Sx*src) + (Sy*src)) + (Sz*src)  (rows assign, columns +=, depth +=)  -- explicit v_pk_mul/add pairing of the
            // taps was tried and is not faster (pass B 167 us either way), so the loop stays scalar
            float tx = (sxx + syx) + szx, ty = (sxy + syy) + szy, tz = (sxz + syz) + szz;
            // update_psi_kernel (solver.cu:64-67)
            float4 u = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
            float4 p = pv[r];
            p.x -= u.x;
            p.y -= u.y;
            p.z -= u.z;
            if (x < d.x && y < d.y) {
                if (z >= a.own_lo && z < a.own_hi) msq = fmaxf(msq, norm_sq4(u));
                const size_t i = zcur + (size_t) x + (size_t) d.x * y;
                if (SOBFU_NT >= 1) stv_nt<COMPACT>(a.psi_out, i, p);
                else stv<COMPACT>(a.psi_out, i, p);
                if (WRITE_UPDATES) a.updates[i] = u;
                // apply_kernel (vector_fields.cu:95-98)
                if (COMPACT && SOBFU_NT >= 1) __builtin_nontemporal_store(interp_tsdf_only((const float*) a.phi_n, a.pd, p.x, p.y, p.z), (float*) a.pnp + i);
                else if (COMPACT) ((float*) a.pnp)[i] = interp_tsdf_only((const float*) a.phi_n, a.pd, p.x, p.y, p.z);
                else ((float2*) a.pnp)[i] = interp_tsdf((const float2*) a.phi_n, a.pd, p.x, p.y, p.z);
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

    // max ||u||^2 over the voxels this workgroup owns
    uint32_t m = __float_as_uint(msq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
    if (lx == 0) s_max[wy] = m;
    __syncthreads();
    if (lx == 0 && wy == 0) {
#pragma unroll
        for (int w = 1; w < WY; ++w) m = max(m, s_max[w]);
        atomicMax(a.slots + (blockIdx.x & 255u), m);
    }
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

}  // namespace

// Tile configuration of the fused passes (see DESIGN.md "Kernel tuning").
#ifndef SOBFU_RPT
#define SOBFU_RPT 1
#endif
#ifndef SOBFU_WY
#define SOBFU_WY 8
#endif

namespace sobfu_hip {

// z-chunk length of a fused pass.  A launch has tiles * ceil(nz / zc) workgroups; `capacity` of them are resident at
// once on the chip (256 CUs x workgroups per CU allowed by VGPRs / LDS / waves).  Cost model: time ~ (1 + refill / zc)
// / utilisation, where utilisation = groups / (ceil(groups / capacity) * capacity) penalises a ragged last wave of
// workgroups (measured at 256^3, pass B: 768 groups = exactly 3 per CU: 166 us; 512 groups: 184 us; 1024: 195 us) and
// refill = planes re-read when a march starts (2 for pass A, 6 for pass B).  Small grids end up with many short
// marches, which is what the latency-bound regime wants (64^3: zc = 2 is 1.4x faster than zc = 8).
int pick_zc(int X, int Y, int nz, int ty, int capacity, int refill, const char* env) {
    if (const char* e = getenv(env)) {  // tuning override
        int v = atoi(e);
        if (v > 0) return v < nz ? v : nz;
    }
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

int launch_pass_a(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z,
                  const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact, int z_lo, int z_hi,
                  int z_lo2, int z_hi2) {
    constexpr int TY = SOBFU_RPT * SOBFU_WY;
    if (z_hi <= 0 && z_hi2 <= z_lo2) { z_lo = 0; z_hi = Z; }  // no range given: the whole grid
    if (z_hi <= z_lo) { z_lo = z_lo2; z_hi = z_hi2; z_lo2 = z_hi2 = 0; }
    if (z_hi <= z_lo) return 0;
    const bool two = z_hi2 > z_lo2;
    const int nz = std::max(z_hi - z_lo, two ? z_hi2 - z_lo2 : 0);  // the longer range sets the march length
    if (zc <= 0) zc = pick_zc(X, Y, nz, TY, (256 * 4) / (two ? 2 : 1), 2, "SOBFU_ZC_A");  // <= 52 VGPR, 22 KB LDS: 4 workgroups of 8 waves per CU
    PassAArgs a{pnp, pg, psi, nU, {X, Y, Z}, w_reg, zc, z_lo, z_hi, z_lo2, z_hi2, prev_slots, max_update_norm};
    const int nchunks = (z_hi - z_lo + zc - 1) / zc + (two ? (z_hi2 - z_lo2 + zc - 1) / zc : 0);
    dim3 grid(((X + TX - 1) / TX) * ((Y + TY - 1) / TY) * nchunks);
    if (compact) hipLaunchKernelGGL((fused_potential_gradient_kernel<SOBFU_RPT, SOBFU_WY, true>), grid, dim3(TX, SOBFU_WY), 0, stream, a);
    else hipLaunchKernelGGL((fused_potential_gradient_kernel<SOBFU_RPT, SOBFU_WY, false>), grid, dim3(TX, SOBFU_WY), 0, stream, a);
    return (int) hipGetLastError();
}

int launch_pass_b(const float* nU, float* psi, const float* phi_n, float* pnp, float* updates, uint32_t* slots,
                  const float taps[7], float alpha, int X, int Y, int Z, const uint32_t* prev_slots,
                  float max_update_norm, int zc, hipStream_t stream, int phi_Z, int own_lo, int own_hi, bool compact, int z_lo,
                  int z_hi, int z_lo2, int z_hi2, float* psi_out, int prev_rows) {
    if (phi_Z <= 0) { phi_Z = Z; own_lo = 0; own_hi = Z; }
    constexpr int TY = SOBFU_RPT * SOBFU_WY;
    if (z_hi <= 0 && z_hi2 <= z_lo2) { z_lo = 0; z_hi = Z; }  // no range given: the whole grid
    if (z_hi <= z_lo) { z_lo = z_lo2; z_hi = z_hi2; z_lo2 = z_hi2 = 0; }
    if (z_hi <= z_lo) return 0;
    const bool two = z_hi2 > z_lo2;
    const int nz = std::max(z_hi - z_lo, two ? z_hi2 - z_lo2 : 0);
    if (zc <= 0) zc = pick_zc(X, Y, nz, TY, (256 * 3) / (two ? 2 : 1), 6, "SOBFU_ZC_B");  // <= 80 VGPR (launch bounds), 32 KB LDS: 3 per CU
    PassBArgs a{nU, psi, phi_n, pnp, (float4*) updates, slots, {X, Y, Z}, {}, alpha, zc, z_lo, z_hi, z_lo2, z_hi2, prev_slots, max_update_norm, {X, Y, phi_Z}, own_lo, own_hi, prev_rows, psi_out ? psi_out : psi};
    for (int i = 0; i < 7; ++i) a.S.s[i] = taps[i];
    const int nchunks = (z_hi - z_lo + zc - 1) / zc + (two ? (z_hi2 - z_lo2 + zc - 1) / zc : 0);
    dim3 grid(((X + TX - 1) / TX) * ((Y + TY - 1) / TY) * nchunks);
#define SOBFU_LAUNCH_B(UPD, CMP) \
    hipLaunchKernelGGL((fused_smooth_update_apply_kernel<SOBFU_RPT, SOBFU_WY, UPD, CMP>), grid, dim3(TX, SOBFU_WY), 0, stream, a)
    if (updates && compact) SOBFU_LAUNCH_B(true, true);
    else if (updates) SOBFU_LAUNCH_B(true, false);
    else if (compact) SOBFU_LAUNCH_B(false, true);
    else SOBFU_LAUNCH_B(false, false);
#undef SOBFU_LAUNCH_B
    return (int) hipGetLastError();
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
int launch_apply_tsdf_only(const float* phi1, float* out1, const float* psi3, int X, int Y, int Z, hipStream_t stream, int phi_Z) {
    hipLaunchKernelGGL(apply_tsdf_only_kernel, voxel_grid(X, Y, Z), voxel_block(), 0, stream, phi1, out1, (const P3*) psi3, Dims{X, Y, Z},
                       Dims{X, Y, phi_Z > 0 ? phi_Z : Z});
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

int sobfu_hip_tile_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_psi, float* d_nabla_U,
                                      float w_reg, int X, int Y, int Lz, int z_begin, int z_end, const uint32_t* d_prev_slots,
                                      float max_update_norm, int compact, void* stream) {
    SOBFU_CHECK_ARGS(d_phi_n_psi && d_phi_global && d_psi && d_nabla_U && X > 1 && Y > 1 && Lz > 1 && z_begin >= 0 && z_begin <= z_end &&
                     z_end <= Lz);
    if (z_begin == z_end) return 0;
    if ((size_t) X * Y * Lz > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;
    return sobfu_hip::launch_pass_a(d_phi_n_psi, d_phi_global, d_psi, d_nabla_U, w_reg, X, Y, Lz, d_prev_slots, max_update_norm, 0,
                                    (hipStream_t) stream, compact != 0, z_begin, z_end);
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
int sobfu_hip_tile_apply_tsdf_only(const float* d_phi1, int Zg, float* d_out1, const float* d_psi3, int X, int Y, int Lz, void* stream) {
    SOBFU_CHECK_ARGS(d_phi1 && d_out1 && d_psi3 && X > 0 && Y > 0 && Lz > 0 && Zg > 0);
    return sobfu_hip::launch_apply_tsdf_only(d_phi1, d_out1, d_psi3, X, Y, Lz, (hipStream_t) stream, Zg);
}

int sobfu_hip_tile_smooth_update_apply(const float* d_nabla_U, float* d_psi, const float* d_phi_n, float* d_phi_n_psi,
                                       float* d_updates, uint32_t* d_max_sq_slots, const float taps[7], float alpha, int X, int Y,
                                       int Lz, int Zg, int z_own_lo, int z_own_hi, int z_begin, int z_end,
                                       const uint32_t* d_prev_slots, float max_update_norm, int compact, void* stream) {
    SOBFU_CHECK_ARGS(d_nabla_U && d_psi && d_phi_n && d_phi_n_psi && d_max_sq_slots && taps && X > 0 && Y > 0 && Lz > 0 && Zg >= 1 &&
                     z_own_lo >= 0 && z_own_lo <= z_own_hi && z_own_hi <= Lz && z_begin >= 0 && z_begin <= z_end && z_end <= Lz);
    if (z_begin == z_end) return 0;
    if ((size_t) X * Y * Lz > (size_t) 0x7fffffff || (size_t) X * Y * Zg > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;
    return sobfu_hip::launch_pass_b(d_nabla_U, d_psi, d_phi_n, d_phi_n_psi, d_updates, d_max_sq_slots, taps, alpha, X, Y, Lz,
                                    d_prev_slots, max_update_norm, 0, (hipStream_t) stream, Zg, z_own_lo, z_own_hi, compact != 0, z_begin, z_end);
}

}  // extern "C"
