// Vector-field kernels for gfx950: identity init, warp (apply), inverse fixed point, TSDF gradient,
// negative Laplacian, Jacobian.  Reference behaviour: src/sobfu/cuda/vector_fields.cu.
// One lane per voxel, wave = 64 consecutive x (coalesced float4 = 1 KiB per wave access), 3-D grid.
#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"

using namespace sobfu_hip;

namespace {

#define VOXEL_XYZ(d)                                                         \
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z; \
    if (x >= (d).x || y >= (d).y) return;
// the same under the XCD-aware tile map (sobfu_device.hpp) -- for the stencils, whose work per voxel is uniform.  NOT for the warps and the
// inverse: their cost follows the data (fixed-point sweeps until the value repeats), a band of tile rows per XCD then leaves the XCDs that
// own the middle of the volume with most of the work (inverse + warp at 256^3: 630 -> 800 us, measured in round 6).
#define VOXEL_XYZ_XCD(d)                                                     \
    int bx_, by_, z;                                                         \
    xcd_tile(bx_, by_, z);                                                   \
    const int x = bx_ * kBX + threadIdx.x, y = by_ * kBY + threadIdx.y;      \
    if (x >= (d).x || y >= (d).y) return;

// init_identity_kernel -- vector_fields.cu:64-79.  (float) z equals the reference's z-fold sum of 1.f exactly.
// base: global coordinates of local cell (0, 0, 0) (multi-GPU tiles; 0 on a single GPU)
__global__ void __launch_bounds__(256) init_identity_kernel(float4* __restrict__ psi, Dims d, Dims base) {
    VOXEL_XYZ(d);
    psi[vidx(d, x, y, z)] = f4((float) (x + base.x), (float) (y + base.y), (float) (z + base.z));
}

// apply_kernel -- vector_fields.cu:81-100
// pd: dims of phi (the whole volume); d: dims of psi / out (== pd on a single GPU, a z-slab on multiple GPUs)
template <bool NT>
__global__ void __launch_bounds__(256) apply_kernel(const float2* __restrict__ phi, float2* __restrict__ out,
                                                    const float4* __restrict__ psi, Dims d, Dims pd) {
    VOXEL_XYZ(d);
    size_t i = vidx(d, x, y, z);
    float4 p = ld4<NT>(&psi[i]);
    st2<NT>(&out[i], interp_tsdf(phi, pd, p.x, p.y, p.z));
}

// estimate_inverse_kernel x n_sweeps -- vector_fields.cu:111-138.  A sweep reads psi (never written) and the
// voxel's OWN psi_inv value only, so the reference's 48 launches collapse into one kernel that iterates the
// fixed point p <- x - u(p) in registers: psi_inv is read once and written once instead of 48 times, bit-identical result.
// The float iteration soon repeats itself: once a sweep returns its input bit for bit every later sweep does too, and once
// it returns the value before that (period 2) the tail just alternates -- in both cases the value after n_sweeps is known
// and the lane stops (same bits as running all sweeps; typically < 10 of the 48 are needed).
// From one sweep to the next p moves by a fraction of a cell, so the eight corners it interpolates between are almost always the eight of the
// sweep before: they stay in registers (as displacements, psi - id of the corner cell: what interpolate_field_inv reads) and are re-loaded only
// when p enters another cell -- a sweep then costs arithmetic instead of eight dependent gathers.  Same values, same lerp chain, same bits.
// disp(x, y, z): the displacement psi - id of a cell (whole volume: disp_at; a tile's window: disp_at_win)
template <class Disp>
SOBFU_DEV float4 inverse_fixed_point(Disp disp, const Dims& pd, float4 v, const float4& id, int n_sweeps) {
    float4 w = v;  // p_it = v, p_(it-1) = w
    auto same = [](const float4& a, const float4& b) {
        return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
               __float_as_uint(a.z) == __float_as_uint(b.z);
    };
    int cx = -1, cy = -1, cz = -1, hx = -1, hy = -1, hz = -1;  // the cell whose corners are held
    float4 hhh, hhg, hgh, hgg, ghh, ghg, ggh, ggg;             // (x, y, z) corner: h = upper index, g = lower (interp_disp's order)
    for (int it = 0; it < n_sweeps; ++it) {
        const Tri a = tri_setup(v.x, pd.x), b = tri_setup(v.y, pd.y), c = tri_setup(v.z, pd.z);
        if (a.g != cx || b.g != cy || c.g != cz || a.h != hx || b.h != hy || c.h != hz) {
            cx = a.g, cy = b.g, cz = c.g, hx = a.h, hy = b.h, hz = c.h;
            hhh = disp(a.h, b.h, c.h), hhg = disp(a.h, b.h, c.g), hgh = disp(a.h, b.g, c.h), hgg = disp(a.h, b.g, c.g);
            ghh = disp(a.g, b.h, c.h), ghg = disp(a.g, b.h, c.g), ggh = disp(a.g, b.g, c.h), ggg = disp(a.g, b.g, c.g);
        }
        const float4 u = lerp4(lerp4(lerp4(hhh, hhg, c.t), lerp4(hgh, hgg, c.t), b.t), lerp4(lerp4(ghh, ghg, c.t), lerp4(ggh, ggg, c.t), b.t), a.t);  // interp_disp
        const float4 nv = sub4(id, mul4(u, 1.f));  // p_(it+1)
        const int left  = n_sweeps - (it + 1);     // sweeps still to run after this one
        if (same(nv, v)) {                         // fixed point
            v = nv;
            break;
        }
        if (it > 0 && same(nv, w)) {               // period 2: ..., w, v, w (= nv), v, w, ...
            // (component by component: a select between two float4 OBJECTS is compiled to a load through a selected stack address -- 48 B of scratch
            //  per lane and a scratch store per sweep, found in the ISA in round 6)
            const bool keep = (left & 1) != 0;
            v = f4(keep ? v.x : nv.x, keep ? v.y : nv.y, keep ? v.z : nv.z);
            break;
        }
        w = v;
        v = nv;
    }
    return v;
}

template <bool NT>
__global__ void __launch_bounds__(256) inverse_fixed_point_kernel(const float4* __restrict__ psi, float4* __restrict__ psi_inv,
                                                                  Dims d, Dims pd, Dims base, int n_sweeps) {
    VOXEL_XYZ(d);
    size_t i = vidx(d, x, y, z);
    const float4 id = f4((float) (x + base.x), (float) (y + base.y), (float) (z + base.z));
    st4<NT>(&psi_inv[i], inverse_fixed_point([&](int a, int b, int c) { return disp_at(psi, pd, a, b, c); }, pd, ld4<NT>(&psi_inv[i]), id, n_sweeps));
}

// The tail of Solver::estimate_psi in one pass (solver.cu:196-199): psi^-1 <- identity, 48 sweeps, phi_global o psi^-1.
// The lane still holds psi^-1(x) when it warps phi_global there: no identity field is written and read back, psi^-1 is not
// re-read by the warp.
template <bool NT>
__global__ void __launch_bounds__(256) inverse_from_identity_and_warp_kernel(const float4* __restrict__ psi, float4* __restrict__ psi_inv,
                                                                             const float2* __restrict__ phi, float2* __restrict__ phi_warped,
                                                                             Dims d, int n_sweeps) {
    VOXEL_XYZ(d);
    size_t i = vidx(d, x, y, z);
    const float4 id = f4((float) x, (float) y, (float) z);
    const float4 v  = inverse_fixed_point([&](int a, int b, int c) { return disp_at(psi, d, a, b, c); }, d, id, id, n_sweeps);
    st4<NT>(&psi_inv[i], v);
    st2<NT>(&phi_warped[i], interp_tsdf(phi, d, v.x, v.y, v.z));
}

// ---- the per-frame tail of a multi-GPU tile on a WINDOW of the sources instead of the whole volume --------------------------------
// psi^-1(x) is the fixed point of p <- x - u(p) started at x (vector_fields.cu:111-138): every p it visits, and psi^-1(x) itself, lies
// within r = max |psi - id| (component-wise, over the whole volume) of x, and the trilinear samplers read floor(p) and floor(p) + 1
// (include/sobfu/cuda/utils.hpp:124-164, 50-86).  So a tile needs psi -- and, for the canonical -> live warp at psi^-1(x), phi_global --
// only on its owned cells widened by ceil(r) + 1 cells, not the all-gathered volume (134 + 268 MB per frame at 256^3 that do not
// shrink with N).  The window is an array (wd) whose cell (0, 0, 0) is global cell wb; the clamps act on the GLOBAL extents pd exactly
// as in the whole-volume samplers, so the bits are the same.  A sample that would fall outside the window (a reach bound that did not
// hold) is clamped into it AND recorded in *violation: the host then redoes the frame's tail on all-gathered sources.
struct Window {
    Dims pd, wd, wb;
    int* violation;
};
SOBFU_DEV size_t win_idx(const Window& w, int x, int y, int z) {
    const int lx = x - w.wb.x, ly = y - w.wb.y, lz = z - w.wb.z;
    if ((unsigned) lx >= (unsigned) w.wd.x || (unsigned) ly >= (unsigned) w.wd.y || (unsigned) lz >= (unsigned) w.wd.z) {
        if (w.violation != nullptr) *w.violation = 1;
        return vidx(w.wd, min(max(lx, 0), w.wd.x - 1), min(max(ly, 0), w.wd.y - 1), min(max(lz, 0), w.wd.z - 1));
    }
    return vidx(w.wd, lx, ly, lz);
}
SOBFU_DEV float4 disp_at_win(const float4* __restrict__ psi, const Window& w, int x, int y, int z) {
    return sub4(psi[win_idx(w, x, y, z)], f4((float) x, (float) y, (float) z));
}
// interpolate_tsdf (utils.hpp:50-86) on a window
SOBFU_DEV float2 interp_tsdf_win(const float2* __restrict__ v, const Window& w, float px, float py, float pz) {
    const Tri a = tri_setup(px, w.pd.x), b = tri_setup(py, w.pd.y), c = tri_setup(pz, w.pd.z);
    const float2 ggg = v[win_idx(w, a.g, b.g, c.g)];
    const float hhh = v[win_idx(w, a.h, b.h, c.h)].x, hhg = v[win_idx(w, a.h, b.h, c.g)].x, hgh = v[win_idx(w, a.h, b.g, c.h)].x,
                hgg = v[win_idx(w, a.h, b.g, c.g)].x;
    const float ghh = v[win_idx(w, a.g, b.h, c.h)].x, ghg = v[win_idx(w, a.g, b.h, c.g)].x, ggh = v[win_idx(w, a.g, b.g, c.h)].x;
    const float t = lerp1(lerp1(lerp1(hhh, hhg, c.t), lerp1(hgh, hgg, c.t), b.t), lerp1(lerp1(ghh, ghg, c.t), lerp1(ggh, ggg.x, c.t), b.t), a.t);
    return make_float2(t, ggg.y);
}
// the cells of `box` (local coordinates) of a tile: psi^-1 <- n_sweeps of the fixed point started at the identity, then, when phi != null,
// phi o psi^-1 -- the lane still holds psi^-1(x) when it warps (the single-GPU tail kernel above, on windows)
__global__ void __launch_bounds__(256) tile_inverse_window_kernel(const float4* __restrict__ psi_win, Window wpsi, float4* __restrict__ psi_inv, Dims d,
                                                                  Dims base, int x0, int x1, int y0, int y1, int z0, int z1, int n_sweeps) {
    const int x = x0 + blockIdx.x * kBX + threadIdx.x, y = y0 + blockIdx.y * kBY + threadIdx.y, z = z0 + blockIdx.z;
    if (x >= x1 || y >= y1 || z >= z1) return;
    const float4 id = f4((float) (x + base.x), (float) (y + base.y), (float) (z + base.z));
    // inverse_fixed_point() with the windowed sampler (interpolate_field_inv, utils.hpp:124-164, on a window): the same early exits, the same bits
    const float4 v = inverse_fixed_point([&](int a, int b, int c) { return disp_at_win(psi_win, wpsi, a, b, c); }, wpsi.pd, id, id, n_sweeps);
    psi_inv[vidx(d, x, y, z)] = v;
}
__global__ void __launch_bounds__(256) tile_apply_window_kernel(const float2* __restrict__ phi_win, Window wphi, float2* __restrict__ out,
                                                                const float4* __restrict__ psi, Dims d, int x0, int x1, int y0, int y1, int z0, int z1) {
    const int x = x0 + blockIdx.x * kBX + threadIdx.x, y = y0 + blockIdx.y * kBY + threadIdx.y, z = z0 + blockIdx.z;
    if (x >= x1 || y >= y1 || z >= z1) return;
    const size_t i = vidx(d, x, y, z);
    const float4 p = psi[i];
    out[i] = interp_tsdf_win(phi_win, wphi, p.x, p.y, p.z);
}
// max over the cells of `box` of max(|psi.x - x|, |psi.y - y|, |psi.z - z|) (global coordinates) -> atomicMax on the float's bit pattern
__global__ void __launch_bounds__(256) tile_max_displacement_kernel(const float4* __restrict__ psi, Dims d, Dims base, int x0, int x1, int y0, int y1,
                                                                    int z0, int z1, uint32_t* __restrict__ out) {
    const int x = x0 + blockIdx.x * kBX + threadIdx.x, y = y0 + blockIdx.y * kBY + threadIdx.y, z = z0 + blockIdx.z;
    float m = 0.f;
    if (x < x1 && y < y1 && z < z1) {
        const float4 p = psi[vidx(d, x, y, z)];
        m = fmaxf(fmaxf(fabsf(p.x - (float) (x + base.x)), fabsf(p.y - (float) (y + base.y))), fabsf(p.z - (float) (z + base.z)));
        if (!(m == m) || m > 3.0e38f) m = 3.0e38f;  // NaN / inf: "unbounded" (the caller falls back to the all-gather)
    }
    uint32_t b = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) b = max(b, (uint32_t) __shfl_xor((int) b, o, 64));
    if (threadIdx.x == 0 && b != 0u) atomicMax(out, b);
}

// TsdfDifferentiator::operator() -- vector_fields.cu:157-208 (mirrored neighbour on boundary faces => exact 0)
// (A z-marching form of the three differentiators -- centre values of planes z - 1, z, z + 1 in registers -- was measured in round 6: no
// faster; their memory traffic is already the compulsory one, 1.05 - 1.15 x: profiles/r06/launcher_table_256.md.)
template <bool NT>
__global__ void __launch_bounds__(256) tsdf_gradient_kernel(const float2* __restrict__ vol, float4* __restrict__ grad, Dims d) {
    VOXEL_XYZ_XCD(d);
    int x1 = x + 1, x2 = x - 1, y1 = y + 1, y2 = y - 1, z1 = z + 1, z2 = z - 1;
    if (x == 0) x2 = x + 1; else if (x == d.x - 1) x1 = x - 1;
    if (y == 0) y2 = y + 1; else if (y == d.y - 1) y1 = y - 1;
    if (z == 0) z2 = z + 1; else if (z == d.z - 1) z1 = z - 1;
    x1 = min(x1, d.x - 1), x2 = min(x2, d.x - 1), y1 = min(y1, d.y - 1), y2 = min(y2, d.y - 1), z1 = min(z1, d.z - 1), z2 = min(z2, d.z - 1);  // (an extent of 1: the reference reads out of bounds)
    float nx = (vol[vidx(d, x1, y, z)].x - vol[vidx(d, x2, y, z)].x) / 2.f;
    float ny = (vol[vidx(d, x, y1, z)].x - vol[vidx(d, x, y2, z)].x) / 2.f;
    float nz = (vol[vidx(d, x, y, z1)].x - vol[vidx(d, x, y, z2)].x) / 2.f;
    st4<NT>(&grad[vidx(d, x, y, z)], f4(nx, ny, nz));
}

// SecondOrderDifferentiator::laplacian -- vector_fields.cu:291-337 (both neighbours <- centre on a boundary face)
template <bool NT>
__global__ void __launch_bounds__(256) laplacian_kernel(const float4* __restrict__ psi, float4* __restrict__ L, Dims d) {
    VOXEL_XYZ_XCD(d);
    int x1 = x + 1, x2 = x - 1, y1 = y + 1, y2 = y - 1, z1 = z + 1, z2 = z - 1;
    if (x == 0 || x == d.x - 1) x1 = x2 = x;
    if (y == 0 || y == d.y - 1) y1 = y2 = y;
    if (z == 0 || z == d.z - 1) z1 = z2 = z;
    float4 v = mul4(psi[vidx(d, x, y, z)], -6.f);
    v = add4(v, psi[vidx(d, x1, y, z)]);
    v = add4(v, psi[vidx(d, x2, y, z)]);
    v = add4(v, psi[vidx(d, x, y1, z)]);
    v = add4(v, psi[vidx(d, x, y2, z)]);
    v = add4(v, psi[vidx(d, x, y, z1)]);
    v = add4(v, psi[vidx(d, x, y, z2)]);
    st4<NT>(&L[vidx(d, x, y, z)], mul4(v, -1.f));
}

// Differentiator::operator()(J, mode) -- vector_fields.cu:415-472.  A voxel's Jacobian is 4 x float4 = 64 B: stored lane by lane, each of the
// four store instructions of a wave scatters 16-byte pieces at a 64-byte stride (eight write requests per 128-byte line: 488 us at 256^3, the
// write path saturated).  Here a wave transposes its 64 Jacobians through 4 KiB of LDS and stores four CONTIGUOUS 1 KiB segments.
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) jacobian_kernel(const float4* __restrict__ psi, float4* __restrict__ J, Dims d) {
    __shared__ float4 t_rows[kBY][4 * kBX];  // per wave: voxel l's row r at [4 * l + r]
    int bx_, by_, z;
    xcd_tile(bx_, by_, z);
    const int x0 = bx_ * kBX, lane = threadIdx.x, y = by_ * kBY + threadIdx.y;
    if (y >= d.y) return;                    // (wave-uniform: a wave is one row)
    const int x = min(x0 + lane, d.x - 1);   // lanes beyond the row compute a copy of its last voxel and store nothing
    int x1 = x + 1, x2 = x - 1, y1 = y + 1, y2 = y - 1, z1 = z + 1, z2 = z - 1;
    if (x == 0) x2 = x + 1; else if (x == d.x - 1) x1 = x - 1;
    if (y == 0) y2 = y + 1; else if (y == d.y - 1) y1 = y - 1;
    if (z == 0) z2 = z + 1; else if (z == d.z - 1) z1 = z - 1;
    x1 = min(x1, d.x - 1), x2 = min(x2, d.x - 1), y1 = min(y1, d.y - 1), y2 = min(y2, d.y - 1), z1 = min(z1, d.z - 1), z2 = min(z2, d.z - 1);  // (an extent of 1: the reference reads out of bounds)
    auto P = [&](int a, int b, int c) { return MODE == 0 ? psi[vidx(d, a, b, c)] : disp_at(psi, d, a, b, c); };
    const float4 jx = half4(sub4(P(x1, y, z), P(x2, y, z)));
    const float4 jy = half4(sub4(P(x, y1, z), P(x, y2, z)));
    const float4 jz = half4(sub4(P(x, y, z1), P(x, y, z2)));
    float4* t = t_rows[threadIdx.y];
    t[4 * lane + 0] = f4(jx.x, jy.x, jz.x);
    t[4 * lane + 1] = f4(jx.y, jy.y, jz.y);
    t[4 * lane + 2] = f4(jx.z, jy.z, jz.z);
    t[4 * lane + 3] = f4(0.f, 0.f, 0.f);  // the reference leaves row 3 uninitialised
    __builtin_amdgcn_wave_barrier();      // the wave's own LDS traffic is in order: no workgroup barrier
    float4* o = J + 4 * vidx(d, x0, y, z);
    const int n = 4 * min(kBX, d.x - x0);  // float4s this wave owns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = i * kBX + lane;
        if (e < n) st4<NT>(&o[e], t[e]);
    }
}

}  // namespace

#define LAUNCH_VOXEL(kern, X, Y, Z, stream, ...) \
    hipLaunchKernelGGL((kern), voxel_grid(X, Y, Z), voxel_block(), 0, (hipStream_t) (stream), __VA_ARGS__)
// kern<.., NT>: the streaming instantiation on grids beyond the Infinity Cache (launcher_streams, sobfu_device.hpp), the plain one below
#define LAUNCH_VOXEL_NT(kern_streaming, kern_plain, X, Y, Z, stream, ...)                     \
    do {                                                                                     \
        if (launcher_streams(X, Y, Z)) LAUNCH_VOXEL(kern_streaming, X, Y, Z, stream, __VA_ARGS__); \
        else LAUNCH_VOXEL(kern_plain, X, Y, Z, stream, __VA_ARGS__);                          \
    } while (0)

extern "C" {

int sobfu_hip_clear_field(float* d_field, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_field && X > 0 && Y > 0 && Z > 0);
    return (int) hipMemsetAsync(d_field, 0, sizeof(float4) * (size_t) X * Y * Z, (hipStream_t) stream);
}

int sobfu_hip_init_identity(float* d_psi, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && X > 0 && Y > 0 && Z > 0);
    LAUNCH_VOXEL(init_identity_kernel, X, Y, Z, stream, (float4*) d_psi, Dims{X, Y, Z}, Dims{0, 0, 0});
    return (int) hipGetLastError();
}

// 3-D tiles: local arrays (Lx, Ly, Lz) whose cell (0, 0, 0) is global cell (xb, yb, zb) of the (Xg, Yg, Zg) volume
int sobfu_hip_tile3_init_identity(float* d_psi, int Lx, int Ly, int Lz, int xb, int yb, int zb, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && Lx > 0 && Ly > 0 && Lz > 0 && xb >= 0 && yb >= 0 && zb >= 0);
    LAUNCH_VOXEL(init_identity_kernel, Lx, Ly, Lz, stream, (float4*) d_psi, Dims{Lx, Ly, Lz}, Dims{xb, yb, zb});
    return (int) hipGetLastError();
}

int sobfu_hip_tile3_apply(const float* d_phi, int Xg, int Yg, int Zg, float* d_phi_warped, const float* d_psi, int Lx, int Ly, int Lz,
                          void* stream) {
    SOBFU_CHECK_ARGS(d_phi && d_phi_warped && d_psi && Lx > 0 && Ly > 0 && Lz > 0 && Xg > 0 && Yg > 0 && Zg > 0 && d_phi != d_phi_warped);
    LAUNCH_VOXEL(apply_kernel<false>, Lx, Ly, Lz, stream, (const float2*) d_phi, (float2*) d_phi_warped, (const float4*) d_psi, Dims{Lx, Ly, Lz},
                 Dims{Xg, Yg, Zg});
    return (int) hipGetLastError();
}

int sobfu_hip_tile3_estimate_inverse(const float* d_psi, int Xg, int Yg, int Zg, float* d_psi_inv, int Lx, int Ly, int Lz, int xb, int yb,
                                     int zb, int n_sweeps, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_psi_inv && Lx > 0 && Ly > 0 && Lz > 0 && Xg > 0 && Yg > 0 && Zg > 0 && xb >= 0 && yb >= 0 && zb >= 0 &&
                     n_sweeps >= 0 && d_psi != d_psi_inv);
    if (n_sweeps == 0) return 0;
    LAUNCH_VOXEL(inverse_fixed_point_kernel<false>, Lx, Ly, Lz, stream, (const float4*) d_psi, (float4*) d_psi_inv, Dims{Lx, Ly, Lz},
                 Dims{Xg, Yg, Zg}, Dims{xb, yb, zb}, n_sweeps);
    return (int) hipGetLastError();
}

// ---- windowed per-frame tail of a tile (see Window above).  win = (Wx, Wy, Wz, wbx, wby, wbz): extents of the window array and the
// global cell of its cell (0, 0, 0); box = (x0, x1, y0, y1, z0, z1): the LOCAL cells to produce (a tile's owned cells); d_violation: an
// int the caller zeroed -- set to 1 when a sample fell outside the window (results are then not to be used).
static bool win_ok(const int win[6], int Xg, int Yg, int Zg) {
    return win[0] > 0 && win[1] > 0 && win[2] > 0 && win[3] >= 0 && win[4] >= 0 && win[5] >= 0 && win[3] + win[0] <= Xg && win[4] + win[1] <= Yg &&
           win[5] + win[2] <= Zg;
}
static bool lbox_ok(const int b[6], int Lx, int Ly, int Lz) {
    return b[0] >= 0 && b[0] < b[1] && b[1] <= Lx && b[2] >= 0 && b[2] < b[3] && b[3] <= Ly && b[4] >= 0 && b[4] < b[5] && b[5] <= Lz;
}
#define LAUNCH_BOX(kern, b, stream, ...) \
    hipLaunchKernelGGL(kern, voxel_grid((b)[1] - (b)[0], (b)[3] - (b)[2], (b)[5] - (b)[4]), voxel_block(), 0, (hipStream_t) (stream), __VA_ARGS__)

int sobfu_hip_tile3_estimate_inverse_window(const float* d_psi_win, const int win[6], int Xg, int Yg, int Zg, float* d_psi_inv, int Lx, int Ly, int Lz,
                                            int xb, int yb, int zb, const int box[6], int n_sweeps, int* d_violation, void* stream) {
    SOBFU_CHECK_ARGS(d_psi_win && win && d_psi_inv && box && Lx > 0 && Ly > 0 && Lz > 0 && Xg > 0 && Yg > 0 && Zg > 0 && xb >= 0 && yb >= 0 && zb >= 0 &&
                     n_sweeps >= 0 && d_psi_win != d_psi_inv && win_ok(win, Xg, Yg, Zg) && lbox_ok(box, Lx, Ly, Lz));
    const Window w{{Xg, Yg, Zg}, {win[0], win[1], win[2]}, {win[3], win[4], win[5]}, d_violation};
    LAUNCH_BOX(tile_inverse_window_kernel, box, stream, (const float4*) d_psi_win, w, (float4*) d_psi_inv, Dims{Lx, Ly, Lz}, Dims{xb, yb, zb}, box[0],
               box[1], box[2], box[3], box[4], box[5], n_sweeps);
    return (int) hipGetLastError();
}

int sobfu_hip_tile3_apply_window(const float* d_phi_win, const int win[6], int Xg, int Yg, int Zg, float* d_phi_warped, const float* d_psi, int Lx,
                                 int Ly, int Lz, const int box[6], int* d_violation, void* stream) {
    SOBFU_CHECK_ARGS(d_phi_win && win && d_phi_warped && d_psi && box && Lx > 0 && Ly > 0 && Lz > 0 && Xg > 0 && Yg > 0 && Zg > 0 &&
                     d_phi_win != d_phi_warped && win_ok(win, Xg, Yg, Zg) && lbox_ok(box, Lx, Ly, Lz));
    const Window w{{Xg, Yg, Zg}, {win[0], win[1], win[2]}, {win[3], win[4], win[5]}, d_violation};
    LAUNCH_BOX(tile_apply_window_kernel, box, stream, (const float2*) d_phi_win, w, (float2*) d_phi_warped, (const float4*) d_psi, Dims{Lx, Ly, Lz},
               box[0], box[1], box[2], box[3], box[4], box[5]);
    return (int) hipGetLastError();
}

int sobfu_hip_tile3_max_displacement(const float* d_psi, int Lx, int Ly, int Lz, int xb, int yb, int zb, const int box[6], uint32_t* d_max_bits,
                                     void* stream) {
    SOBFU_CHECK_ARGS(d_psi && box && d_max_bits && Lx > 0 && Ly > 0 && Lz > 0 && xb >= 0 && yb >= 0 && zb >= 0 && lbox_ok(box, Lx, Ly, Lz));
    LAUNCH_BOX(tile_max_displacement_kernel, box, stream, (const float4*) d_psi, Dims{Lx, Ly, Lz}, Dims{xb, yb, zb}, box[0], box[1], box[2], box[3],
               box[4], box[5], d_max_bits);
    return (int) hipGetLastError();
}

int sobfu_hip_apply(const float* d_phi, float* d_phi_warped, const float* d_psi, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_phi && d_phi_warped && d_psi && X > 0 && Y > 0 && Z > 0 && d_phi != d_phi_warped);
    LAUNCH_VOXEL_NT(apply_kernel<true>, apply_kernel<false>, X, Y, Z, stream, (const float2*) d_phi, (float2*) d_phi_warped, (const float4*) d_psi, Dims{X, Y, Z},
                    Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_estimate_inverse(const float* d_psi, float* d_psi_inv, int X, int Y, int Z, int n_sweeps, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_psi_inv && X > 0 && Y > 0 && Z > 0 && n_sweeps >= 0 && d_psi != d_psi_inv);
    if (n_sweeps == 0) return 0;
    LAUNCH_VOXEL_NT(inverse_fixed_point_kernel<true>, inverse_fixed_point_kernel<false>, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_psi_inv, Dims{X, Y, Z}, Dims{X, Y, Z},
                    Dims{0, 0, 0}, n_sweeps);
    return (int) hipGetLastError();
}

int sobfu_hip_inverse_and_warp(const float* d_psi, float* d_psi_inv, const float* d_phi, float* d_phi_warped, int X, int Y, int Z,
                               int n_sweeps, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_psi_inv && d_phi && d_phi_warped && X > 0 && Y > 0 && Z > 0 && n_sweeps >= 0 && d_psi != d_psi_inv &&
                     d_phi != d_phi_warped);
    LAUNCH_VOXEL_NT(inverse_from_identity_and_warp_kernel<true>, inverse_from_identity_and_warp_kernel<false>, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_psi_inv, (const float2*) d_phi,
                    (float2*) d_phi_warped, Dims{X, Y, Z}, n_sweeps);
    return (int) hipGetLastError();
}

int sobfu_hip_tsdf_gradient(const float* d_vol, float* d_grad, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_vol && d_grad && X > 0 && Y > 0 && Z > 0);
    LAUNCH_VOXEL_NT(tsdf_gradient_kernel<true>, tsdf_gradient_kernel<false>, X, Y, Z, stream, (const float2*) d_vol, (float4*) d_grad, Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_laplacian(const float* d_psi, float* d_L, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_L && X > 0 && Y > 0 && Z > 0 && d_psi != d_L);
    LAUNCH_VOXEL_NT(laplacian_kernel<true>, laplacian_kernel<false>, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_L, Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_jacobian(const float* d_psi, float* d_J, int X, int Y, int Z, int mode, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_J && X > 0 && Y > 0 && Z > 0 && (mode == 0 || mode == 1));
    if (mode == 0) LAUNCH_VOXEL_NT((jacobian_kernel<0, true>), (jacobian_kernel<0, false>), X, Y, Z, stream, (const float4*) d_psi, (float4*) d_J, Dims{X, Y, Z});
    else LAUNCH_VOXEL_NT((jacobian_kernel<1, true>), (jacobian_kernel<1, false>), X, Y, Z, stream, (const float4*) d_psi, (float4*) d_J, Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_clear_jacobian(float* d_J, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_J && X > 0 && Y > 0 && Z > 0);
    return (int) hipMemsetAsync(d_J, 0, 64 * (size_t) X * Y * Z, (hipStream_t) stream);
}

}  // extern "C"
