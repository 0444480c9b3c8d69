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

// init_identity_kernel -- vector_fields.cu:64-79.  (float) z equals the reference's z-fold sum of 1.f exactly.
// base: global coordinates of local cell (0, 0, 0) (multi-GPU tiles; 0 on a single GPU)
__global__ void __launch_bounds__(256) init_identity_kernel(float4* __restrict__ psi, Dims d, Dims base) {
    VOXEL_XYZ(d);
    psi[vidx(d, x, y, z)] = f4((float) (x + base.x), (float) (y + base.y), (float) (z + base.z));
}

// apply_kernel -- vector_fields.cu:81-100
// pd: dims of phi (the whole volume); d: dims of psi / out (== pd on a single GPU, a z-slab on multiple GPUs)
__global__ void __launch_bounds__(256) apply_kernel(const float2* __restrict__ phi, float2* __restrict__ out,
                                                    const float4* __restrict__ psi, Dims d, Dims pd) {
    VOXEL_XYZ(d);
    size_t i = vidx(d, x, y, z);
    float4 p = psi[i];
    out[i]   = interp_tsdf(phi, pd, p.x, p.y, p.z);
}

// estimate_inverse_kernel x n_sweeps -- vector_fields.cu:111-138.  A sweep reads psi (never written) and the
// voxel's OWN psi_inv value only, so the reference's 48 launches collapse into one kernel that iterates the
// fixed point p <- x - u(p) in registers: psi_inv is read once and written once instead of 48 times, bit-identical result.
// The float iteration soon repeats itself: once a sweep returns its input bit for bit every later sweep does too, and once
// it returns the value before that (period 2) the tail just alternates -- in both cases the value after n_sweeps is known
// and the lane stops (same bits as running all sweeps; typically < 10 of the 48 are needed).
SOBFU_DEV float4 inverse_fixed_point(const float4* __restrict__ psi, const Dims& pd, float4 v, const float4& id, int n_sweeps) {
    float4 w = v;  // p_it = v, p_(it-1) = w
    auto same = [](const float4& a, const float4& b) {
        return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
               __float_as_uint(a.z) == __float_as_uint(b.z);
    };
    for (int it = 0; it < n_sweeps; ++it) {
        const float4 u  = interp_disp(psi, pd, v.x, v.y, v.z);
        const float4 nv = sub4(id, mul4(u, 1.f));  // p_(it+1)
        const int left  = n_sweeps - (it + 1);     // sweeps still to run after this one
        if (same(nv, v)) {                         // fixed point
            v = nv;
            break;
        }
        if (it > 0 && same(nv, w)) {               // period 2: ..., w, v, w (= nv), v, w, ...
            v = (left & 1) ? v : nv;
            break;
        }
        w = v;
        v = nv;
    }
    return v;
}

__global__ void __launch_bounds__(256) inverse_fixed_point_kernel(const float4* __restrict__ psi, float4* __restrict__ psi_inv,
                                                                  Dims d, Dims pd, Dims base, int n_sweeps) {
    VOXEL_XYZ(d);
    size_t i = vidx(d, x, y, z);
    const float4 id = f4((float) (x + base.x), (float) (y + base.y), (float) (z + base.z));
    psi_inv[i] = inverse_fixed_point(psi, pd, psi_inv[i], id, n_sweeps);
}

// The tail of Solver::estimate_psi in one pass (solver.cu:196-199): psi^-1 <- identity, 48 sweeps, phi_global o psi^-1.
// The lane still holds psi^-1(x) when it warps phi_global there: no identity field is written and read back, psi^-1 is not
// re-read by the warp.
__global__ void __launch_bounds__(256) inverse_from_identity_and_warp_kernel(const float4* __restrict__ psi, float4* __restrict__ psi_inv,
                                                                             const float2* __restrict__ phi, float2* __restrict__ phi_warped,
                                                                             Dims d, int n_sweeps) {
    VOXEL_XYZ(d);
    size_t i = vidx(d, x, y, z);
    const float4 id = f4((float) x, (float) y, (float) z);
    const float4 v  = inverse_fixed_point(psi, d, id, id, n_sweeps);
    psi_inv[i]      = v;
    phi_warped[i]   = interp_tsdf(phi, d, v.x, v.y, v.z);
}

// TsdfDifferentiator::operator() -- vector_fields.cu:157-208 (mirrored neighbour on boundary faces => exact 0)
__global__ void __launch_bounds__(256) tsdf_gradient_kernel(const float2* __restrict__ vol, float4* __restrict__ grad, Dims d) {
    VOXEL_XYZ(d);
    int x1 = x + 1, x2 = x - 1, y1 = y + 1, y2 = y - 1, z1 = z + 1, z2 = z - 1;
    if (x == 0) x2 = x + 1; else if (x == d.x - 1) x1 = x - 1;
    if (y == 0) y2 = y + 1; else if (y == d.y - 1) y1 = y - 1;
    if (z == 0) z2 = z + 1; else if (z == d.z - 1) z1 = z - 1;
    float nx = (vol[vidx(d, x1, y, z)].x - vol[vidx(d, x2, y, z)].x) / 2.f;
    float ny = (vol[vidx(d, x, y1, z)].x - vol[vidx(d, x, y2, z)].x) / 2.f;
    float nz = (vol[vidx(d, x, y, z1)].x - vol[vidx(d, x, y, z2)].x) / 2.f;
    grad[vidx(d, x, y, z)] = f4(nx, ny, nz);
}

// SecondOrderDifferentiator::laplacian -- vector_fields.cu:291-337 (both neighbours <- centre on a boundary face)
__global__ void __launch_bounds__(256) laplacian_kernel(const float4* __restrict__ psi, float4* __restrict__ L, Dims d) {
    VOXEL_XYZ(d);
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
    L[vidx(d, x, y, z)] = mul4(v, -1.f);
}

// Differentiator::operator()(J, mode) -- vector_fields.cu:415-472
template <int MODE>
__global__ void __launch_bounds__(256) jacobian_kernel(const float4* __restrict__ psi, float4* __restrict__ J, Dims d) {
    VOXEL_XYZ(d);
    int x1 = x + 1, x2 = x - 1, y1 = y + 1, y2 = y - 1, z1 = z + 1, z2 = z - 1;
    if (x == 0) x2 = x + 1; else if (x == d.x - 1) x1 = x - 1;
    if (y == 0) y2 = y + 1; else if (y == d.y - 1) y1 = y - 1;
    if (z == 0) z2 = z + 1; else if (z == d.z - 1) z1 = z - 1;
    auto P = [&](int a, int b, int c) { return MODE == 0 ? psi[vidx(d, a, b, c)] : disp_at(psi, d, a, b, c); };
    float4 jx = half4(sub4(P(x1, y, z), P(x2, y, z)));
    float4 jy = half4(sub4(P(x, y1, z), P(x, y2, z)));
    float4 jz = half4(sub4(P(x, y, z1), P(x, y, z2)));
    float4* o = J + 4 * vidx(d, x, y, z);
    o[0] = f4(jx.x, jy.x, jz.x);
    o[1] = f4(jx.y, jy.y, jz.y);
    o[2] = f4(jx.z, jy.z, jz.z);
    o[3] = f4(0.f, 0.f, 0.f);  // the reference leaves row 3 uninitialised
}

}  // namespace

#define LAUNCH_VOXEL(kern, X, Y, Z, stream, ...) \
    hipLaunchKernelGGL(kern, voxel_grid(X, Y, Z), voxel_block(), 0, (hipStream_t) (stream), __VA_ARGS__)

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
    LAUNCH_VOXEL(apply_kernel, Lx, Ly, Lz, stream, (const float2*) d_phi, (float2*) d_phi_warped, (const float4*) d_psi, Dims{Lx, Ly, Lz},
                 Dims{Xg, Yg, Zg});
    return (int) hipGetLastError();
}

int sobfu_hip_tile3_estimate_inverse(const float* d_psi, int Xg, int Yg, int Zg, float* d_psi_inv, int Lx, int Ly, int Lz, int xb, int yb,
                                     int zb, int n_sweeps, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_psi_inv && Lx > 0 && Ly > 0 && Lz > 0 && Xg > 0 && Yg > 0 && Zg > 0 && xb >= 0 && yb >= 0 && zb >= 0 &&
                     n_sweeps >= 0 && d_psi != d_psi_inv);
    if (n_sweeps == 0) return 0;
    LAUNCH_VOXEL(inverse_fixed_point_kernel, Lx, Ly, Lz, stream, (const float4*) d_psi, (float4*) d_psi_inv, Dims{Lx, Ly, Lz},
                 Dims{Xg, Yg, Zg}, Dims{xb, yb, zb}, n_sweeps);
    return (int) hipGetLastError();
}

int sobfu_hip_apply(const float* d_phi, float* d_phi_warped, const float* d_psi, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_phi && d_phi_warped && d_psi && X > 0 && Y > 0 && Z > 0 && d_phi != d_phi_warped);
    LAUNCH_VOXEL(apply_kernel, X, Y, Z, stream, (const float2*) d_phi, (float2*) d_phi_warped, (const float4*) d_psi, Dims{X, Y, Z},
                 Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_estimate_inverse(const float* d_psi, float* d_psi_inv, int X, int Y, int Z, int n_sweeps, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_psi_inv && X > 0 && Y > 0 && Z > 0 && n_sweeps >= 0 && d_psi != d_psi_inv);
    if (n_sweeps == 0) return 0;
    LAUNCH_VOXEL(inverse_fixed_point_kernel, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_psi_inv, Dims{X, Y, Z}, Dims{X, Y, Z},
                 Dims{0, 0, 0}, n_sweeps);
    return (int) hipGetLastError();
}

int sobfu_hip_inverse_and_warp(const float* d_psi, float* d_psi_inv, const float* d_phi, float* d_phi_warped, int X, int Y, int Z,
                               int n_sweeps, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_psi_inv && d_phi && d_phi_warped && X > 0 && Y > 0 && Z > 0 && n_sweeps >= 0 && d_psi != d_psi_inv &&
                     d_phi != d_phi_warped);
    LAUNCH_VOXEL(inverse_from_identity_and_warp_kernel, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_psi_inv, (const float2*) d_phi,
                 (float2*) d_phi_warped, Dims{X, Y, Z}, n_sweeps);
    return (int) hipGetLastError();
}

int sobfu_hip_tsdf_gradient(const float* d_vol, float* d_grad, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_vol && d_grad && X > 0 && Y > 0 && Z > 0);
    LAUNCH_VOXEL(tsdf_gradient_kernel, X, Y, Z, stream, (const float2*) d_vol, (float4*) d_grad, Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_laplacian(const float* d_psi, float* d_L, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_L && X > 0 && Y > 0 && Z > 0 && d_psi != d_L);
    LAUNCH_VOXEL(laplacian_kernel, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_L, Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_jacobian(const float* d_psi, float* d_J, int X, int Y, int Z, int mode, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_J && X > 0 && Y > 0 && Z > 0 && (mode == 0 || mode == 1));
    if (mode == 0) LAUNCH_VOXEL(jacobian_kernel<0>, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_J, Dims{X, Y, Z});
    else LAUNCH_VOXEL(jacobian_kernel<1>, X, Y, Z, stream, (const float4*) d_psi, (float4*) d_J, Dims{X, Y, Z});
    return (int) hipGetLastError();
}

int sobfu_hip_clear_jacobian(float* d_J, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_J && X > 0 && Y > 0 && Z > 0);
    return (int) hipMemsetAsync(d_J, 0, 64 * (size_t) X * Y * Z, (hipStream_t) stream);
}

}  // extern "C"
