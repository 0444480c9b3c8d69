// Launcher-for-launcher counterparts of include/sobfu/solver.hpp:109-136 for gfx950: potential gradient, the three 1-D Sobolev
// convolutions, psi update -- the reference's own decomposition of an iteration, for callers that drive it step by step (its tests do).
// The solver does not run these in its loop: it runs the two fused passes of solver_kernels.hip, whose results are bit-identical.
// One lane per voxel, a wave = 64 consecutive x; the x / y convolutions take their taps through the caches under the XCD-aware tile map
// (sobfu_device.hpp), the z convolution marches along z with its seven taps in registers.
#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"

using namespace sobfu_hip;

namespace {

struct Taps {
    float s[7];
};


// calculate_potential_gradient_kernel -- solver.cu:15-33
template <bool NT>
__global__ void __launch_bounds__(256) potential_gradient_kernel(const float2* __restrict__ pnp, const float2* __restrict__ pg,
                                                                 const float4* __restrict__ grad, const float4* __restrict__ L,
                                                                 float4* __restrict__ nU, float w_reg, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float d = ld2<NT>(&pnp[i]).x - ld2<NT>(&pg[i]).x;
    st4<NT>(&nU[i], add4(mul4(ld4<NT>(&grad[i]), d), mul4(ld4<NT>(&L[i]), w_reg)));
}

// convolution_{rows,columns,depth}_kernel -- solver.cu:237-446: sum = 0; for j=-3..3: sum += S[3-j]*src(clamp(i+j))
template <int AXIS, bool NT>
__global__ void __launch_bounds__(256) conv1d_kernel(float4* __restrict__ dst, const float4* __restrict__ src, Taps S, Dims d) {
    int bx, by, z;
    xcd_tile(bx, by, z);
    const int x = bx * kBX + threadIdx.x, y = by * kBY + threadIdx.y;
    if (x >= d.x || y >= d.y) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        int xx = x, yy = y, zz = z;
        if (AXIS == 0) xx = min(max(x + j, 0), d.x - 1);
        if (AXIS == 1) yy = min(max(y + j, 0), d.y - 1);
        float4 v = src[vidx(d, xx, yy, zz)];
        float s  = S.s[3 - j];
        sx += v.x * s;
        sy += v.y * s;
        sz += v.z * s;
    }
    float4* o = dst + vidx(d, x, y, z);
    if (AXIS == 0) {
        st4<NT>(o, f4(sx, sy, sz));  // rows assign (solver.cu:290)
    } else {                         // columns accumulate, w untouched (solver.cu:366; utils.hpp:253-258)
        float4 c = ld4<NT>(o);
        c.x += sx;
        c.y += sy;
        c.z += sz;
        st4<NT>(o, c);
    }
}

// update_psi_kernel -- solver.cu:53-69
template <bool NT>
__global__ void __launch_bounds__(256) update_psi_kernel(float4* __restrict__ psi, const float4* __restrict__ nUS,
                                                         float4* __restrict__ updates, float alpha, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 u = mul4(ld4<NT>(&nUS[i]), alpha);
    st4<NT>(&updates[i], u);
    float4 p = ld4<NT>(&psi[i]);
    p.x -= u.x;
    p.y -= u.y;
    p.z -= u.z;
    st4<NT>(&psi[i], p);
}


// convolution_depth_kernel -- solver.cu:372-446, marching: a lane owns an (x, y) column of a z chunk and keeps the seven taps in registers, so
// a plane of src is read once per chunk (+ 6 when a chunk starts) instead of seven times through the caches; the next plane's src and dst
// loads are issued before this plane's arithmetic.  At 256^3 (round 6): one lane per voxel through the caches 164 us, marching in chunks of 32 / 64 / 128
// planes 145 / 138 / 134 us (the reference's own shared-memory kernel on this GPU: 159).  Same products, same ascending-j sum as conv1d_kernel.
template <bool NT>
__global__ void __launch_bounds__(256) conv_depth_march_kernel(float4* __restrict__ dst, const float4* __restrict__ src, Taps S, Dims d, int zc) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y;
    if (x >= d.x || y >= d.y) return;
    const int z0 = blockIdx.z * zc, z1 = min(z0 + zc, d.z);
    const size_t plane = (size_t) d.x * d.y, col = vidx(d, x, y, 0);
    auto ld = [&](int z) { return ld4<NT>(&src[col + plane * (size_t) min(max(z, 0), d.z - 1)]); };
    float4 w0 = ld(z0 - 3), w1 = ld(z0 - 2), w2 = ld(z0 - 1), w3 = ld(z0), w4 = ld(z0 + 1), w5 = ld(z0 + 2), w6 = ld(z0 + 3);
    float4 c = ld4<NT>(&dst[col + plane * (size_t) z0]);
    for (int z = z0; z < z1; ++z) {
        float4 w7 = w6, cn = c;
        if (z + 1 < z1) {
            w7 = ld(z + 4);
            cn = ld4<NT>(&dst[col + plane * (size_t) (z + 1)]);
        }
        float sx = 0.f, sy = 0.f, sz = 0.f;
#define SOBFU_TAP(w, k) sx += (w).x * S.s[k], sy += (w).y * S.s[k], sz += (w).z * S.s[k];
        SOBFU_TAP(w0, 6) SOBFU_TAP(w1, 5) SOBFU_TAP(w2, 4) SOBFU_TAP(w3, 3) SOBFU_TAP(w4, 2) SOBFU_TAP(w5, 1) SOBFU_TAP(w6, 0)
#undef SOBFU_TAP
        c.x += sx;  // accumulate, w untouched (solver.cu:443; utils.hpp:253-258)
        c.y += sy;
        c.z += sz;
        st4<NT>(&dst[col + plane * (size_t) z], c);
        w0 = w1, w1 = w2, w2 = w3, w3 = w4, w4 = w5, w5 = w6, w6 = w7, c = cn;
    }
}

}  // namespace

extern "C" {

int sobfu_hip_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_grad, const float* d_L,
                                 float* d_nabla_U, float w_reg, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_phi_n_psi && d_phi_global && d_grad && d_L && d_nabla_U && X > 0 && Y > 0 && Z > 0);
    size_t N = (size_t) X * Y * Z;
    if (launcher_streams(X, Y, Z))
        hipLaunchKernelGGL(potential_gradient_kernel<true>, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream, (const float2*) d_phi_n_psi,
                           (const float2*) d_phi_global, (const float4*) d_grad, (const float4*) d_L, (float4*) d_nabla_U, w_reg, N);
    else
        hipLaunchKernelGGL(potential_gradient_kernel<false>, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream, (const float2*) d_phi_n_psi,
                           (const float2*) d_phi_global, (const float4*) d_grad, (const float4*) d_L, (float4*) d_nabla_U, w_reg, N);
    return (int) hipGetLastError();
}

#define CONV_IMPL(name, AXIS)                                                                                       \
    int name(float* d_dst, const float* d_src, const float taps[7], int w, int h, int d, void* stream) {            \
        SOBFU_CHECK_ARGS(d_dst && d_src && taps && w > 0 && h > 0 && d > 0 && d_dst != d_src);                       \
        Taps S;                                                                                                     \
        for (int i = 0; i < 7; ++i) S.s[i] = taps[i];                                                               \
        if (launcher_streams(w, h, d))                                                                              \
            hipLaunchKernelGGL((conv1d_kernel<AXIS, true>), voxel_grid(w, h, d), voxel_block(), 0, (hipStream_t) stream, \
                               (float4*) d_dst, (const float4*) d_src, S, Dims{w, h, d});                           \
        else                                                                                                        \
            hipLaunchKernelGGL((conv1d_kernel<AXIS, false>), voxel_grid(w, h, d), voxel_block(), 0, (hipStream_t) stream, \
                               (float4*) d_dst, (const float4*) d_src, S, Dims{w, h, d});                           \
        return (int) hipGetLastError();                                                                             \
    }
CONV_IMPL(sobfu_hip_convolution_rows, 0)
CONV_IMPL(sobfu_hip_convolution_columns, 1)

int sobfu_hip_convolution_depth(float* d_dst, const float* d_src, const float taps[7], int w, int h, int d, void* stream) {
    SOBFU_CHECK_ARGS(d_dst && d_src && taps && w > 0 && h > 0 && d > 0 && d_dst != d_src);
    Taps S;
    for (int i = 0; i < 7; ++i) S.s[i] = taps[i];
    const int zc = march_zc(w, h, d);
    dim3 grid = voxel_grid(w, h, (d + zc - 1) / zc);
    if (launcher_streams(w, h, d))
        hipLaunchKernelGGL(conv_depth_march_kernel<true>, grid, voxel_block(), 0, (hipStream_t) stream, (float4*) d_dst, (const float4*) d_src, S, Dims{w, h, d}, zc);
    else
        hipLaunchKernelGGL(conv_depth_march_kernel<false>, grid, voxel_block(), 0, (hipStream_t) stream, (float4*) d_dst, (const float4*) d_src, S, Dims{w, h, d}, zc);
    return (int) hipGetLastError();
}

int sobfu_hip_update_psi(float* d_psi, const float* d_nabla_U_S, float* d_updates, float alpha, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_nabla_U_S && d_updates && X > 0 && Y > 0 && Z > 0);
    size_t N = (size_t) X * Y * Z;
    if (launcher_streams(X, Y, Z))
        hipLaunchKernelGGL(update_psi_kernel<true>, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream, (float4*) d_psi,
                           (const float4*) d_nabla_U_S, (float4*) d_updates, alpha, N);
    else
        hipLaunchKernelGGL(update_psi_kernel<false>, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream, (float4*) d_psi,
                           (const float4*) d_nabla_U_S, (float4*) d_updates, alpha, N);
    return (int) hipGetLastError();
}

}  // extern "C"
