// TSDF volume kernels + depth pre-steps for gfx950.
// Reference behaviour: src/kfusion/cuda/tsdf_volume.cu, src/kfusion/cuda/imgproc.cu:8-77,233-254.
// Launch shape: true 3-D grids (sobfu_device.hpp) instead of the reference's 2-D block(64,16) z-loop; the
// z-accumulated quantities (vc_cam.z, vc.z) are rebuilt per voxel by the same chain of float additions so
// the values stay bit-identical to the reference's running sums.
#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"

using namespace sobfu_hip;

namespace {

// z-chunked kernels: each thread handles ZC consecutive slices so the reference's running `+= vz` sums can be
// replayed cheaply (the prefix for slice z0 costs z0 additions; ZC amortises it).
constexpr int kZC = 16;

struct IntegrateArgs {
    const float* dists;
    int step, rows, cols;
    float2* vol;
    Dims d;
    float vsx, vsy, vsz, trunc, eta;
    float R[9], t[3];
    float fx, fy, cx, cy;
    int zbase;  // global z of local plane 0 (multi-GPU tiles; 0 for a whole volume)
    int xbase, ybase;  // likewise along x / y (3-D tiles)
};

// TsdfIntegrator::operator()(TsdfVolume&) -- tsdf_volume.cu:62-101
__global__ void __launch_bounds__(256) integrate_depth_kernel(IntegrateArgs a) {
    int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y;
    if (x >= a.d.x || y >= a.d.y) return;
    int z0 = blockIdx.z * kZC;
    float vcx = (x + a.xbase) * a.vsx + a.vsx / 2.f, vcy = (y + a.ybase) * a.vsy + a.vsy / 2.f, vcz = a.vsz / 2.f;
    float camx = dot3(a.R + 0, vcx, vcy, vcz) + a.t[0];
    float camy = dot3(a.R + 3, vcx, vcy, vcz) + a.t[1];
    float camz = dot3(a.R + 6, vcx, vcy, vcz) + a.t[2];
    for (int i = 0; i < z0 + a.zbase; ++i) camx += 0.f, camy += 0.f, camz += a.vsz;  // replay of `vc_cam += zstep` (:76)
    int z1 = min(z0 + kZC, a.d.z);
    for (int z = z0; z < z1; ++z, camx += 0.f, camy += 0.f, camz += a.vsz) {
        float coox = __builtin_fmaf(a.fx, camx / camz, a.cx), cooy = __builtin_fmaf(a.fy, camy / camz, a.cy);
        if (coox < 0 || cooy < 0 || coox >= (float) a.cols || cooy >= (float) a.rows) continue;
        if (!(camz > 0)) continue;
        if (!(coox == coox) || !(cooy == cooy)) continue;
        int px = (int) floorf(coox), py = (int) floorf(cooy);
        float Dp = *(const float*) ((const char*) a.dists + (size_t) py * a.step + (size_t) px * 4);
        if (Dp <= 0.f) continue;
        float psdf   = Dp - camz;
        float weight = (psdf > -a.eta) ? 1.f : 0.f;
        a.vol[vidx(a.d, x, y, z)] = pack_tsdf(psdf, a.trunc, weight);
    }
}

// TsdfIntegrator::operator()(phi_global, phi_n_psi) -- tsdf_volume.cu:103-130.  Pure streaming: 1-D grid,
// two voxels (one float4) per lane.
// (Streaming hints were measured on the three kernels above in round 6 and are NOT used: the conditional 8-byte stores of the fusion and of
// integrate(depth) leave partial lines that the L2 merges only when they are stored plainly -- fusion 66 -> 106 us with the hints at 256^3;
// the write-only initialisers gain nothing.)
__global__ void __launch_bounds__(256) integrate_fuse_kernel(float2* __restrict__ g, const float2* __restrict__ n,
                                                             size_t N, float max_weight) {
    size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= N) return;
    auto one = [&](float2 t, float2 p, bool& w) -> float2 {
        w = !(t.y == 0.f || (t.y == 1.f && (t.x == 0.f || t.x == -1.f)));
        return make_float2(__builtin_fmaf(p.y, p.x, t.x) / (p.y + 1.f), fminf(p.y + 1.f, max_weight));
    };
    if (i + 1 < N) {
        float4 t = *(const float4*) (n + i), p = *(const float4*) (g + i);
        bool w0, w1;
        float2 o0 = one(make_float2(t.x, t.y), make_float2(p.x, p.y), w0);
        float2 o1 = one(make_float2(t.z, t.w), make_float2(p.z, p.w), w1);
        if (w0 && w1) *(float4*) (g + i) = make_float4(o0.x, o0.y, o1.x, o1.y);
        else if (w0) g[i] = o0;
        else if (w1) g[i + 1] = o1;
    } else {
        bool w0;
        float2 o0 = one(n[i], g[i], w0);
        if (w0) g[i] = o0;
    }
}

enum Prim { SPHERE, BOX, ELLIPSOID, PLANE, TORUS };
struct PrimArgs {
    float2* vol;
    Dims d;
    float vsx, vsy, vsz, trunc, eta;
    float p[3];
    float r;
};

SOBFU_DEV float norm3_fma(float x, float y, float z) {  // kfusion::device::norm(float3), temp_utils.hpp:86
    float a[3] = {x, y, z};
    return __builtin_sqrtf(dot3(a, x, y, z));
}

// init_*_kernel -- tsdf_volume.cu:181-334
template <int P>
__global__ void __launch_bounds__(256) init_prim_kernel(PrimArgs a) {
    int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y;
    if (x >= a.d.x || y >= a.d.y) return;
    int z0 = blockIdx.z * kZC;
    float vx = x * a.vsx + a.vsx / 2.f, vy = y * a.vsy + a.vsy / 2.f, vz = a.vsz / 2.f;
    if (P == BOX || P == ELLIPSOID || P == TORUS) {  // centring (:189-194)
        vx = vx - a.d.x / 2.f * a.vsx;
        vy = vy - a.d.y / 2.f * a.vsy;
        vz = vz - a.d.z / 2.f * a.vsz;
    }
    for (int i = 0; i < z0; ++i) vz += a.vsz;  // replay of `vc += zstep`
    int z1 = min(z0 + kZC, a.d.z);
    for (int z = z0; z < z1; ++z, vz += a.vsz) {
        float sdf, w = 1.f;
        if (P == SPHERE) {
            // powf(d, 2) of the reference (:262) is evaluated as the correctly rounded d*d
            float dx = vx - a.p[0], dy = vy - a.p[1], dz = vz - a.p[2];
            sdf = __builtin_sqrtf(dx * dx + dy * dy + dz * dz) - a.r;
            w   = (sdf > -a.eta) ? 1.f : 0.f;
        } else if (P == BOX) {
            float dx = fabsf(vx) - a.p[0], dy = fabsf(vy) - a.p[1], dz = fabsf(vz) - a.p[2];
            sdf = fminf(fmaxf(dx, fmaxf(dy, dz)), 0.f) + norm3_fma(fmaxf(dx, 0.f), fmaxf(dy, 0.f), fmaxf(dz, 0.f));
        } else if (P == ELLIPSOID) {
            float k0 = norm3_fma(vx / a.p[0], vy / a.p[1], vz / a.p[2]);
            float k1 = norm3_fma(vx / (a.p[0] * a.p[0]), vy / (a.p[1] * a.p[1]), vz / (a.p[2] * a.p[2]));
            sdf      = k0 * (k0 - 1.f) / k1;
        } else if (P == PLANE) {
            sdf = vz - a.p[0];
        } else {
            float qx = __builtin_sqrtf(vx * vx + vz * vz) - a.p[0];
            sdf      = __builtin_sqrtf(qx * qx + vy * vy) - a.p[1];
        }
        a.vol[vidx(a.d, x, y, z)] = pack_tsdf(sdf, a.trunc, w);
    }
}

// bilateral_kernel -- imgproc.cu:8-37
__global__ void __launch_bounds__(256) bilateral_kernel(const uint16_t* src, int sstep, uint16_t* dst, int dstep,
                                                        int rows, int cols, int ksz, float sss, float sds) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows) return;
    auto S = [&](int yy, int xx) { return (int) *(const uint16_t*) ((const char*) src + (size_t) yy * sstep + (size_t) xx * 2); };
    int value = S(y, x);
    int tx = min(x - ksz / 2 + ksz, cols - 1), ty = min(y - ksz / 2 + ksz, rows - 1);
    float sum1 = 0, sum2 = 0;
    for (int cy = max(y - ksz / 2, 0); cy < ty; ++cy)
        for (int cx = max(x - ksz / 2, 0); cx < tx; ++cx) {
            int depth    = S(cy, cx);
            float space2 = (float) ((x - cx) * (x - cx) + (y - cy) * (y - cy));
            float color2 = (float) (int) ((unsigned) (value - depth) * (unsigned) (value - depth));
            float weight = expf(-(space2 * sss + color2 * sds));
            sum1 += depth * weight;
            sum2 += weight;
        }
    float q = sum1 / sum2;
    int r   = (q == q) ? (int) rintf(q) : 0;
    *(uint16_t*) ((char*) dst + (size_t) y * dstep + (size_t) x * 2) = (uint16_t) r;
}

// truncate_depth_kernel -- imgproc.cu:60-68
__global__ void truncate_depth_kernel(uint16_t* depth, int step, int rows, int cols, uint16_t max_mm) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows) return;
    uint16_t* p = (uint16_t*) ((char*) depth + (size_t) y * step) + x;
    if (*p > max_mm) *p = 0;
}

// compute_dists_kernel -- imgproc.cu:233-244 (the reference's `x < cols || y < rows` guard is a latent
// out-of-bounds write; guarded correctly here, in-range results are unaffected)
__global__ void compute_dists_kernel(const uint16_t* depth, int dstep, float* dists, int sstep, int rows, int cols,
                                     float fix, float fiy, float cx, float cy) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows) return;
    float xl = (x - cx) * fix, yl = (y - cy) * fiy;
    float lambda = __builtin_sqrtf(xl * xl + yl * yl + 1);
    int dv       = *((const uint16_t*) ((const char*) depth + (size_t) y * dstep) + x);
    *((float*) ((char*) dists + (size_t) y * sstep) + x) = dv * lambda * 0.001f;
}

inline dim3 chunk_grid(int X, int Y, int Z) { return dim3((X + kBX - 1) / kBX, (Y + kBY - 1) / kBY, (Z + kZC - 1) / kZC); }
inline dim3 img_grid(int rows, int cols) { return dim3((cols + 63) / 64, (rows + 3) / 4); }

template <int P>
int launch_prim(float* d_vol, int X, int Y, int Z, const float vs[3], float trunc, float eta, const float* p, int np,
                float r, void* stream) {
    SOBFU_CHECK_ARGS(d_vol && vs && X > 0 && Y > 0 && Z > 0);
    PrimArgs a{(float2*) d_vol, {X, Y, Z}, vs[0], vs[1], vs[2], trunc, eta, {0, 0, 0}, r};
    for (int i = 0; i < np; ++i) a.p[i] = p[i];
    hipLaunchKernelGGL(init_prim_kernel<P>, chunk_grid(X, Y, Z), voxel_block(), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

}  // namespace

extern "C" {

int sobfu_hip_clear_volume(float* d_vol, int X, int Y, int Z, void* stream) {
    SOBFU_CHECK_ARGS(d_vol && X > 0 && Y > 0 && Z > 0);
    return (int) hipMemsetAsync(d_vol, 0, sizeof(float2) * (size_t) X * Y * Z, (hipStream_t) stream);
}

int sobfu_hip_integrate_depth(const float* d_dists, int step, int rows, int cols, float* d_vol, int X, int Y, int Z,
                              const float vs[3], float trunc, float eta, const float R[9], const float t[3], float fx,
                              float fy, float cx, float cy, void* stream) {
    SOBFU_CHECK_ARGS(d_dists && d_vol && vs && R && t && X > 0 && Y > 0 && Z > 0 && rows > 0 && cols > 0 && step >= cols * 4);
    IntegrateArgs a{d_dists, step, rows, cols, (float2*) d_vol, {X, Y, Z}, vs[0], vs[1], vs[2], trunc, eta, {}, {}, fx, fy, cx, cy, 0, 0, 0};
    for (int i = 0; i < 9; ++i) a.R[i] = R[i];
    for (int i = 0; i < 3; ++i) a.t[i] = t[i];
    hipLaunchKernelGGL(integrate_depth_kernel, chunk_grid(X, Y, Z), voxel_block(), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

int sobfu_hip_tile3_integrate_depth(const float* d_dists, int step, int rows, int cols, float* d_vol_local, int Lx, int Ly, int Lz, int xb,
                                    int yb, int zb, const float vs[3], float trunc, float eta, const float R[9], const float t[3], float fx,
                                    float fy, float cx, float cy, void* stream) {
    SOBFU_CHECK_ARGS(d_dists && d_vol_local && vs && R && t && Lx > 0 && Ly > 0 && Lz > 0 && xb >= 0 && yb >= 0 && zb >= 0 && rows > 0 &&
                     cols > 0 && step >= cols * 4);
    IntegrateArgs a{d_dists, step, rows, cols, (float2*) d_vol_local, {Lx, Ly, Lz}, vs[0], vs[1], vs[2], trunc, eta, {}, {}, fx, fy, cx, cy, zb, xb, yb};
    for (int i = 0; i < 9; ++i) a.R[i] = R[i];
    for (int i = 0; i < 3; ++i) a.t[i] = t[i];
    hipLaunchKernelGGL(integrate_depth_kernel, chunk_grid(Lx, Ly, Lz), voxel_block(), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

int sobfu_hip_integrate_fuse(float* d_phi_global, const float* d_phi_n_psi, int X, int Y, int Z, float max_weight,
                             void* stream) {
    SOBFU_CHECK_ARGS(d_phi_global && d_phi_n_psi && X > 0 && Y > 0 && Z > 0);
    size_t N = (size_t) X * Y * Z, pairs = (N + 1) / 2;
    hipLaunchKernelGGL(integrate_fuse_kernel, dim3((unsigned) ((pairs + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       (float2*) d_phi_global, (const float2*) d_phi_n_psi, N, max_weight);
    return (int) hipGetLastError();
}

int sobfu_hip_init_sphere(float* d_vol, int X, int Y, int Z, const float vs[3], float trunc, float eta,
                          const float c[3], float radius, void* stream) {
    SOBFU_CHECK_ARGS(c);
    return launch_prim<SPHERE>(d_vol, X, Y, Z, vs, trunc, eta, c, 3, radius, stream);
}
int sobfu_hip_init_box(float* d_vol, int X, int Y, int Z, const float vs[3], float trunc, const float b[3], void* stream) {
    SOBFU_CHECK_ARGS(b);
    return launch_prim<BOX>(d_vol, X, Y, Z, vs, trunc, 0.f, b, 3, 0.f, stream);
}
int sobfu_hip_init_ellipsoid(float* d_vol, int X, int Y, int Z, const float vs[3], float trunc, const float r[3], void* stream) {
    SOBFU_CHECK_ARGS(r);
    return launch_prim<ELLIPSOID>(d_vol, X, Y, Z, vs, trunc, 0.f, r, 3, 0.f, stream);
}
int sobfu_hip_init_plane(float* d_vol, int X, int Y, int Z, const float vs[3], float trunc, float z, void* stream) {
    return launch_prim<PLANE>(d_vol, X, Y, Z, vs, trunc, 0.f, &z, 1, 0.f, stream);
}
int sobfu_hip_init_torus(float* d_vol, int X, int Y, int Z, const float vs[3], float trunc, const float t[2], void* stream) {
    SOBFU_CHECK_ARGS(t);
    return launch_prim<TORUS>(d_vol, X, Y, Z, vs, trunc, 0.f, t, 2, 0.f, stream);
}

int sobfu_hip_bilateral_filter(const uint16_t* d_src, int sstep, uint16_t* d_dst, int dstep, int rows, int cols, int ksz,
                               float sigma_spatial, float sigma_depth, void* stream) {
    SOBFU_CHECK_ARGS(d_src && d_dst && rows > 0 && cols > 0 && ksz > 0);
    sigma_depth *= 1000;  // metres -> mm (imgproc.cu:43)
    // (VALU-bound on the 49 correctly rounded expf of a pixel -- the reference's build uses the 2-instruction __expf; block shapes 64x1 .. 16x16
    //  all give 16 - 17 us at 640 x 480, measured in round 6)
    hipLaunchKernelGGL(bilateral_kernel, img_grid(rows, cols), dim3(64, 4), 0, (hipStream_t) stream, d_src, sstep, d_dst,
                       dstep, rows, cols, ksz, 0.5f / (sigma_spatial * sigma_spatial), 0.5f / (sigma_depth * sigma_depth));
    return (int) hipGetLastError();
}

int sobfu_hip_truncate_depth(uint16_t* d_depth, int step, int rows, int cols, float max_dist_m, void* stream) {
    SOBFU_CHECK_ARGS(d_depth && rows > 0 && cols > 0);
    hipLaunchKernelGGL(truncate_depth_kernel, img_grid(rows, cols), dim3(64, 4), 0, (hipStream_t) stream, d_depth, step, rows,
                       cols, (uint16_t) (max_dist_m * 1000.f));
    return (int) hipGetLastError();
}

int sobfu_hip_compute_dists(const uint16_t* d_depth, int dstep, float* d_dists, int sstep, int rows, int cols, float fx,
                            float fy, float cx, float cy, void* stream) {
    SOBFU_CHECK_ARGS(d_depth && d_dists && rows > 0 && cols > 0);
    hipLaunchKernelGGL(compute_dists_kernel, img_grid(rows, cols), dim3(64, 4), 0, (hipStream_t) stream, d_depth, dstep,
                       d_dists, sstep, rows, cols, 1.f / fx, 1.f / fy, cx, cy);
    return (int) hipGetLastError();
}

}  // extern "C"
