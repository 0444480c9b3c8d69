// Device-side arithmetic for the SobolevFusion hot path on gfx950 (CDNA4).
//
// Parity rules (reference include/sobfu/cuda/utils.hpp):
//  * every TU that includes this header is compiled with -ffp-contract=off, so `a*b+c` is never fused;
//    FMAs appear only where the reference spells fma/__fmaf_rn and are written `__builtin_fmaf` here;
//  * float4 operators return w = 0 (utils.hpp:245-275); `+=`/`-=` touch xyz only (utils.hpp:253-265);
//  * `/` and sqrtf are IEEE correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt),
//    fp32 denormals are preserved (gfx9 default) -- the same conventions as oracle/sobfu_oracle.c.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

namespace sobfu_hip {

#define SOBFU_DEV __device__ __forceinline__

struct Dims {
    int x, y, z;
};

SOBFU_DEV size_t vidx(const Dims& d, int x, int y, int z) { return (size_t) x + (size_t) d.x * ((size_t) y + (size_t) d.y * (size_t) z); }

// ---- float4 operators (utils.hpp:245-285) -------------------------------------------------------
SOBFU_DEV float4 f4(float x, float y, float z) { return make_float4(x, y, z, 0.f); }
SOBFU_DEV float4 add4(const float4& a, const float4& b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z); }
SOBFU_DEV float4 sub4(const float4& a, const float4& b) { return f4(a.x + (-b.x), a.y + (-b.y), a.z + (-b.z)); }
SOBFU_DEV float4 mul4(const float4& v, float m) { return f4(v.x * m, v.y * m, v.z * m); }
SOBFU_DEV float4 half4(const float4& v) { return f4(v.x / 2.f, v.y / 2.f, v.z / 2.f); }  // __fdividef(.,2.f): exact
SOBFU_DEV float norm_sq4(const float4& v) { return v.x * v.x + v.y * v.y + v.z * v.z; }

// __fsqrt_rd (utils.hpp:279-281): correctly rounded sqrt, stepped down when it rounded up.
SOBFU_DEV float sqrt_rd(float s) {
    float r = __builtin_sqrtf(s);
    // sign(fma(r, r, -s)) is the exact sign of r*r - s
    if (r > 0.f && __builtin_fmaf(r, r, -s) > 0.f) r = __uint_as_float(__float_as_uint(r) - 1u);
    return r;
}
SOBFU_DEV float norm4(const float4& v) { return sqrt_rd(norm_sq4(v)); }

// ---- lerp (utils.hpp:33-44): fma(t, v_upper, fma(-t, v_lower, v_lower)) -------------------------
SOBFU_DEV float lerp1(float v0, float v1, float t) { return __builtin_fmaf(t, v0, __builtin_fmaf(-t, v1, v1)); }
SOBFU_DEV float4 lerp4(const float4& a, const float4& b, float t) {
    return f4(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t));
}

// clamp / floor / upper-index rule of every trilinear sampler (utils.hpp:52-76)
struct Tri {
    int g, h;
    float t;
};
SOBFU_DEV Tri tri_setup(float p, int dim) {
    float top = (float) dim - 1;
    float cf  = fminf(fmaxf(0.f, p), top);
    Tri r;
    r.g = (int) floorf(cf);
    r.h = r.g + ((cf == 0.f || cf == top) ? 0 : 1);
    r.t = cf - (float) r.g;
    return r;
}

// interpolate_tsdf (utils.hpp:50-86)
SOBFU_DEV float2 interp_tsdf(const float2* __restrict__ v, const Dims& d, float px, float py, float pz) {
    Tri a = tri_setup(px, d.x), b = tri_setup(py, d.y), c = tri_setup(pz, d.z);
    const size_t sy = (size_t) d.x, sz = (size_t) d.x * d.y;
    const float2* pg = v + (size_t) b.g * sy + (size_t) c.g * sz;  // (.., gy, gz)
    const size_t dy = (size_t) (b.h - b.g) * sy, dz = (size_t) (c.h - c.g) * sz;
    float2 ggg = pg[a.g];
    float hhh = pg[a.h + dy + dz].x, hhg = pg[a.h + dy].x, hgh = pg[a.h + dz].x, hgg = pg[a.h].x;
    float ghh = pg[a.g + dy + dz].x, ghg = pg[a.g + dy].x, ggh = pg[a.g + dz].x;
    float t = lerp1(lerp1(lerp1(hhh, hhg, c.t), lerp1(hgh, hgg, c.t), b.t),
                    lerp1(lerp1(ghh, ghg, c.t), lerp1(ggh, ggg.x, c.t), b.t), a.t);
    return make_float2(t, ggg.y);
}

// VectorField::get_displacement (vector_fields.cu:24-26)
SOBFU_DEV float4 disp_at(const float4* __restrict__ psi, const Dims& d, int x, int y, int z) {
    return sub4(psi[vidx(d, x, y, z)], f4((float) x, (float) y, (float) z));
}

// interpolate_field_inv (utils.hpp:124-164)
SOBFU_DEV float4 interp_disp(const float4* __restrict__ psi, const Dims& d, float px, float py, float pz) {
    Tri a = tri_setup(px, d.x), b = tri_setup(py, d.y), c = tri_setup(pz, d.z);
    return lerp4(lerp4(lerp4(disp_at(psi, d, a.h, b.h, c.h), disp_at(psi, d, a.h, b.h, c.g), c.t),
                       lerp4(disp_at(psi, d, a.h, b.g, c.h), disp_at(psi, d, a.h, b.g, c.g), c.t), b.t),
                 lerp4(lerp4(disp_at(psi, d, a.g, b.h, c.h), disp_at(psi, d, a.g, b.h, c.g), c.t),
                       lerp4(disp_at(psi, d, a.g, b.g, c.h), disp_at(psi, d, a.g, b.g, c.g), c.t), b.t),
                 a.t);
}

// kfusion::device::dot (include/kfusion/cuda/temp_utils.hpp:33-35)
SOBFU_DEV float dot3(const float* a, float bx, float by, float bz) {
    return __builtin_fmaf(a[0], bx, __builtin_fmaf(a[1], by, a[2] * bz));
}

// {tsdf, weight} packing shared by integrate / init_* (tsdf_volume.cu:93-99)
SOBFU_DEV float2 pack_tsdf(float sdf, float trunc, float weight) {
    if (sdf >= trunc) return make_float2(1.f, weight);
    if (sdf <= -trunc) return make_float2(-1.f, weight);
    return make_float2(sdf / trunc, weight);
}

// ---- streaming (nontemporal) accesses of the launcher-for-launcher kernels (template parameter NT) -- for arrays a launch reads or writes
// exactly once, on grids whose arrays exceed the 256 MiB Infinity Cache (launcher_streams below).  Measured at 256^3 (round 6,
// profiles/r06/launcher_table_256.md): a plainly stored output allocates in the caches and the stencil's re-read taps lose them --
// laplacian 121 -> 87 us, TSDF gradient 105 -> 65, convolution_rows 105 -> 77 with nothing but the hint on the output store.
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <bool NT>
SOBFU_DEV void st4(float4* p, const float4& v) {
    if (NT) __builtin_nontemporal_store((v4f_t){v.x, v.y, v.z, v.w}, (v4f_t*) p);
    else *p = v;
}
template <bool NT>
SOBFU_DEV void st2(float2* p, const float2& v) {
    if (NT) __builtin_nontemporal_store((v2f_t){v.x, v.y}, (v2f_t*) p);
    else *p = v;
}
template <bool NT>
SOBFU_DEV float4 ld4(const float4* p) {
    if (!NT) return *p;
    const v4f_t t = __builtin_nontemporal_load((const v4f_t*) p);
    return make_float4(t.x, t.y, t.z, t.w);
}
template <bool NT>
SOBFU_DEV float2 ld2(const float2* p) {
    if (!NT) return *p;
    const v2f_t t = __builtin_nontemporal_load((const v2f_t*) p);
    return make_float2(t.x, t.y);
}
// grids of more cells than this stream (a float4 field of 3.3 M cells is 53 MB: a launcher chain's few arrays still fit the Infinity Cache)
// SOBFU_LAUNCHER_NT=0 / 1 overrides (measurement: tests/reference_launcher_table.py).
inline bool launcher_streams(int X, int Y, int Z) {
    if (const char* e = getenv("SOBFU_LAUNCHER_NT")) return e[0] == '1';
    return (long) X * Y * Z > 3300000L;
}

// ---- launch geometry ---------------------------------------------------------------------------------
// Per-voxel kernels: one wave spans 64 consecutive x (1 KiB float4 / 512 B float2 coalesced segments),
// 4 rows per workgroup, one z-slice per blockIdx.z -> >= 16k workgroups at 256^3 (vs. the reference's
// 64 workgroups of block(64,16) looping over z).
constexpr int kBX = 64, kBY = 4;
inline dim3 voxel_block() { return dim3(kBX, kBY, 1); }
inline dim3 voxel_grid(int X, int Y, int Z) { return dim3((X + kBX - 1) / kBX, (Y + kBY - 1) / kBY, Z); }

// XCD-aware workgroup -> tile map of those kernels.  The hardware deals consecutive workgroups (x fastest) round-robin to the 8 XCDs, each
// with its own 4 MB L2.  Numbered as launched, a tile's +-y neighbours (the next workgroup but gx) sit on other XCDs, so a y stencil's rows
// are fetched into several L2s.  Here XCD k owns the k-th eighth of the y range (a band of gy / 8 tile rows) at EVERY z: +-y neighbours
// share its L2 except at the seven band seams, and +-z neighbours (the same tile of the next plane) stay on it as well -- a band of seven
// float4 planes is 0.9 MB at 256^2.  (Whole z ranges per XCD were measured too: a 7-plane window of whole planes does not fit the L2 --
// convolution_depth 183 -> 343 us.)  Grids whose tile-row count is not a multiple of 8 keep the launch order.  For kernels whose work per
// voxel is uniform only: see VOXEL_XYZ_XCD in field_kernels.hip.
SOBFU_DEV void xcd_tile(int& bx, int& by, int& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    if ((gy & 7u) != 0u) {
        bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
        return;
    }
    const unsigned w = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), k = w & 7u, j = w >> 3;
    const unsigned band = gy >> 3, per = gx * band;  // tile rows per band, tiles of a band in one plane
    const unsigned z = j / per, rem = j - z * per, row = rem / gx;
    bx = (int) (rem - row * gx);
    by = (int) (k * band + row);
    bz = (int) z;
}

// Marching launchers (a lane owns an (x, y) column of a z chunk; grid = voxel_grid(X, Y, chunks)): planes per chunk -- >= 512 workgroups on
// the chip when the grid allows it (two per CU: long chunks beat occupancy, measured), never chunks shorter than 8 planes (a chunk re-reads
// its z halo when it starts)
inline int march_zc(int X, int Y, int Z) {
    const long tiles = (long) ((X + kBX - 1) / kBX) * ((Y + kBY - 1) / kBY);
    const long chunks = 512 / tiles > 1 ? 512 / tiles : 1;
    const long zc = (Z + chunks - 1) / chunks;
    return (int) (zc > 8 ? zc : 8);
}

}  // namespace sobfu_hip
