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

// ---- launch geometry ---------------------------------------------------------------------------------
// Per-voxel kernels: one wave spans 64 consecutive x (1 KiB float4 / 512 B float2 coalesced segments),
// 4 rows per workgroup, one z-slice per blockIdx.z -> >= 16k workgroups at 256^3 (vs. the reference's
// 64 workgroups of block(64,16) looping over z).
constexpr int kBX = 64, kBY = 4;
inline dim3 voxel_block() { return dim3(kBX, kBY, 1); }
inline dim3 voxel_grid(int X, int Y, int Z) { return dim3((X + kBX - 1) / kBX, (Y + kBY - 1) / kBY, Z); }

}  // namespace sobfu_hip
