// EXPERIMENT (not part of libsobfu_hip.so): ONE launch per solver iteration in which nabla_U never leaves the CU.
//
// VERDICT round 4, item 1 asks for a go / no-go on this decomposition in the regime of a multi-GPU tile (128^3, cache-resident,
// latency- / launch-bound): round 1 built it for 256^3, lost to the two passes there (277 vs 259 us, VALU-bound) and removed it in
// round 2; this file is that kernel again (same cell bookkeeping), brought up to the current tree's idioms -- packed fp32 taps,
// 32-bit gather offsets, XCD-aware tile map, an even z-chunk split chosen by the caller -- so that tools/fused_go_nogo.py can time
// it beside today's two-pass loop on the same grid in the same process and compare the bits.
//
// A workgroup owns a 64 x 8 xy tile and marches a z-chunk.  Per step z it
//   (3) produces nabla_U of plane p = z + 3 on the tile PLUS a 3-cell ring (E3 = 70 x 14 cells: 1.9x pass A's arithmetic) from
//       psi / F = (phi_n o psi).tsdf -- plane p and p + 1 in registers, plane p - 1 from the previous step's LDS buffer, in-plane
//       neighbours from this step's buffer (pass A's arithmetic, op for op: potential_gradient_cell of solver_kernels.hip);
//   (4) smooths / updates / warps plane z of the tile exactly like pass B: x / y taps from an LDS nabla_U tile (ring cells arrive
//       through a 4-plane delay line), z taps from a 7-plane register pipeline.
// psi and F are ping-ponged between two arrays (a tile's ring must see the previous iteration's values while neighbours already
// write the next).  Per voxel-iteration: reads psi 12 + F 4 + G 4 (+ ring) + gather 4, writes psi 12 + F 4 = 40 B (two passes: 76).
// A march of n planes runs n + 6 steps (the first six only produce), which is what short marches pay for.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sobfu_device.hpp"

using namespace sobfu_hip;

namespace {
constexpr int TX = 64, FY = 8, FE3X = TX + 6, FE3Y = FY + 6;
constexpr int FEXTRA = FE3X * FE3Y - TX * FY;  // 468 ring cells, one per lane of the first 468

struct Taps {
    float s[7];
};
struct FusedArgs {
    const float* psi_in;  // 12-byte cells
    const float* f_in;    // (phi_n o psi).tsdf
    const float* g;       // phi_global.tsdf
    const float* phi_n;   // phi_n.tsdf
    float* psi_out;
    float* f_out;
    uint32_t* slots;
    Dims d;
    Taps S;
    float alpha, w_reg;
    int zc, rem;  // z-chunks: the first `rem` march zc + 1 planes
    int nt;       // streaming stores
};
typedef float v3f __attribute__((ext_vector_type(3)));
typedef v3f __attribute__((aligned(4))) v3f_u;
typedef float v2f __attribute__((ext_vector_type(2)));

SOBFU_DEV unsigned xcd_swizzle(unsigned t, unsigned nb) {
    const unsigned q = nb / 8u, rem = nb % 8u, xcd = t % 8u, slot = t / 8u;
    return xcd * q + min(xcd, rem) + slot;
}
// interpolate_tsdf on a tsdf-only volume with 32-bit byte offsets (solver_kernels.hip: interp_tsdf_only32)
SOBFU_DEV float interp32(const float* __restrict__ v, const Dims& d, float px, float py, float pz) {
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
// pass A's arithmetic for one cell (vector_fields.cu:165-191, 299-331; solver.cu:28-31).  c = {psi.xyz, F} of the cell; xp / xm /
// yp / ym in-plane neighbours, zp / zm the planes above / below.
SOBFU_DEV float4 nabla_u_cell(const float4& c, float4 xp, float4 xm, float4 yp, float4 ym, float4 zp, float4 zm, float g, float w_reg, bool xlo,
                              bool xhi, bool ylo, bool yhi, bool zlo, bool zhi) {
    const float gx1 = xhi ? xm.w : xp.w, gx2 = xlo ? xp.w : xm.w;
    const float gy1 = yhi ? ym.w : yp.w, gy2 = ylo ? yp.w : ym.w;
    const float gz1 = zhi ? zm.w : zp.w, gz2 = zlo ? zp.w : zm.w;
    const float4 gr = f4((gx1 - gx2) / 2.f, (gy1 - gy2) / 2.f, (gz1 - gz2) / 2.f);
    if (xlo || xhi) { xp = c; xm = c; }
    if (ylo || yhi) { yp = c; ym = c; }
    if (zlo || zhi) { zp = c; zm = c; }
    float4 v = mul4(c, -6.f);
    v = add4(v, xp);
    v = add4(v, xm);
    v = add4(v, yp);
    v = add4(v, ym);
    v = add4(v, zp);
    v = add4(v, zm);
    const float4 L = mul4(v, -1.f);
    return add4(mul4(gr, c.w - g), mul4(L, w_reg));
}

__global__ void __launch_bounds__(TX* FY, 4) fused_iteration_kernel(FusedArgs a) {
    __shared__ float4 t_pf[3][FE3Y][FE3X];  // {psi.xyz, F} of plane p on E3, addressed by CLAMPED cell position (three buffers: a step
                                            // reads plane p and plane p - 1, the next step writes a third; one barrier per step)
    __shared__ float4 t_nu[2][FE3Y][FE3X];  // nabla_U of plane z on E3, addressed by RAW cell position
    __shared__ uint32_t s_max[FY];

    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y, tid = wy * TX + lx;
    const unsigned ntx = (unsigned) ((d.x + TX - 1) / TX), nty = (unsigned) ((d.y + FY - 1) / FY);
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int x0 = (int) (t % ntx) * TX, y0 = (int) ((t / ntx) % nty) * FY, ck = (int) (t / (ntx * nty));
    const int zb = ck * a.zc + min(ck, a.rem), ze = min(zb + a.zc + (ck < a.rem ? 1 : 0), d.z);
    const uint32_t plane = (uint32_t) d.x * d.y;
    auto clampx = [&](int v) { return min(max(v, 0), d.x - 1); };
    auto clampy = [&](int v) { return min(max(v, 0), d.y - 1); };
    auto clampz = [&](int v) { return (uint32_t) min(max(v, 0), d.z - 1); };

    // main cell
    const int x = x0 + lx, y = y0 + wy;
    const int mgx = clampx(x), mgy = clampy(y);
    const uint32_t m_off = (uint32_t) mgx + (uint32_t) d.x * mgy;
    const int mcx = mgx - (x0 - 3), mcy = mgy - (y0 - 3);
    const bool m_in = x < d.x && y < d.y;
    // ring cell (E3 \ tile)
    const bool has_e = tid < FEXTRA;
    int ex = 0, ey = 0;
    if (tid < 210) { ey = tid / FE3X; ex = tid % FE3X; }
    else if (tid < 420) { ey = 11 + (tid - 210) / FE3X; ex = (tid - 210) % FE3X; }
    else { const int e = tid - 420, c6 = e % 6; ey = 3 + e / 6; ex = c6 < 3 ? c6 : TX + c6; }
    const int egx = clampx(x0 - 3 + ex), egy = clampy(y0 - 3 + ey);
    const uint32_t e_off = (uint32_t) egx + (uint32_t) d.x * egy;
    const int ecx = egx - (x0 - 3), ecy = egy - (y0 - 3);
    // the one neighbour of an E3-perimeter cell outside E3 (corner cells of E3 are never read by the plus-shaped convolution)
    int o_dir = 0;
    if (has_e) {
        if (ex == 0) o_dir = 1; else if (ex == FE3X - 1) o_dir = 2; else if (ey == 0) o_dir = 3; else if (ey == FE3Y - 1) o_dir = 4;
    }
    const uint32_t o_off = (uint32_t) clampx(egx + (o_dir == 1 ? -1 : o_dir == 2 ? 1 : 0)) + (uint32_t) d.x * clampy(egy + (o_dir == 3 ? -1 : o_dir == 4 ? 1 : 0));

    auto ld_pf = [&](uint32_t off, int zz) -> float4 {
        const uint32_t i = clampz(zz) * plane + off;
        const v3f v = *(const v3f_u*) (a.psi_in + 3 * (size_t) i);
        return make_float4(v.x, v.y, v.z, a.f_in[i]);
    };

    const int z_start = max(zb - 6, -3);
    float4 mc_, mn_, ec_ = f4(0.f, 0.f, 0.f), en_ = f4(0.f, 0.f, 0.f), on_ = f4(0.f, 0.f, 0.f);
    {
        const int p = z_start + 3;
        mc_ = ld_pf(m_off, p);
        mn_ = ld_pf(m_off, p + 1);
        if (has_e) { ec_ = ld_pf(e_off, p); en_ = ld_pf(e_off, p + 1); }
        if (o_dir) on_ = ld_pf(o_off, p);
        t_pf[2][mcy][mcx] = ld_pf(m_off, p - 1);
        if (has_e) t_pf[2][ecy][ecx] = ld_pf(e_off, p - 1);
    }
    float gm = a.g[clampz(z_start + 3) * plane + m_off], ge = has_e ? a.g[clampz(z_start + 3) * plane + e_off] : 0.f;
    float4 q[7], dl[4];
#pragma unroll
    for (int k = 0; k < 7; ++k) q[k] = f4(0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) dl[k] = f4(0.f, 0.f, 0.f);
    const bool mxlo = mgx == 0, mxhi = mgx == d.x - 1, mylo = mgy == 0, myhi = mgy == d.y - 1;
    const bool exlo = egx == 0, exhi = egx == d.x - 1, eylo = egy == 0, eyhi = egy == d.y - 1;

    float msq = 0.f;
    for (int z = z_start; z < ze; ++z) {
        const int buf = (z - z_start) & 1, pb = (z - z_start) % 3, pbm = (pb + 2) % 3, p = z + 3;
        t_pf[pb][mcy][mcx] = mc_;
        if (has_e) t_pf[pb][ecy][ecx] = ec_;
        t_nu[buf][wy + 3][lx + 3] = q[3];
        if (has_e) t_nu[buf][ey][ex] = dl[0];
        const float4 nmn = ld_pf(m_off, p + 2);
        float4 nen = f4(0.f, 0.f, 0.f), non = f4(0.f, 0.f, 0.f);
        if (has_e) nen = ld_pf(e_off, p + 2);
        if (o_dir) non = ld_pf(o_off, p + 1);
        const float ngm = a.g[clampz(p + 1) * plane + m_off], nge = has_e ? a.g[clampz(p + 1) * plane + e_off] : 0.f;
        float4 pin = f4(0.f, 0.f, 0.f);
        if (z >= zb) {
            const v3f v = *(const v3f_u*) (a.psi_in + 3 * (size_t) ((uint32_t) z * plane + m_off));
            pin = make_float4(v.x, v.y, v.z, 0.f);
        }
        __syncthreads();

        if (p >= 0 && p < d.z) {
            const bool zlo = p == 0, zhi = p == d.z - 1;
            q[6] = nabla_u_cell(mc_, t_pf[pb][mcy][mcx + 1], t_pf[pb][mcy][mcx - 1], t_pf[pb][mcy + 1][mcx], t_pf[pb][mcy - 1][mcx], mn_,
                                t_pf[pbm][mcy][mcx], gm, a.w_reg, mxlo, mxhi, mylo, myhi, zlo, zhi);
            if (has_e) {
                const float4 xp = o_dir == 2 ? on_ : t_pf[pb][ecy][min(ecx + 1, FE3X - 1)];
                const float4 xm = o_dir == 1 ? on_ : t_pf[pb][ecy][max(ecx - 1, 0)];
                const float4 yp = o_dir == 4 ? on_ : t_pf[pb][min(ecy + 1, FE3Y - 1)][ecx];
                const float4 ym = o_dir == 3 ? on_ : t_pf[pb][max(ecy - 1, 0)][ecx];
                dl[3] = nabla_u_cell(ec_, xp, xm, yp, ym, en_, t_pf[pbm][ecy][ecx], ge, a.w_reg, exlo, exhi, eylo, eyhi, zlo, zhi);
            }
            if (p == 0) {
#pragma unroll
                for (int k = 3; k < 6; ++k) q[k] = q[6];
#pragma unroll
                for (int k = 0; k < 3; ++k) dl[k] = dl[3];
            }
        } else if (p >= d.z) {
            q[6]  = q[5];
            dl[3] = dl[2];
        }

        if (z >= zb) {
            v2f l01 = {0.f, 0.f}, l23 = {0.f, 0.f}, r01 = {0.f, 0.f}, r23 = {0.f, 0.f}, z01 = {0.f, 0.f}, z23 = {0.f, 0.f};
#pragma unroll
            for (int j = -3; j <= 3; ++j) {
                const v2f s2 = {a.S.s[3 - j], a.S.s[3 - j]};
                const float4 vl = (j == 0) ? q[3] : t_nu[buf][wy + 3][lx + 3 + j];
                l01 += v2f{vl.x, vl.y} * s2;
                l23 += v2f{vl.z, vl.w} * s2;
                const float4 vr = (j == 0) ? q[3] : t_nu[buf][wy + 3 + j][lx + 3];
                r01 += v2f{vr.x, vr.y} * s2;
                r23 += v2f{vr.z, vr.w} * s2;
                const float4 vz = q[3 + j];
                z01 += v2f{vz.x, vz.y} * s2;
                z23 += v2f{vz.z, vz.w} * s2;
            }
            const v2f t01 = (l01 + r01) + z01;
            const float tx = t01.x, ty = t01.y, tz = (l23.x + r23.x) + z23.x;
            const float4 u = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
            float4 pnew = pin;
            pnew.x -= u.x;
            pnew.y -= u.y;
            pnew.z -= u.z;
            if (m_in) {
                msq = fmaxf(msq, norm_sq4(u));
                const uint32_t i = (uint32_t) z * plane + (uint32_t) x + (uint32_t) d.x * y;
                const v3f o = {pnew.x, pnew.y, pnew.z};
                const float f = interp32(a.phi_n, d, pnew.x, pnew.y, pnew.z);
                if (a.nt) {
                    __builtin_nontemporal_store(o, (v3f_u*) (a.psi_out + 3 * (size_t) i));
                    __builtin_nontemporal_store(f, a.f_out + i);
                } else {
                    *(v3f_u*) (a.psi_out + 3 * (size_t) i) = o;
                    a.f_out[i] = f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            q[k] = q[k + 1];
            asm volatile("" : "+v"(q[k].x), "+v"(q[k].y), "+v"(q[k].z));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dl[k] = dl[k + 1];
        mc_ = mn_; mn_ = nmn;
        ec_ = en_; en_ = nen;
        on_ = non;
        gm = ngm;
        ge = nge;
    }

    uint32_t m = __float_as_uint(msq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
    if (lx == 0) s_max[wy] = m;
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < FY; ++w) m = max(m, s_max[w]);
        atomicMax(a.slots + (blockIdx.x & 255u), m);
    }
}
}  // namespace

// one iteration; nch z-chunks (even split).  Returns the hipError of the launch.
extern "C" __attribute__((visibility("default"))) int calib_fused_iteration(const float* psi_in3, const float* f_in, const float* g, const float* phi_n1,
                                                                            float* psi_out3, float* f_out, uint32_t* slots, const float taps[7],
                                                                            float alpha, float w_reg, int X, int Y, int Z, int nch, int nt, void* stream) {
    if (nch < 1) nch = 1;
    if (nch > Z) nch = Z;
    FusedArgs a{psi_in3, f_in, g, phi_n1, psi_out3, f_out, slots, {X, Y, Z}, {}, alpha, w_reg, Z / nch, Z % nch, nt};
    for (int i = 0; i < 7; ++i) a.S.s[i] = taps[i];
    const dim3 grid((unsigned) (((X + TX - 1) / TX) * ((Y + FY - 1) / FY) * nch));
    hipLaunchKernelGGL(fused_iteration_kernel, grid, dim3(TX, FY), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

// n iterations, ping-ponging between the two halves (half k & 1 is iteration k's input); enqueued from C so that the host side of the
// comparison is the same as the solver handle's
extern "C" __attribute__((visibility("default"))) int calib_fused_iterate(float* psi3[2], float* f[2], const float* g, const float* phi_n1, uint32_t* slots,
                                                                          const float taps[7], float alpha, float w_reg, int X, int Y, int Z, int nch,
                                                                          int nt, int n, void* stream) {
    for (int k = 0; k < n; ++k) {
        const int a = k & 1, b = a ^ 1;
        const int rc = calib_fused_iteration(psi3[a], f[a], g, phi_n1, psi3[b], f[b], slots, taps, alpha, w_reg, X, Y, Z, nch, nt, stream);
        if (rc != 0) return rc;
    }
    return 0;
}
