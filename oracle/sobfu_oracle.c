/*
 * sobfu_oracle.c -- CPU restatement of the SobolevFusion solver hot path of dgrzech/sobfu.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: it may be imported / linked / executed
 * only by tests/, __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py -- and there only
 * as the checker, never as the thing measured or shipped.  The product path (sobfu_amd/) never
 * calls into it and fails loudly if the HIP library is missing.
 *
 * Every function cites the reference file:line (relative to the reference repo root) it restates.
 * The reference is CUDA-only (CMakeLists.txt:31) and cannot be built in this image (no nvcc, no CUDA
 * runtime, no OpenCV/PCL/Boost): it is "unbuildable here", there is no oracle/_ref.
 *
 * Pinning: this restatement is checked (tests/test_oracle_pins.py, tests/test_reference_tables.py) against
 *   (1) the six value-pinning gtest cases of the reference (test/deformation_field_test.cpp:92-336,
 *       test/reductions_test.cpp:86-101) -- identity init, TSDF gradient, Jacobian, Laplacian, data energy;
 *   (2) reference-held constant tables committed as fixture data (tests/golden/reference_tables.json): every
 *       raw tap of the Sobolev filter table (src/sobfu/solver.cpp:160-251), numVertsTable and a hash of
 *       triTable (src/kfusion/marching_cubes.cpp:81-358);
 *   (3) the known-answer values recorded in SURVEY.md Appendix B (energies, max update norms and
 *       warp-field statistics that the reference's own code printed for test/solver_test.cpp:109-132
 *       and for a two-frame depth pipeline; convolution impulse responses).
 * PARITY UNPINNED FOR THE HOT PATH by reference-held vectors: the reference's solver tests assert nothing
 * (test/solver_test.cpp:109-208), so convolution / update / apply / loop / inverse / fusion are tied to the
 * reference only by (3) and by (4) below -- arrays produced by running the reference's own sources under a
 * host-emulation shim, which is evidence, not a vector the reference holds (a build through stand-in headers
 * pins nothing by the task's rules: the parity grade is "partial").  Marching cubes: (2) for its tables, (4).
 *   (4) tests/golden/ref_*.npz (tests/test_reference_fixtures.py): full arrays of every launcher, of whole
 *       Solver::estimate_psi runs, of SobFusion::operator() frame by frame and of MarchingCubes::run, computed by
 *       the reference's source lines under tools/ref_emulation/ and regenerable byte-identically in the build
 *       container (tests/golden/make_reference_fixtures.py, which also re-derives the numbers of (3)).
 *
 * Arithmetic conventions: IEEE-754 binary32, round-to-nearest, no FTZ,
 * NO floating-point contraction (build with -ffp-contract=off); FMAs only where the reference spells
 * fma / __fmaf_rn.  float4.w is 0 wherever a float4 operator of include/sobfu/cuda/utils.hpp:245-275
 * produced the value.  CUDA fast-math approximations of the reference build (--prec-div=false,
 * --prec-sqrt=false, __expf, powf) are evaluated here with IEEE `/`, sqrtf and libm.
 *
 * Layout: dense, x fastest: idx = x + X*(y + Y*z) (src/sobfu/cuda/vector_fields.cu:20-22).
 * TSDF voxel = float2 {tsdf, weight}; vector field voxel = float4; Jacobian voxel = 4 x float4.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

/* Negative controls for the known-answer tests (tests/test_oracle_mutants.py): -DSO_MUTANT=k builds this restatement with ONE
 * deliberate deviation from the reference on the hot path; the closed-form suite and the reference's own gtest cases must fail on
 * every one of them.  0 (the default, the only build anything else uses) is the restatement itself.
 *   1 the three 1-D passes COMPOSED instead of summed        2 tap S[3+j] instead of S[3-j]      3 lerp operands swapped
 *   4 upper index g+1 also at coordinate exactly 0           5 psi += u                            6 Laplacian sign
 *   7 gradient clamps at a face instead of mirroring          8 sqrtf for __fsqrt_rd               9 (phi_global - phi_n o psi)
 *  10 zero padding instead of clamp-to-edge                  11 weight from the upper corner       12 Laplacian mirrors at a face */
#ifndef SO_MUTANT
#define SO_MUTANT 0
#endif

/* -DSO_NVCC_MODE=1: the same restatement evaluated the way the reference AS BUILT may evaluate it.  The reference compiles with
 * nvcc --ftz=true --prec-div=false --prec-sqrt=false and the default --fmad=true (CMakeLists.txt:40-46); this file and the HIP
 * kernels evaluate IEEE.  The mode exists to put a NUMBER on that distance (tests/test_nvcc_distance.py):
 *   - flush-to-zero of binary32 subnormal operands and results: the caller switches the CPU's MXCSR FTZ|DAZ bits on in every
 *     OpenMP thread around the calls (so_nvcc_ftz, below) -- the same semantics as PTX's .ftz, for every operation of the file;
 *   - every device-side `/` and __fdividef (div.approx.ftz.f32, <= 2 ulp): the correctly rounded quotient moved by k ulp,
 *     k in {-2..2} drawn from a hash of the operands and a seed; exact when the divisor is a power of two (x / 2.f);
 *   - device-side sqrtf (sqrt.approx.ftz.f32, ~1 ulp): k in {-1..1};  powf (<= 4 ulp, CUDA C Programming Guide, table of
 *     single-precision function errors): k in {-4..4};  __expf (2 + floor(|1.16 x|) ulp): k in that range;
 *   - plain `a * b + c` in device code contracted into one fma where the product is not an __fmul_rn intrinsic
 *     (tsdf_volume.cu:70-71,189-196,256-257; imgproc.cu:28-34,238-240; reductor.cu:26-31).  Nothing in the per-iteration
 *     recurrence is affected: it is written in rn intrinsics and explicit fma throughout.
 * The draws are a deterministic function of (operands, seed): a randomised within-spec perturbation, not an adversarial bound.
 * Host-side arithmetic of the reference (1.f / f.x in compute_dists' wrapper, 0.5f / sigma^2 in bilateralFilter's) stays IEEE.
 * 0 (the default) is the oracle everything else uses; only tests/test_nvcc_distance.py builds and loads the other. */
#ifndef SO_NVCC_MODE
#define SO_NVCC_MODE 0
#endif
#if SO_NVCC_MODE
#include <xmmintrin.h>
static uint32_t so_nvcc_seed = 1u, so_nvcc_mask = 31u;
void so_nvcc_set_seed(unsigned s) { so_nvcc_seed = s; }
/* which approximations are on (for attribution): 1 divide, 2 sqrtf, 4 powf, 8 __expf, 16 fmad contraction; default all */
void so_nvcc_set_mask(unsigned m) { so_nvcc_mask = m; }
/* FTZ|DAZ on (1) / off (0) in the calling thread and every thread of the OpenMP pool; returns the caller's previous state */
int so_nvcc_ftz(int on) {
    int was = (_mm_getcsr() & 0x8040u) == 0x8040u;
#pragma omp parallel
    { _mm_setcsr(on ? (_mm_getcsr() | 0x8040u) : (_mm_getcsr() & ~0x8040u)); }
    _mm_setcsr(on ? (_mm_getcsr() | 0x8040u) : (_mm_getcsr() & ~0x8040u));
    return was;
}
static inline uint32_t nv_bits(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
static inline uint32_t nv_hash(uint32_t a, uint32_t b) {
    uint32_t h = (a ^ so_nvcc_seed) * 0x9E3779B1u;
    h ^= h >> 15; h = (h ^ b) * 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16;
    return h;
}
static inline float nv_step(float v, int k) { /* k ulp away from (k > 0) / toward (k < 0) zero */
    uint32_t u = nv_bits(v), mag = u & 0x7FFFFFFFu;
    if (mag == 0 || mag >= 0x7F800000u || k == 0) return v;
    int64_t m = (int64_t) mag + k;
    if (m < 0) m = 0;
    if (m > 0x7F7FFFFF) m = 0x7F7FFFFF;
    u = (u & 0x80000000u) | (uint32_t) m;
    memcpy(&v, &u, 4);
    return v;
}
static inline int nv_draw(float a, float b, int range) { return (int) (nv_hash(nv_bits(a), nv_bits(b)) % (uint32_t) (2 * range + 1)) - range; }
static inline float so_div(float a, float b) {
    float q = a / b;
    if (!(so_nvcc_mask & 1u) || (nv_bits(b) & 0x007FFFFFu) == 0) return q; /* power of two: exact */
    return nv_step(q, nv_draw(a, b, 2));
}
static inline float so_sqrt(float a) { return (so_nvcc_mask & 2u) ? nv_step(sqrtf(a), nv_draw(a, 0.5f, 1)) : sqrtf(a); }
static inline float so_pow(float a, float b) { return (so_nvcc_mask & 4u) ? nv_step(powf(a, b), nv_draw(a, b, 4)) : powf(a, b); }
static inline float so_exp_fast(float x) {
    return (so_nvcc_mask & 8u) ? nv_step(expf(x), nv_draw(x, 2.f, 2 + (int) floorf(fabsf(1.16f * x)))) : expf(x);
}
static inline float so_mad(float a, float b, float c) {
    if (so_nvcc_mask & 16u) return fmaf(a, b, c);
    volatile float p = a * b; /* keep the product rounded */
    return p + c;
}
#define SO_MAD(a, b, c) so_mad((a), (b), (c))
#else
#define so_div(a, b) ((a) / (b))
#define so_sqrt(a) sqrtf(a)
#define so_pow(a, b) powf((a), (b))
#define so_exp_fast(x) expf(x)
#define SO_MAD(a, b, c) ((a) * (b) + (c))
#endif

typedef struct { float x, y; } f2;
typedef struct { float x, y, z, w; } f4;
typedef struct { f4 r[4]; } m4;

#define IDX(x, y, z) ((size_t)(x) + (size_t)X * ((size_t)(y) + (size_t)Y * (size_t)(z)))

/* ------------------------------------------------------------------------------------------------
 * float4 operators -- include/sobfu/cuda/utils.hpp:245-285
 * ---------------------------------------------------------------------------------------------- */
static inline f4 mk4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
static inline f4 add4(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, 0.f); }          /* :245 */
static inline f4 sub4(f4 a, f4 b) { return mk4(a.x + (-b.x), a.y + (-b.y), a.z + (-b.z), 0.f); } /* :249 */
static inline f4 mul4(f4 v, float m) { return mk4(v.x * m, v.y * m, v.z * m, 0.f); }              /* :267 */
static inline f4 div4(f4 v, float d) { return mk4(so_div(v.x, d), so_div(v.y, d), so_div(v.z, d), 0.f); } /* :273 (__fdividef) */
static inline float norm_sq4(f4 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }                  /* :283 */

/* __fsqrt_rd: sqrt rounded toward -inf (utils.hpp:279-281 `norm`) */
static inline float sqrt_rd(float s) {
    float r = sqrtf(s);
#if SO_MUTANT != 8
    if (r > 0.f && (double) r * (double) r > (double) s) r = nextafterf(r, -INFINITY);
#endif
    return r;
}
static inline float norm4(f4 v) { return sqrt_rd(norm_sq4(v)); }

/* lerp -- utils.hpp:33-36 : fma(t, v0, fma(-t, v1, v1)), v0 = UPPER sample */
#if SO_MUTANT == 3
static inline float lerp1(float v0, float v1, float t) { return fmaf(t, v1, fmaf(-t, v0, v0)); }
#else
static inline float lerp1(float v0, float v1, float t) { return fmaf(t, v0, fmaf(-t, v1, v1)); }
#endif
static inline f4 lerp4(f4 a, f4 b, float t) {                                                    /* :42-44 */
    return mk4(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t), 0.f);
}

/* clamp + floor + upper-index rule shared by all trilinear samplers -- utils.hpp:52-76 */
static inline void tri_setup(float p, int dim, int *g, int *h, float *frac) {
    float cf = fminf(fmaxf(0.f, p), (float) dim - 1);
    int gi   = (int) floorf(cf);
    int hi   = gi + 1;
#if SO_MUTANT == 4
    if (cf == (float) dim - 1) hi--; /* (the top end stays inside the array) */
#else
    if (cf == 0.f || cf == (float) dim - 1) hi--;
#endif
    *g    = gi;
    *h    = hi;
    *frac = cf - (float) gi;
}

/* interpolate_tsdf -- utils.hpp:50-86 */
static inline f2 interp_tsdf(const f2 *v, int X, int Y, int Z, float px, float py, float pz) {
    int gx, hx, gy, hy, gz, hz;
    float a, b, c;
    tri_setup(px, X, &gx, &hx, &a);
    tri_setup(py, Y, &gy, &hy, &b);
    tri_setup(pz, Z, &gz, &hz, &c);
    float t = lerp1(lerp1(lerp1(v[IDX(hx, hy, hz)].x, v[IDX(hx, hy, gz)].x, c),
                          lerp1(v[IDX(hx, gy, hz)].x, v[IDX(hx, gy, gz)].x, c), b),
                    lerp1(lerp1(v[IDX(gx, hy, hz)].x, v[IDX(gx, hy, gz)].x, c),
                          lerp1(v[IDX(gx, gy, hz)].x, v[IDX(gx, gy, gz)].x, c), b),
                    a);
#if SO_MUTANT == 11
    f2 r = {t, v[IDX(hx, hy, hz)].y};
#else
    f2 r = {t, v[IDX(gx, gy, gz)].y};
#endif
    return r;
}

/* VectorField::get_displacement -- src/sobfu/cuda/vector_fields.cu:24-26 */
static inline f4 disp(const f4 *psi, int X, int Y, int x, int y, int z) {
    return sub4(psi[IDX(x, y, z)], mk4((float) x, (float) y, (float) z, 0.f));
}

/* interpolate_field_inv -- utils.hpp:124-164 */
static inline f4 interp_disp(const f4 *psi, int X, int Y, int Z, float px, float py, float pz) {
    int gx, hx, gy, hy, gz, hz;
    float a, b, c;
    tri_setup(px, X, &gx, &hx, &a);
    tri_setup(py, Y, &gy, &hy, &b);
    tri_setup(pz, Z, &gz, &hz, &c);
    return lerp4(lerp4(lerp4(disp(psi, X, Y, hx, hy, hz), disp(psi, X, Y, hx, hy, gz), c),
                       lerp4(disp(psi, X, Y, hx, gy, hz), disp(psi, X, Y, hx, gy, gz), c), b),
                 lerp4(lerp4(disp(psi, X, Y, gx, hy, hz), disp(psi, X, Y, gx, hy, gz), c),
                       lerp4(disp(psi, X, Y, gx, gy, hz), disp(psi, X, Y, gx, gy, gz), c), b),
                 a);
}

/* ================================================================================================
 * TSDF volume -- src/kfusion/cuda/tsdf_volume.cu
 * ============================================================================================== */

/* clear_volume_kernel -- tsdf_volume.cu:23-46 */
void so_clear_volume(f2 *vol, int X, int Y, int Z) { memset(vol, 0, sizeof(f2) * (size_t) X * Y * Z); }

static inline f2 pack_tsdf(float sdf, float trunc, float weight) { /* tsdf_volume.cu:93-99, 267-273 */
    f2 r;
    r.y = weight;
    if (sdf >= trunc) r.x = 1.f;
    else if (sdf <= -trunc) r.x = -1.f;
    else r.x = so_div(sdf, trunc); /* __fdividef */
    return r;
}

/* kfusion::device::dot -- include/kfusion/cuda/temp_utils.hpp:33-35 */
static inline float dot3(const float *a, float bx, float by, float bz) { return fmaf(a[0], bx, fmaf(a[1], by, a[2] * bz)); }

/* TsdfIntegrator::operator()(TsdfVolume&) -- tsdf_volume.cu:62-101 ; Projector device.hpp:36-41 ;
 * Aff3f*Vec3f device.hpp:57-61.  dists: pitched float image (step in BYTES), point-sampled. */
void so_integrate_depth(const float *dists, int step_bytes, int rows, int cols, f2 *vol, int X, int Y, int Z,
                        float vsx, float vsy, float vsz, float trunc, float eta, const float *R /*9, row major*/,
                        const float *t /*3*/, float fx, float fy, float cx, float cy) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            float vcx = SO_MAD(x, vsx, vsx / 2.f), vcy = SO_MAD(y, vsy, vsy / 2.f), vcz = vsz / 2.f; /* :70-71 */
            float camx = dot3(R + 0, vcx, vcy, vcz) + t[0];                              /* :72 */
            float camy = dot3(R + 3, vcx, vcy, vcz) + t[1];
            float camz = dot3(R + 6, vcx, vcy, vcz) + t[2];
            for (int i = 0; i < Z; ++i, camx += 0.f, camy += 0.f, camz += vsz) {        /* :76 */
                float coox = fmaf(fx, so_div(camx, camz), cx), cooy = fmaf(fy, so_div(camy, camz), cy); /* device.hpp:38-39 */
                if (coox < 0 || cooy < 0 || coox >= (float) cols || cooy >= (float) rows) continue; /* :79 */
                if (!(camz > 0)) continue; /* :84 second clause, hoisted before the fetch (both `continue`) */
                if (!(coox == coox) || !(cooy == cooy)) continue; /* NaN coo only arises with camz<=0 */
                int px = (int) floorf(coox), py = (int) floorf(cooy); /* tex2D point sampling :83 */
                float Dp = *(const float *) ((const char *) dists + (size_t) py * step_bytes + (size_t) px * 4);
                if (Dp <= 0.f) continue;                              /* :84 */
                float psdf   = Dp - camz;                             /* :89 */
                float weight = (psdf > -eta) ? 1.f : 0.f;             /* :91 */
                vol[IDX(x, y, i)] = pack_tsdf(psdf, trunc, weight);   /* :93-99 */
            }
        }
}

/* TsdfIntegrator::operator()(phi_global, phi_n_psi) -- tsdf_volume.cu:103-130 */
void so_integrate_fuse(f2 *phi_global, const f2 *phi_n_psi, int X, int Y, int Z, float max_weight) {
    size_t N = (size_t) X * Y * Z;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; ++i) {
        f2 t = phi_n_psi[i];
        if (t.y == 0.f || (t.y == 1.f && (t.x == 0.f || t.x == -1.f))) continue; /* :118 */
        f2 p = phi_global[i];
        f2 o;
        o.x = so_div(fmaf(p.y, p.x, t.x), p.y + 1.f);  /* :124 (__fdividef) */
        o.y = fminf(p.y + 1.f, (float) max_weight); /* :125 */
        phi_global[i] = o;
    }
}

/* init_sphere_kernel -- tsdf_volume.cu:249-275 */
void so_init_sphere(f2 *vol, int X, int Y, int Z, float vsx, float vsy, float vsz, float trunc, float eta, float cx,
                    float cy, float cz, float radius) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            float vx = SO_MAD(x, vsx, vsx / 2.f), vy = SO_MAD(y, vsy, vsy / 2.f), vz = vsz / 2.f; /* :256-257 */
            for (int i = 0; i < Z; vz += vsz, ++i) {
                float d   = so_sqrt(so_pow(vx - cx, 2) + so_pow(vy - cy, 2) + so_pow(vz - cz, 2)); /* :262 */
                float sdf = d - radius;
                float w   = (sdf > -eta) ? 1.f : 0.f;
                vol[IDX(x, y, i)] = pack_tsdf(sdf, trunc, w);
            }
        }
}

static inline float norm3_fma(float x, float y, float z) { /* temp_utils.hpp:86: sqrt(dot(v,v)) */
    float a[3] = {x, y, z};
    return so_sqrt(dot3(a, x, y, z));
}

/* init_box_kernel -- tsdf_volume.cu:181-213 */
void so_init_box(f2 *vol, int X, int Y, int Z, float vsx, float vsy, float vsz, float trunc, float bx, float by,
                 float bz) {
    float ccx = X / 2.f * vsx, ccy = Y / 2.f * vsy, ccz = Z / 2.f * vsz; /* :189-190 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            float vx = SO_MAD(x, vsx, vsx / 2.f) - ccx, vy = SO_MAD(y, vsy, vsy / 2.f) - ccy, vz = (vsz / 2.f) - ccz;
            for (int i = 0; i < Z; vz += vsz, ++i) {
                float dx = fabsf(vx) - bx, dy = fabsf(vy) - by, dz = fabsf(vz) - bz; /* :199 */
                float sdf = fminf(fmaxf(dx, fmaxf(dy, dz)), 0.f) +
                            norm3_fma(fmaxf(dx, 0.f), fmaxf(dy, 0.f), fmaxf(dz, 0.f)); /* :201-202 */
                vol[IDX(x, y, i)] = pack_tsdf(sdf, trunc, 1.f);
            }
        }
}

/* init_ellipsoid_kernel -- tsdf_volume.cu:215-247 */
void so_init_ellipsoid(f2 *vol, int X, int Y, int Z, float vsx, float vsy, float vsz, float trunc, float rx, float ry,
                       float rz) {
    float ccx = X / 2.f * vsx, ccy = Y / 2.f * vsy, ccz = Z / 2.f * vsz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            float vx = SO_MAD(x, vsx, vsx / 2.f) - ccx, vy = SO_MAD(y, vsy, vsy / 2.f) - ccy, vz = (vsz / 2.f) - ccz;
            for (int i = 0; i < Z; vz += vsz, ++i) {
                float k0  = norm3_fma(so_div(vx, rx), so_div(vy, ry), so_div(vz, rz));     /* :233 */
                float k1  = norm3_fma(so_div(vx, rx * rx), so_div(vy, ry * ry), so_div(vz, rz * rz)); /* :234 */
                float sdf = so_div(k0 * (k0 - 1.f), k1);                                   /* :236 */
                vol[IDX(x, y, i)] = pack_tsdf(sdf, trunc, 1.f);
            }
        }
}

/* init_plane_kernel -- tsdf_volume.cu:277-301 */
void so_init_plane(f2 *vol, int X, int Y, int Z, float vsx, float vsy, float vsz, float trunc, float zp) {
    (void) vsx; (void) vsy;
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            float vz = vsz / 2.f;
            for (int i = 0; i < Z; vz += vsz, ++i) vol[IDX(x, y, i)] = pack_tsdf(vz - zp, trunc, 1.f);
        }
}

/* init_torus_kernel -- tsdf_volume.cu:303-334 ; norm(float2) utils.hpp:212-214 */
void so_init_torus(f2 *vol, int X, int Y, int Z, float vsx, float vsy, float vsz, float trunc, float t0, float t1) {
    float ccx = X / 2.f * vsx, ccy = Y / 2.f * vsy, ccz = Z / 2.f * vsz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            float vx = SO_MAD(x, vsx, vsx / 2.f) - ccx, vy = SO_MAD(y, vsy, vsy / 2.f) - ccy, vz = (vsz / 2.f) - ccz;
            for (int i = 0; i < Z; vz += vsz, ++i) {
                float qx  = sqrtf(vx * vx + vz * vz) - t0; /* :321 */
                float sdf = sqrtf(qx * qx + vy * vy) - t1; /* :323 */
                vol[IDX(x, y, i)] = pack_tsdf(sdf, trunc, 1.f);
            }
        }
}

/* ================================================================================================
 * depth image pre-steps -- src/kfusion/cuda/imgproc.cu
 * ============================================================================================== */

/* bilateral_kernel -- imgproc.cu:8-53 (sigma_depth in metres, scaled by 1000 at :43) */
void so_bilateral(const uint16_t *src, int src_step, uint16_t *dst, int dst_step, int rows, int cols, int ksz,
                  float sigma_spatial, float sigma_depth) {
    sigma_depth *= 1000;
    float sss = 0.5f / (sigma_spatial * sigma_spatial), sds = 0.5f / (sigma_depth * sigma_depth); /* :49-50 */
#define SRC(yy, xx) (*(const uint16_t *) ((const char *) src + (size_t)(yy) * src_step + (size_t)(xx) * 2))
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int value = SRC(y, x);
            int tx = (x - ksz / 2 + ksz < cols - 1) ? x - ksz / 2 + ksz : cols - 1; /* :18 */
            int ty = (y - ksz / 2 + ksz < rows - 1) ? y - ksz / 2 + ksz : rows - 1; /* :19 */
            float sum1 = 0, sum2 = 0;
            for (int cy = (y - ksz / 2 > 0 ? y - ksz / 2 : 0); cy < ty; ++cy)
                for (int cx = (x - ksz / 2 > 0 ? x - ksz / 2 : 0); cx < tx; ++cx) {
                    int depth    = SRC(cy, cx);
                    float space2 = (float) ((x - cx) * (x - cx) + (y - cy) * (y - cy));                     /* :28 */
                    float color2 = (float) (int32_t) ((uint32_t)(value - depth) * (uint32_t)(value - depth)); /* :29 */
                    float weight = so_exp_fast(-SO_MAD(space2, sss, color2 * sds));                         /* :31 (__expf) */
                    sum1 = SO_MAD((float) depth, weight, sum1);
                    sum2 += weight;
                }
            float q = so_div(sum1, sum2);
            int r   = (q == q) ? (int) lrintf(q) : 0; /* __float2int_rn; NaN -> 0 */
            *(uint16_t *) ((char *) dst + (size_t) y * dst_step + (size_t) x * 2) = (uint16_t) r; /* :36 */
        }
#undef SRC
}

/* truncate_depth_kernel -- imgproc.cu:60-77 */
void so_truncate_depth(uint16_t *depth, int step, int rows, int cols, float max_dist_m) {
    uint16_t md = (uint16_t) (max_dist_m * 1000.f); /* :75 */
    for (int y = 0; y < rows; ++y) {
        uint16_t *row = (uint16_t *) ((char *) depth + (size_t) y * step);
        for (int x = 0; x < cols; ++x)
            if (row[x] > md) row[x] = 0;
    }
}

/* compute_dists_kernel -- imgproc.cu:233-254 (finv = 1/f computed on the host, :252) */
void so_compute_dists(const uint16_t *depth, int dstep, float *dists, int sstep, int rows, int cols, float fx,
                      float fy, float cx, float cy) {
    float fix = 1.f / fx, fiy = 1.f / fy;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint16_t *drow = (const uint16_t *) ((const char *) depth + (size_t) y * dstep);
        float *srow          = (float *) ((char *) dists + (size_t) y * sstep);
        for (int x = 0; x < cols; ++x) {
            float xl     = (x - cx) * fix;
            float yl     = (y - cy) * fiy;
            float lambda = so_sqrt(SO_MAD(xl, xl, yl * yl) + 1);
            srow[x]      = drow[x] * lambda * 0.001f;
        }
    }
}

/* ================================================================================================
 * vector fields -- src/sobfu/cuda/vector_fields.cu
 * ============================================================================================== */

/* clear_kernel -- vector_fields.cu:36-50 */
void so_clear_field(f4 *f, int X, int Y, int Z) { memset(f, 0, sizeof(f4) * (size_t) X * Y * Z); }

/* init_identity_kernel -- vector_fields.cu:64-79 (z built by repeated += 1.f) */
void so_init_identity(f4 *psi, int X, int Y, int Z) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            float zc = 0.f;
            for (int i = 0; i < Z; zc += 1.f, ++i) psi[IDX(x, y, i)] = mk4((float) x, (float) y, zc, 0.f);
        }
}

/* apply_kernel -- vector_fields.cu:81-100 */
void so_apply(const f2 *phi, f2 *phi_warped, const f4 *psi, int X, int Y, int Z) {
    size_t N = (size_t) X * Y * Z;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; ++i) {
        f4 p          = psi[i];
        phi_warped[i] = interp_tsdf(phi, X, Y, Z, p.x, p.y, p.z);
    }
}

/* apply_kernel on a z-slab: phi is the whole (X, Y, Zg) volume, psi / phi_warped are (X, Y, Lz) slabs whose psi
 * values are absolute voxel coordinates (multi-GPU tiling tests; same arithmetic as so_apply) */
void so_apply_tile(const f2 *phi, int Zg, f2 *phi_warped, const f4 *psi, int X, int Y, int Lz) {
    size_t N = (size_t) X * Y * Lz;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; ++i) {
        f4 p          = psi[i];
        phi_warped[i] = interp_tsdf(phi, X, Y, Zg, p.x, p.y, p.z);
    }
}

/* the same on a 3-D tile: phi is the whole (Xg, Yg, Zg) volume, psi / phi_warped are (Lx, Ly, Lz) local arrays */
void so_apply_tile3(const f2 *phi, int Xg, int Yg, int Zg, f2 *phi_warped, const f4 *psi, int Lx, int Ly, int Lz) {
    size_t N = (size_t) Lx * Ly * Lz;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; ++i) {
        f4 p          = psi[i];
        phi_warped[i] = interp_tsdf(phi, Xg, Yg, Zg, p.x, p.y, p.z);
    }
}

/* estimate_inverse_kernel x n_iters -- vector_fields.cu:111-138 (reference: 48 sweeps, in place) */
void so_estimate_inverse(const f4 *psi, f4 *psi_inv, int X, int Y, int Z, int n_iters) {
    for (int it = 0; it < n_iters; ++it) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int z = 0; z < Z; ++z)
            for (int y = 0; y < Y; ++y)
                for (int x = 0; x < X; ++x) {
                    f4 v = psi_inv[IDX(x, y, z)];
                    f4 u = interp_disp(psi, X, Y, Z, v.x, v.y, v.z);
                    psi_inv[IDX(x, y, z)] = sub4(mk4((float) x, (float) y, (float) z, 0.f), mul4(u, 1.f)); /* :123-124 */
                }
    }
}

/* TsdfDifferentiator::operator() -- vector_fields.cu:157-208 */
void so_tsdf_gradient(const f2 *vol, f4 *grad, int X, int Y, int Z) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y) {
            int z1 = z + 1, z2 = z - 1;
            int y1 = y + 1, y2 = y - 1;
#if SO_MUTANT == 7
            if (z == 0) z2 = z; else if (z == Z - 1) z1 = z;
            if (y == 0) y2 = y; else if (y == Y - 1) y1 = y;
#else
            if (z == 0) z2 = z + 1; else if (z == Z - 1) z1 = z - 1;
            if (y == 0) y2 = y + 1; else if (y == Y - 1) y1 = y - 1;
#endif
            for (int x = 0; x < X; ++x) {
                int x1 = x + 1, x2 = x - 1;
#if SO_MUTANT == 7
                if (x == 0) x2 = x; else if (x == X - 1) x1 = x;
#else
                if (x == 0) x2 = x + 1; else if (x == X - 1) x1 = x - 1;
#endif
                float nx = (vol[IDX(x1, y, z)].x - vol[IDX(x2, y, z)].x) / 2.f;
                float ny = (vol[IDX(x, y1, z)].x - vol[IDX(x, y2, z)].x) / 2.f;
                float nz = (vol[IDX(x, y, z1)].x - vol[IDX(x, y, z2)].x) / 2.f;
                grad[IDX(x, y, z)] = mk4(nx, ny, nz, 0.f);
            }
        }
}

/* SecondOrderDifferentiator::laplacian -- vector_fields.cu:291-337 (NEGATIVE Laplacian) */
void so_laplacian(const f4 *psi, f4 *L, int X, int Y, int Z) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y) {
            int z1 = z + 1, z2 = z - 1;
            int y1 = y + 1, y2 = y - 1;
#if SO_MUTANT == 12
            if (z == 0) z2 = z1; else if (z == Z - 1) z1 = z2;
            if (y == 0) y2 = y1; else if (y == Y - 1) y1 = y2;
#else
            if (z == 0 || z == Z - 1) z1 = z2 = z;
            if (y == 0 || y == Y - 1) y1 = y2 = y;
#endif
            for (int x = 0; x < X; ++x) {
                int x1 = x + 1, x2 = x - 1;
#if SO_MUTANT == 12
                if (x == 0) x2 = x1; else if (x == X - 1) x1 = x2;
#else
                if (x == 0 || x == X - 1) x1 = x2 = x;
#endif
                f4 v = mul4(psi[IDX(x, y, z)], -6.f);
                v    = add4(v, psi[IDX(x1, y, z)]);
                v    = add4(v, psi[IDX(x2, y, z)]);
                v    = add4(v, psi[IDX(x, y1, z)]);
                v    = add4(v, psi[IDX(x, y2, z)]);
                v    = add4(v, psi[IDX(x, y, z1)]);
                v    = add4(v, psi[IDX(x, y, z2)]);
#if SO_MUTANT == 6
                L[IDX(x, y, z)] = mul4(v, 1.f);
#else
                L[IDX(x, y, z)] = mul4(v, -1.f);
#endif
            }
        }
}

/* Differentiator::operator()(J, mode) -- vector_fields.cu:415-472.  Row 3 of Mat4f is left
 * uninitialised by the reference; written as zeros here. */
void so_jacobian(const f4 *psi, m4 *J, int X, int Y, int Z, int mode) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y) {
            int z1 = z + 1, z2 = z - 1;
            if (z == 0) z2 = z + 1; else if (z == Z - 1) z1 = z - 1;
            int y1 = y + 1, y2 = y - 1;
            if (y == 0) y2 = y + 1; else if (y == Y - 1) y1 = y - 1;
            for (int x = 0; x < X; ++x) {
                int x1 = x + 1, x2 = x - 1;
                if (x == 0) x2 = x + 1; else if (x == X - 1) x1 = x - 1;
                f4 jx, jy, jz;
                if (mode == 0) {
                    jx = div4(sub4(psi[IDX(x1, y, z)], psi[IDX(x2, y, z)]), 2.f);
                    jy = div4(sub4(psi[IDX(x, y1, z)], psi[IDX(x, y2, z)]), 2.f);
                    jz = div4(sub4(psi[IDX(x, y, z1)], psi[IDX(x, y, z2)]), 2.f);
                } else {
                    jx = div4(sub4(disp(psi, X, Y, x1, y, z), disp(psi, X, Y, x2, y, z)), 2.f);
                    jy = div4(sub4(disp(psi, X, Y, x, y1, z), disp(psi, X, Y, x, y2, z)), 2.f);
                    jz = div4(sub4(disp(psi, X, Y, x, y, z1), disp(psi, X, Y, x, y, z2)), 2.f);
                }
                m4 v;
                v.r[0] = mk4(jx.x, jy.x, jz.x, 0.f);
                v.r[1] = mk4(jx.y, jy.y, jz.y, 0.f);
                v.r[2] = mk4(jx.z, jy.z, jz.z, 0.f);
                v.r[3] = mk4(0.f, 0.f, 0.f, 0.f);
                J[IDX(x, y, z)] = v;
            }
        }
}

/* ================================================================================================
 * solver pieces -- src/sobfu/cuda/solver.cu
 * ============================================================================================== */

/* calculate_potential_gradient_kernel -- solver.cu:15-33 */
void so_potential_gradient(const f2 *phi_n_psi, const f2 *phi_global, const f4 *grad, const f4 *L, f4 *nabla_U,
                           float w_reg, int X, int Y, int Z) {
    size_t N = (size_t) X * Y * Z;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; ++i) {
#if SO_MUTANT == 9
        float d    = phi_global[i].x - phi_n_psi[i].x;
#else
        float d    = phi_n_psi[i].x - phi_global[i].x;
#endif
        nabla_U[i] = add4(mul4(grad[i], d), mul4(L[i], w_reg)); /* :31 */
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* one 1-D pass: axis 0 rows (assign, solver.cu:237-293), 1 columns (+=, :309-369), 2 depth (+=, :385-446).
 * sum = 0; for j=-3..3: sum += S[3-j] * src(clamp(i+j))  (solver.cu:283-288, clamp-to-edge :246-271) */
static void conv_axis(f4 *dst, const f4 *src, const float *S, int X, int Y, int Z, int axis) {
#if SO_MUTANT == 1 /* columns / depth filter the PREVIOUS pass's result (a composition) instead of adding their own pass over src */
    f4 *prev = NULL;
    if (axis != 0) {
        prev = (f4 *) malloc(sizeof(f4) * (size_t) X * Y * Z);
        memcpy(prev, dst, sizeof(f4) * (size_t) X * Y * Z);
        src = prev;
    }
#endif
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                float sx = 0.f, sy = 0.f, sz = 0.f;
                for (int j = -3; j <= 3; ++j) {
                    int xx = x, yy = y, zz = z;
                    if (axis == 0) xx = clampi(x + j, 0, X - 1);
                    else if (axis == 1) yy = clampi(y + j, 0, Y - 1);
                    else zz = clampi(z + j, 0, Z - 1);
                    f4 v    = src[IDX(xx, yy, zz)];
#if SO_MUTANT == 10
                    if ((axis == 0 && xx != x + j) || (axis == 1 && yy != y + j) || (axis == 2 && zz != z + j)) v = mk4(0.f, 0.f, 0.f, 0.f);
#endif
#if SO_MUTANT == 2
                    float s = S[3 + j];
#else
                    float s = S[3 - j];
#endif
                    sx += v.x * s;
                    sy += v.y * s;
                    sz += v.z * s;
                }
                f4 *d = &dst[IDX(x, y, z)];
                if (axis == 0 || SO_MUTANT == 1) {
                    *d = mk4(sx, sy, sz, 0.f);
                } else {
                    d->x += sx;
                    d->y += sy;
                    d->z += sz;
                }
            }
#if SO_MUTANT == 1
    free(prev);
#endif
}
void so_convolution_rows(f4 *dst, const f4 *src, const float *S, int X, int Y, int Z) { conv_axis(dst, src, S, X, Y, Z, 0); }
void so_convolution_columns(f4 *dst, const f4 *src, const float *S, int X, int Y, int Z) { conv_axis(dst, src, S, X, Y, Z, 1); }
void so_convolution_depth(f4 *dst, const f4 *src, const float *S, int X, int Y, int Z) { conv_axis(dst, src, S, X, Y, Z, 2); }

/* update_psi_kernel -- solver.cu:53-69 */
void so_update_psi(f4 *psi, const f4 *nabla_U_S, f4 *updates, float alpha, int X, int Y, int Z) {
    size_t N = (size_t) X * Y * Z;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; ++i) {
        f4 u       = mul4(nabla_U_S[i], alpha);
        updates[i] = u;
#if SO_MUTANT == 5
        psi[i].x += u.x;
        psi[i].y += u.y;
        psi[i].z += u.z;
#else
        psi[i].x -= u.x;
        psi[i].y -= u.y;
        psi[i].z -= u.z;
#endif
    }
}

/* decompose_sobolev_filter -- src/sobfu/solver.cpp:160-262.  Returns 0 on success, -1 for an (s, lambda)
 * pair the reference's table does not list (the reference leaves h_S_i uninitialised there). */
int so_sobolev_filter(int s, float lambda, float *h) {
    int ok = 0;
    if (s == 3 && lambda == 0.1f) { h[0] = 0.06537f; h[1] = 0.99572f; h[2] = h[0]; ok = 1; }
    if (s == 7) {
        if (lambda == 0.05f) { h[0] = 0.00006f; h[1] = 0.00015f; h[2] = 0.03917f; h[3] = 0.99846f; ok = 1; }
        if (lambda == 0.1f) { h[0] = 0.00030f; h[1] = 0.00441f; h[2] = 0.06571f; h[3] = 0.99565f; ok = 1; }
        if (lambda == 0.2f) { h[0] = 0.00120f; h[1] = 0.01094f; h[2] = 0.10204f; h[3] = 0.98941f; ok = 1; }
        if (lambda == 0.4f) { h[0] = 0.00169f; h[1] = 0.01312f; h[2] = 0.10927f; h[3] = 0.98781f; ok = 1; }
        if (ok) { h[4] = h[2]; h[5] = h[1]; h[6] = h[0]; }
    }
    if (s == 9) {
        if (lambda == 0.05f) { h[0] = 0.000003f; h[1] = 0.00006f; h[2] = 0.00155f; h[3] = 0.03917f; h[4] = 0.99846f; ok = 1; }
        if (lambda == 0.1f) { h[0] = 0.00002f; h[1] = 0.00030f; h[2] = 0.00441f; h[3] = 0.06571f; h[4] = 0.99565f; ok = 1; }
        if (ok) { h[5] = h[3]; h[6] = h[2]; h[7] = h[1]; h[8] = h[0]; }
    }
    if (s == 11 && lambda == 0.1f) {
        h[0] = 0.0000015f; h[1] = 0.00002f; h[2] = 0.00030f; h[3] = 0.00441f; h[4] = 0.06571f; h[5] = 0.99565f;
        h[6] = h[4]; h[7] = h[3]; h[8] = h[2]; h[9] = h[1]; h[10] = h[0];
        ok = 1;
    }
    if (!ok) return -1;
    float sum = 0.f;
    for (int i = 0; i < s; ++i) sum += h[i]; /* :253-256 */
    for (int i = 0; i < s; ++i) h[i] /= sum; /* :258-260 */
    return 0;
}

/* ================================================================================================
 * reductions -- src/sobfu/cuda/reductor.cu, src/sobfu/reductor.cpp, src/sobfu/precomp.cpp
 * ============================================================================================== */

static int next_pow2(int x) { /* precomp.cpp:8-18 */
    if (x < 0) return 0;
    --x;
    x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
    return x + 1;
}

/* get_num_blocks_and_threads(n, 65536, 512, ...) -- precomp.cpp:20-43, reductor.cpp:16 */
void so_reduce_config(int n, int *blocks, int *threads) {
    int maxThreads = 512, maxBlocks = 65536;
    int t = (n < maxThreads * 2) ? next_pow2((n + 1) / 2) : maxThreads;
    int b = (n + (t * 2 - 1)) / (t * 2);
    if (b > maxBlocks) b = maxBlocks;
    *blocks  = b;
    *threads = t;
}

/* shared tree: per-thread accumulation over a grid-stride of 2*threads*blocks (reductor.cu:17-35), then
 * stride-halving pairwise tree (reductor.cu:41-107), then sequential host sum (reductor.cpp:68-79). */
typedef float (*elem_fn)(const void *a, const void *b, size_t i);
static float el_data(const void *a, const void *b, size_t i) {
    float d = ((const f2 *) a)[i].x - ((const f2 *) b)[i].x;
    return d * d; /* reductor.cu:26 */
}
static float el_reg(const void *a, const void *b, size_t i) {
    (void) b;
    const m4 *J = (const m4 *) a;
    return norm_sq4(J[i].r[0]) + norm_sq4(J[i].r[1]) + norm_sq4(J[i].r[2]); /* reductor.cu:129 */
}
static float tree_sum(const void *a, const void *b, size_t n, elem_fn fn, float *partials_out) {
    int blocks, threads;
    so_reduce_config((int) n, &blocks, &threads);
    float *partials = partials_out ? partials_out : (float *) malloc(sizeof(float) * blocks);
    size_t grid     = (size_t) threads * 2 * blocks;
#pragma omp parallel for schedule(static)
    for (int blk = 0; blk < blocks; ++blk) {
        float s[512];
        for (int t = 0; t < threads; ++t) {
            float my = 0.f;
            for (size_t i = (size_t) blk * threads * 2 + t; i < n; i += grid) {
#if SO_NVCC_MODE
                if (fn == el_data && (so_nvcc_mask & 16u)) { /* `mySum += d * d` contracts (reductor.cu:26-31) */
                    float d = ((const f2 *) a)[i].x - ((const f2 *) b)[i].x;
                    my      = fmaf(d, d, my);
                    if (i + threads < n) {
                        d  = ((const f2 *) a)[i + threads].x - ((const f2 *) b)[i + threads].x;
                        my = fmaf(d, d, my);
                    }
                    continue;
                }
#endif
                my += fn(a, b, i);
                if (i + threads < n) my += fn(a, b, i + threads);
            }
            s[t] = my;
        }
        for (int h = threads / 2; h >= 1; h /= 2)
            for (int t = 0; t < h; ++t) s[t] = s[t] + s[t + h];
        partials[blk] = s[0];
    }
    float r = 0.f;
    for (int i = 0; i < blocks; ++i) r += partials[i];
    if (!partials_out) free(partials);
    return r;
}

/* Reductor::data_energy -- reductor.cpp:38-43 */
float so_data_energy(const f2 *phi_global, const f2 *phi_n, int n) {
    return 0.5f * tree_sum(phi_global, phi_n, (size_t) n, el_data, NULL);
}
/* Reductor::reg_energy_sobolev -- reductor.cpp:45-50 */
float so_reg_energy_sobolev(const m4 *J, int n) { return 0.5f * tree_sum(J, NULL, (size_t) n, el_reg, NULL); }

/* Reductor::max_update_norm -- reductor.cpp:52-57; reduce_max_kernel reductor.cu:342-456;
 * final_reduce_max reductor.cpp:81-94.  out[0] = max norm, out[1] = float-encoded index. */
void so_max_update_norm(const f4 *updates, int n_, float *out) {
    size_t n = (size_t) n_;
    int blocks, threads;
    so_reduce_config(n_, &blocks, &threads);
    f2 *partials = (f2 *) malloc(sizeof(f2) * blocks);
    size_t grid  = (size_t) threads * 2 * blocks;
#pragma omp parallel for schedule(static)
    for (int blk = 0; blk < blocks; ++blk) {
        f2 s[512];
        for (int t = 0; t < threads; ++t) {
            f2 lm = {0.f, 0.f};
            for (size_t i = (size_t) blk * threads * 2 + t; i < n; i += grid) {
                float v = norm4(updates[i]);
                if (v > lm.x) { lm.x = v; lm.y = (float) (unsigned) i; }
                if (i + threads < n) {
                    float w = norm4(updates[i + threads]);
                    if (w > lm.x) { lm.x = w; lm.y = (float) (unsigned) i + threads; } /* reductor.cu:371 */
                }
            }
            s[t] = lm;
        }
        for (int h = threads / 2; h >= 1; h /= 2)
            for (int t = 0; t < h; ++t)
                if (s[t + h].x > s[t].x) s[t] = s[t + h];
        partials[blk] = s[0];
    }
    f2 r = {0.f, 0.f};
    for (int i = 0; i < blocks; ++i)
        if (partials[i].x > r.x) r = partials[i];
    free(partials);
    out[0] = r.x;
    out[1] = r.y;
}

/* ================================================================================================
 * the solver loop -- sobfu::device::estimate_psi, src/sobfu/cuda/solver.cu:85-205
 * ============================================================================================== */

typedef struct {
    int verbosity, max_iter, s;
    float max_update_norm, lambda, alpha, w_reg;
} so_solver_params; /* include/sobfu/solver.hpp:16-19 */

/* Workspace the reference's Solver ctor allocates (src/sobfu/solver.cpp:41-58): caller passes scratch
 * fields grad, L, nabla_U, nabla_U_S, updates (float4 x N) and J (Mat4f x N, may be NULL when
 * compute_jacobian == 0).  trace (may be NULL): per iteration {e_data, e_reg, max_norm, max_idx}; e_* are
 * NaN on iterations where the reference would not have evaluated them (solver.cu:132-133).
 * compute_jacobian = 1 reproduces the reference's per-iteration (dead unless verbose) Jacobian pass.
 * Returns the number of iterations executed (iter at break, or max_iter). */
int so_estimate_psi(const f2 *phi_global, f2 *phi_global_psi_inv, const f2 *phi_n, f2 *phi_n_psi, f4 *psi, f4 *psi_inv,
                    f4 *grad, f4 *L, f4 *nabla_U, f4 *nabla_U_S, f4 *updates, m4 *J, int X, int Y, int Z,
                    const so_solver_params *p, int compute_jacobian, int inverse_iters, float *trace) {
    float S[16];
    if (so_sobolev_filter(p->s, p->lambda, S) != 0) return -1;
    int N = X * Y * Z;

    so_apply(phi_n, phi_n_psi, psi, X, Y, Z); /* :106 */
    int iter = 1, executed = 0;
    while (iter <= p->max_iter) { /* :114 */
        so_tsdf_gradient(phi_n_psi, grad, X, Y, Z);                       /* :120 */
        if (compute_jacobian && J) so_jacobian(psi, J, X, Y, Z, 1);       /* :124 */
        so_laplacian(psi, L, X, Y, Z);                                    /* :127 */
        int report = (p->verbosity == 1 && (iter == 1 || iter % 50 == 0 || iter == p->max_iter)) || p->verbosity == 2;
        float e_data = NAN, e_reg = NAN;
        if (report && J) { /* :132-142 */
            if (!compute_jacobian) so_jacobian(psi, J, X, Y, Z, 1);
            e_data = so_data_energy(phi_global, phi_n_psi, N);
            e_reg  = so_reg_energy_sobolev(J, N);
        }
        so_potential_gradient(phi_n_psi, phi_global, grad, L, nabla_U, p->w_reg, X, Y, Z); /* :149 */
        so_convolution_rows(nabla_U_S, nabla_U, S, X, Y, Z);                               /* :155 */
        so_convolution_columns(nabla_U_S, nabla_U, S, X, Y, Z);                            /* :157 */
        so_convolution_depth(nabla_U_S, nabla_U, S, X, Y, Z);                              /* :159 */
        so_update_psi(psi, nabla_U_S, updates, p->alpha, X, Y, Z);                         /* :163 */
        so_apply(phi_n, phi_n_psi, psi, X, Y, Z);                                          /* :168 */
        float mx[2];
        so_max_update_norm(updates, N, mx); /* :172 */
        executed = iter;
        if (trace) {
            trace[4 * (iter - 1) + 0] = e_data;
            trace[4 * (iter - 1) + 1] = e_reg;
            trace[4 * (iter - 1) + 2] = mx[0];
            trace[4 * (iter - 1) + 3] = mx[1];
        }
        if (mx[0] <= p->max_update_norm) break; /* :183 */
        iter++;
    }
    so_init_identity(psi_inv, X, Y, Z);                          /* :196 */
    so_estimate_inverse(psi, psi_inv, X, Y, Z, inverse_iters);   /* :197 (48) */
    so_apply(phi_global, phi_global_psi_inv, psi_inv, X, Y, Z);  /* :199 */
    return executed;
}

int so_num_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void so_set_num_threads(int n) {
#if defined(_OPENMP)
    omp_set_num_threads(n);
#else
    (void) n;
#endif
}

/* ================================================================================================
 * marching cubes -- src/kfusion/cuda/marching_cubes.cu, src/kfusion/marching_cubes.cpp
 *
 * PARITY UNPINNED for this block: the reference's tests and SURVEY Appendix B hold no marching-cubes numbers.  Arithmetic
 * follows the oracle conventions of the rest of this file (IEEE / and sqrt, rsqrt(x) = 1/sqrt(x), no contraction of
 * plain a*b+c, explicit fma only where the reference writes one: dot(), Aff3f * v).
 * The case table is the Lorensen-Cline / Bourke table (mc_table.inc, see tools/pack_mc_table.py).
 * ============================================================================================== */
static const uint64_t MC_TRI[256] = {
#include "mc_table.inc"
};
static inline int mc_edge(int cube, int k) { return (int) ((MC_TRI[cube] >> (4 * k)) & 15u); } /* triTable[cube][k]; 15 = -1 */
int so_mc_num_verts(int cube) { /* numVertsTable, marching_cubes.cpp:358 = entries before the first -1 */
    int n = 0;
    while (n < 16 && mc_edge(cube, n) != 15) ++n;
    return n;
}

/* CubeIndexEstimator::computeCubeIndex -- marching_cubes.cu:38-79 (isoValue = 0, internal.hpp:95) */
static int mc_cube_index(const f2 *vol, int X, int Y, int Z, int x, int y, int z, float f[8]) {
    (void) Z;
    static const int dx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, dy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, dz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    for (int c = 0; c < 8; ++c) {
        f2 v = vol[IDX(x + dx[c], y + dy[c], z + dz[c])];
        f[c] = v.x;
        if (v.y == 0.f) return 0;
    }
    int cube = 0;
    for (int c = 0; c < 8; ++c) cube += (f[c] < 0.f) << c;
    return cube;
}

/* getOccupiedVoxels -- marching_cubes.cu:81-163.  occupied: 3 rows of `stride` ints (voxel index, vertex count, vertex
 * offset).  The reference appends in warp-arrival order (atomicAdd on a global counter), i.e. every run is some
 * permutation; this restatement and the HIP path emit the canonical member of that family, ascending voxel index (which
 * a run whose warps advance in lockstep produces), and cap at max_size in that order.  Returns min(found, max_size). */
int so_mc_occupied_voxels(const f2 *vol, int X, int Y, int Z, int *occupied, int stride, int max_size) {
    int count = 0;
    for (int z = 0; z < Z - 1; ++z) /* :94 */
        for (int y = 0; y + 1 < Y; ++y)
            for (int x = 0; x + 1 < X; ++x) { /* :97 */
                float f[8];
                int cube = mc_cube_index(vol, X, Y, Z, x, y, z, f);
                int nv   = (cube == 0 || cube == 255) ? 0 : so_mc_num_verts(cube); /* :102 */
                if (nv > 0) {
                    if (count < max_size) { /* :119-122 */
                        occupied[count]          = X * Y * z + X * y + x;
                        occupied[stride + count] = nv;
                    }
                    ++count;
                }
            }
    return count < max_size ? count : max_size; /* :148 */
}

/* computeOffsetsAndTotalVertices -- marching_cubes.cu:165-181: exclusive scan of row 1 into row 2 */
int so_mc_offsets(int *occupied, int stride, int count) {
    int run = 0;
    for (int i = 0; i < count; ++i) {
        occupied[2 * stride + i] = run;
        run += occupied[stride + i];
    }
    return run;
}

static inline void mc_interp(const float p0[3], const float p1[3], float f0, float f1, float out[3]) { /* :193-199 */
    float t = (0.f - f0) / (f1 - f0 + 1e-15f);
    for (int k = 0; k < 3; ++k) out[k] = p0[k] + t * (p1[k] - p0[k]);
}

/* generateTriangles -- marching_cubes.cu:201-313.  pose: R (9, row major), t (3).  Vertices / normals are float4
 * (x, -y, -z, 1) (store_point :270-273).  Triangles whose three vertices do not fit below max_vertices are dropped (the
 * reference does not check its 6M-vertex buffer). */
void so_mc_generate_triangles(const f2 *vol, int X, int Y, int Z, const int *occupied, int stride, int count, float sx,
                              float sy, float sz, const float *R, const float *t, f4 *out_v, f4 *out_n, int max_vertices) {
    static const int dx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, dy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, dz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    static const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7}; /* :232-243 */
    const float cs[3] = {sx / X, sy / Y, sz / Z}; /* :292-294 */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < count; ++idx) {
        int voxel = occupied[idx];
        int z = voxel / (X * Y), y = (voxel - z * X * Y) / X, x = (voxel - z * X * Y) - y * X; /* :210-212 */
        float f[8] = {0}, v[8][3], vl[12][3];
        int cube = mc_cube_index(vol, X, Y, Z, x, y, z, f);
        for (int c = 0; c < 8; ++c) { /* get_node_coo :183-191 */
            v[c][0] = ((float) (x + dx[c]) + 0.5f) * cs[0];
            v[c][1] = ((float) (y + dy[c]) + 0.5f) * cs[1];
            v[c][2] = ((float) (z + dz[c]) + 0.5f) * cs[2];
        }
        for (int e = 0; e < 12; ++e) mc_interp(v[ea[e]], v[eb[e]], f[ea[e]], f[eb[e]], vl[e]);
        int nv = so_mc_num_verts(cube); /* :247 */
        for (int i = 0; i < nv; i += 3) {
            int index = occupied[2 * stride + idx] + i;
            if (index + 3 > max_vertices) break;
            const float *p1 = vl[mc_edge(cube, i)], *p2 = vl[mc_edge(cube, i + 1)], *p3 = vl[mc_edge(cube, i + 2)];
            float a[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]}, b[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
            float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; /* cross, temp_utils.hpp:92 */
            float inv = 1.f / sqrtf(dot3(c, c[0], c[1], c[2]));                                          /* normalized :90 */
            f4 n = {c[0] * inv, -(c[1] * inv), -(c[2] * inv), 1.f};
            const float *p[3] = {p1, p2, p3};
            for (int k = 0; k < 3; ++k) { /* pose * vertex (device.hpp:57-61), store_point */
                float wx = dot3(R + 0, p[k][0], p[k][1], p[k][2]) + t[0], wy = dot3(R + 3, p[k][0], p[k][1], p[k][2]) + t[1],
                      wz = dot3(R + 6, p[k][0], p[k][1], p[k][2]) + t[2];
                out_v[index + k] = (f4){wx, -wy, -wz, 1.f};
                out_n[index + k] = n;
            }
        }
    }
}
