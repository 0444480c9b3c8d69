/*
 * sobfu_hip.h -- C ABI of the MI355X-native SobolevFusion hot path (libsobfu_hip.so, gfx950).
 *
 * This is the drop-in boundary: one entry point per host-callable launcher of the reference's "L1 device
 * API" (include/sobfu/{solver,vector_fields,reductor}.hpp `namespace device`, include/kfusion/internal.hpp
 * :189-257), plus an opaque solver handle replacing sobfu::cuda::Solver's workspace + hot loop.  The C++
 * shells under include/sobfu_amd/ rebuild the reference's class surface on top of it; INTEGRATION.md shows
 * the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer marked `d_` is a DEVICE pointer the callee does not own;
 *  - volumes are dense, x fastest: idx = x + X*(y + Y*z) (reference: src/sobfu/cuda/vector_fields.cu:20-22);
 *    TSDF voxel = float2 {tsdf, weight} (8 B), vector-field voxel = float4 with w == 0 (16 B), Jacobian voxel
 *    = 4 x float4 (64 B, row 3 unused) -- the reference's layouts, unchanged;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Launches are asynchronous; nothing
 *    synchronises unless documented (the reference's per-call cudaDeviceSynchronize is NOT reproduced);
 *  - return value: 0 on success, otherwise a hipError_t (positive) or a SOBFU_E_* code (negative).  The
 *    reference prints and calls exit(0) on any CUDA error (src/kfusion/device_memory.cpp:7-10); the C++ shells
 *    keep that behaviour, the C ABI itself never exits.
 */
#ifndef SOBFU_HIP_H
#define SOBFU_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define SOBFU_HIP_ABI_VERSION 3

#define SOBFU_E_BADARG (-1)      /* null pointer, non-positive dims, ... */
#define SOBFU_E_FILTER (-2)      /* (s, lambda) not in the reference's Sobolev filter table */
#define SOBFU_E_UNSUPPORTED (-3) /* valid in the reference but outside this build's limits */

int sobfu_hip_abi_version(void);
/* Human-readable text for a return code of this library. */
const char* sobfu_hip_error_string(int code);

/* ------------------------------------------------------------------------------------------------------
 * TSDF volume  (kfusion::device::*, include/kfusion/internal.hpp:189-198, src/kfusion/cuda/tsdf_volume.cu)
 * ---------------------------------------------------------------------------------------------------- */
/* clear_volume (tsdf_volume.cu:23-46) */
int sobfu_hip_clear_volume(float* d_vol, int X, int Y, int Z, void* stream);
/* integrate(dists, volume, aff, proj) (tsdf_volume.cu:56-101,141-162).  d_dists: pitched float image
 * (step in bytes) of ray lengths in metres; R (row-major 3x3) and t: vol2cam = camera_pose^-1 * volume_pose
 * (src/kfusion/tsdf_volume.cpp:95-106); voxel_size[3]; trunc/eta in metres.  Voxels that project outside
 * the image, onto Dp <= 0 or behind the camera are left untouched. */
int sobfu_hip_integrate_depth(const float* d_dists, int dists_step_bytes, int rows, int cols, float* d_vol, int X,
                              int Y, int Z, const float voxel_size[3], float trunc_dist, float eta,
                              const float R[9], const float t[3], float fx, float fy, float cx, float cy,
                              void* stream);
/* integrate(phi_global, phi_n_psi) (tsdf_volume.cu:103-130,164-173): running weighted average fusion. */
int sobfu_hip_integrate_fuse(float* d_phi_global, const float* d_phi_n_psi, int X, int Y, int Z, float max_weight,
                             void* stream);
/* init_{sphere,box,ellipsoid,plane,torus} (tsdf_volume.cu:181-382): analytic truncated SDFs. */
int sobfu_hip_init_sphere(float* d_vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist, float eta,
                          const float centre[3], float radius, void* stream);
int sobfu_hip_init_box(float* d_vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                       const float b[3], void* stream);
int sobfu_hip_init_ellipsoid(float* d_vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                             const float r[3], void* stream);
int sobfu_hip_init_plane(float* d_vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist, float z,
                         void* stream);
int sobfu_hip_init_torus(float* d_vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                         const float t[2], void* stream);

/* ------------------------------------------------------------------------------------------------------
 * depth pre-steps  (include/kfusion/internal.hpp:236-239, src/kfusion/cuda/imgproc.cu:8-77,233-254)
 * ---------------------------------------------------------------------------------------------------- */
/* bilateralFilter: uint16 mm depth, sigma_depth in metres (scaled x1000 inside, imgproc.cu:43). */
int sobfu_hip_bilateral_filter(const uint16_t* d_src, int src_step_bytes, uint16_t* d_dst, int dst_step_bytes,
                               int rows, int cols, int kernel_size, float sigma_spatial, float sigma_depth,
                               void* stream);
/* truncateDepth: depth > max_dist_m*1000 -> 0, in place. */
int sobfu_hip_truncate_depth(uint16_t* d_depth, int step_bytes, int rows, int cols, float max_dist_m, void* stream);
/* compute_dists: uint16 mm depth -> ray length in metres. */
int sobfu_hip_compute_dists(const uint16_t* d_depth, int depth_step_bytes, float* d_dists, int dists_step_bytes,
                            int rows, int cols, float fx, float fy, float cx, float cy, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * vector fields  (sobfu::device::*, include/sobfu/vector_fields.hpp:140-241, src/sobfu/cuda/vector_fields.cu)
 * ---------------------------------------------------------------------------------------------------- */
int sobfu_hip_clear_field(float* d_field, int X, int Y, int Z, void* stream);   /* clear (:28-50) */
int sobfu_hip_init_identity(float* d_psi, int X, int Y, int Z, void* stream);   /* init_identity (:56-79) */
/* apply (:81-109): phi_warped(x) = trilinear phi(psi(x)), weight = nearest-floor weight. */
int sobfu_hip_apply(const float* d_phi, float* d_phi_warped, const float* d_psi, int X, int Y, int Z, void* stream);
/* estimate_inverse (:111-138): n_sweeps (reference: 48) in-place fixed-point sweeps on d_psi_inv. */
int sobfu_hip_estimate_inverse(const float* d_psi, float* d_psi_inv, int X, int Y, int Z, int n_sweeps,
                               void* stream);
/* The tail of Solver::estimate_psi fused (solver.cu:196-199): psi_inv <- identity, n_sweeps fixed-point sweeps, and
 * phi_warped = phi o psi_inv, in one pass (same values as init_identity + estimate_inverse + apply). */
int sobfu_hip_inverse_and_warp(const float* d_psi, float* d_psi_inv, const float* d_phi, float* d_phi_warped, int X, int Y, int Z,
                               int n_sweeps, void* stream);
/* TsdfDifferentiator::calculate (:144-208): central-difference gradient, exact 0 on boundary faces. */
int sobfu_hip_tsdf_gradient(const float* d_vol, float* d_grad, int X, int Y, int Z, void* stream);
/* SecondOrderDifferentiator::calculate (:278-337): NEGATIVE 7-point Laplacian of psi. */
int sobfu_hip_laplacian(const float* d_psi, float* d_L, int X, int Y, int Z, void* stream);
/* Differentiator::calculate (mode 0) / calculate_deformation_jacobian (mode 1) (:389-472). */
int sobfu_hip_jacobian(const float* d_psi, float* d_J, int X, int Y, int Z, int mode, void* stream);
int sobfu_hip_clear_jacobian(float* d_J, int X, int Y, int Z, void* stream);    /* clear(Jacobian&) (:353-383) */

/* ------------------------------------------------------------------------------------------------------
 * solver launchers  (include/sobfu/solver.hpp:109-136, src/sobfu/cuda/solver.cu)
 * ---------------------------------------------------------------------------------------------------- */
/* decompose_sobolev_filter (src/sobfu/solver.cpp:160-262): writes s normalised taps to host array out. */
int sobfu_hip_sobolev_filter(int s, float lambda, float* out);
/* calculate_potential_gradient (solver.cu:15-47) */
int sobfu_hip_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_grad,
                                 const float* d_L, float* d_nabla_U, float w_reg, int X, int Y, int Z, void* stream);
/* convolution_{rows,columns,depth} (solver.cu:237-459): rows ASSIGNS dst = Sx*src, columns / depth ACCUMULATE
 * dst += S*src.  taps: 7 HOST floats (the reference's __constant__ S, solver.cu:229-234, passed by value so the
 * library holds no global state). */
int sobfu_hip_convolution_rows(float* d_dst, const float* d_src, const float taps[7], int w, int h, int d,
                               void* stream);
int sobfu_hip_convolution_columns(float* d_dst, const float* d_src, const float taps[7], int w, int h, int d,
                                  void* stream);
int sobfu_hip_convolution_depth(float* d_dst, const float* d_src, const float taps[7], int w, int h, int d,
                                void* stream);
/* update_psi (solver.cu:53-79): updates = alpha*nabla_U_S; psi -= updates. */
int sobfu_hip_update_psi(float* d_psi, const float* d_nabla_U_S, float* d_updates, float alpha, int X, int Y, int Z,
                         void* stream);

/* ------------------------------------------------------------------------------------------------------
 * reductions  (sobfu::device::Reductor, include/sobfu/reductor.hpp:24-50, src/sobfu/reductor.cpp,
 * src/sobfu/cuda/reductor.cu, launch sizing src/sobfu/precomp.cpp:20-43)
 * ---------------------------------------------------------------------------------------------------- */
int sobfu_hip_reduce_config(int n, int* blocks, int* threads);
/* Each call runs the block reduction with the reference's tree shape, copies the block partials to the host
 * and finishes on the CPU exactly as final_reduce / final_reduce_max do; it SYNCHRONISES the stream.
 * d_scratch: device scratch of >= blocks*8 bytes. */
int sobfu_hip_data_energy(const float* d_phi_global, const float* d_phi_n, int n, void* d_scratch, float* out,
                          void* stream);
int sobfu_hip_reg_energy_sobolev(const float* d_J, int n, void* d_scratch, float* out, void* stream);
/* out[0] = max ||update|| (sqrt rounded down), out[1] = float-encoded linear index (reductor.cu:357-368). */
int sobfu_hip_max_update_norm(const float* d_updates, int n, void* d_scratch, float out[2], void* stream);
/* Same value as reg_energy_sobolev(J(psi, mode 1)) without materialising the 64 B/voxel Jacobian. */
int sobfu_hip_reg_energy_sobolev_from_psi(const float* d_psi, int X, int Y, int Z, void* d_scratch, float* out,
                                          void* stream);

/* ------------------------------------------------------------------------------------------------------
 * fused iteration kernels (MI355X-native decomposition of one pass of solver.cu:114-193; no reference
 * counterpart -- results are bit-identical to the launcher sequence above)
 * ---------------------------------------------------------------------------------------------------- */
/* pass A: nabla_U = (phi_n_psi - phi_global) * grad(phi_n_psi) + w_reg * (-Lap psi)   [a14 + a15 + a17] */
int sobfu_hip_fused_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_psi,
                                       float* d_nabla_U, float w_reg, int X, int Y, int Z, void* stream);
/* pass B: u = alpha * (Sx + Sy + Sz)(nabla_U); psi -= u; phi_n_psi = phi_n o psi; max ||u||^2 folded into
 * d_max_sq_slots[256] with atomic max (uint32 view of non-negative floats)   [a18 + a19 + a12 + a20].
 * d_updates may be NULL (updates consumed in registers). */
int sobfu_hip_fused_smooth_update_apply(const float* d_nabla_U, float* d_psi, const float* d_phi_n,
                                        float* d_phi_n_psi, float* d_updates, uint32_t* d_max_sq_slots,
                                        const float taps[7], float alpha, int X, int Y, int Z, void* stream);

/* Compact-format conversions (n = number of voxels; the format the multi-GPU tile entry points below accept with compact != 0):
 * float4 <-> packed xyz (unpack leaves .w untouched) and float2 {tsdf, weight} -> tsdf. */
int sobfu_hip_pack_vec3(const float* d_src4, float* d_dst3, size_t n, void* stream);
int sobfu_hip_unpack_vec3(const float* d_src3, float* d_dst4, size_t n, void* stream);
int sobfu_hip_extract_tsdf(const float* d_src2, float* d_dst1, size_t n, void* stream);

/* Multi-GPU tiles (the 2x2x2 split of BASELINE config 4, z-slabs as 1x1xN): every field argument is a LOCAL
 * array (Lx, Ly, Lz) whose cell (0, 0, 0) is global cell (xb, yb, zb) of the (Xg, Yg, Zg) volume and that carries halo cells on every side that faces a neighbour tile;
 * d_phi_n / d_phi / the d_psi of estimate_inverse are WHOLE volumes.  `box` = (x0, x1, y0, y1, z0, z1): the cells a launch produces;
 * `own`: the cells that belong to this rank (they alone enter the max-norm).  thin != 0 evaluates the box DIRECTLY -- one lane per
 * cell, every tap read through the caches -- instead of by a z-march: for boxes a few cells thick (the one-cell shells of a tile,
 * halo faces), same results.  A thin pass-A launch is never gated (it writes scratch only).  Boundary rules apply at array edges, which are volume boundaries
 * exactly where a tile has no halo.  compact != 0: the field arguments are in the compact iteration format -- psi / nabla_U 12-byte
 * xyz triples, phi_n o psi / phi_global / phi_n tsdf-only floats.  d_prev_slots (may be NULL) and max_update_norm form the
 * device-side convergence gate (see the solver handle). */
int sobfu_hip_tile3_init_identity(float* d_psi, int Lx, int Ly, int Lz, int xb, int yb, int zb, void* stream);
int sobfu_hip_tile3_apply(const float* d_phi, int Xg, int Yg, int Zg, float* d_phi_warped, const float* d_psi, int Lx, int Ly, int Lz,
                          void* stream);
int sobfu_hip_tile3_estimate_inverse(const float* d_psi, int Xg, int Yg, int Zg, float* d_psi_inv, int Lx, int Ly, int Lz, int xb, int yb,
                                     int zb, int n_sweeps, void* stream);
/* The per-frame tail of a tile (reference src/sobfu/cuda/solver.cu:196-199; src/sobfu/cuda/vector_fields.cu:111-138) on a WINDOW of its
 * sources instead of the all-gathered volume: psi^-1(x), and every point its fixed-point iteration visits, lies within
 * r = max |psi - id| of x, so a tile needs psi / phi_global only on its owned cells widened by ceil(r) + 1 cells.
 * win = (Wx, Wy, Wz, wbx, wby, wbz): extents of the window array and the global cell of its cell (0, 0, 0); box = (x0, x1, y0, y1, z0, z1):
 * the LOCAL cells to produce (others are left untouched); clamps act on the global extents (Xg, Yg, Zg) as in the whole-volume kernels:
 * same bits.  *d_violation (zeroed by the caller) becomes 1 when a sample fell outside the window -- the results are then invalid and the
 * caller repeats the tail on all-gathered sources.  sobfu_hip_tile3_max_displacement folds max(|psi - id| components) over `box` into
 * *d_max_bits (atomicMax on the float's bit pattern; zero it first; NaN / inf count as 3e38). */
int sobfu_hip_tile3_estimate_inverse_window(const float* d_psi_win, const int win[6], int Xg, int Yg, int Zg, float* d_psi_inv, int Lx, int Ly, int Lz,
                                            int xb, int yb, int zb, const int box[6], int n_sweeps, int* d_violation, void* stream);
int sobfu_hip_tile3_apply_window(const float* d_phi_win, const int win[6], int Xg, int Yg, int Zg, float* d_phi_warped, const float* d_psi, int Lx,
                                 int Ly, int Lz, const int box[6], int* d_violation, void* stream);
int sobfu_hip_tile3_max_displacement(const float* d_psi, int Lx, int Ly, int Lz, int xb, int yb, int zb, const int box[6], uint32_t* d_max_bits,
                                     void* stream);
int sobfu_hip_tile3_integrate_depth(const float* d_dists, int dists_step_bytes, int rows, int cols, float* d_vol_local, int Lx, int Ly,
                                    int Lz, int xb, int yb, int zb, const float voxel_size[3], float trunc_dist, float eta,
                                    const float R[9], const float t[3], float fx, float fy, float cx, float cy, void* stream);
int sobfu_hip_tile3_potential_gradient(const float* d_phi_n_psi, const float* d_phi_global, const float* d_psi, float* d_nabla_U,
                                       float w_reg, int Lx, int Ly, int Lz, const int box[6], int thin,
                                       const uint32_t* d_prev_slots, float max_update_norm, int compact, void* stream);
int sobfu_hip_tile3_smooth_update_apply(const float* d_nabla_U, float* d_psi, const float* d_phi_n, float* d_phi_n_psi, float* d_updates,
                                        uint32_t* d_max_sq_slots, const float taps[7], float alpha, int Lx, int Ly, int Lz, int Xg,
                                        int Yg, int Zg, const int own[6], const int box[6], int thin,
                                        const uint32_t* d_prev_slots, float max_update_norm, int compact, void* stream);
int sobfu_hip_tile3_apply_tsdf_only(const float* d_phi1, int Xg, int Yg, int Zg, float* d_out1, const float* d_psi3, int Lx, int Ly, int Lz,
                                    void* stream);
/* Halo messages: n_boxes boxes (6 ints each) of a 12-byte (compact) field <-> consecutive segments of d_buf, x fastest inside a
 * box -- the send side packs the cells a neighbour needs, the receive side scatters them into its halo cells (<= 18 boxes). */
int sobfu_hip_tile3_pack(const float* d_field3, int Lx, int Ly, int Lz, float* d_buf, const int* boxes, int n_boxes, void* stream);
int sobfu_hip_tile3_unpack(float* d_field3, int Lx, int Ly, int Lz, const float* d_buf, const int* boxes, int n_boxes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * solver handle  (sobfu::cuda::Solver, include/sobfu/solver.hpp:52-101, src/sobfu/solver.cpp:7-101)
 * ---------------------------------------------------------------------------------------------------- */
typedef struct sobfu_hip_solver sobfu_hip_solver; /* opaque */

typedef struct {
    int verbosity;         /* 0 quiet; 1 energies on iterations {1, k*50, max_iter}; 2 every iteration */
    int max_iter;
    int s;                 /* Sobolev filter length (only 7 taps reach the kernels, solver.cu:211-234) */
    float max_update_norm; /* convergence threshold: break when max ||update|| <= this (solver.cu:183) */
    float lambda;
    float alpha;
    float w_reg;
} sobfu_hip_solver_params; /* = SolverParams, include/sobfu/solver.hpp:16-19 */

typedef struct {
    int iterations;        /* iterations executed (iter at break, or max_iter) */
    int converged;         /* 1 if the max-update-norm test fired */
    float last_max_update_norm;
    float last_max_update_index; /* float-encoded linear voxel index of the last iteration, when that iteration reported (verbosity 2; 1, 50 k, max_iter at verbosity 1: solver.cu:173); else NaN */
    float last_e_data, last_e_reg; /* energies of the last reporting iteration (verbosity > 0), else NaN */
} sobfu_hip_solver_report;

/* Allocates the per-solver device workspace (the reference's SpatialGradients + Reductor, minus the fields
 * it never touches: nabla_phi_n, J_inv, L_o_psi_inv -- src/sobfu/vector_fields.cpp:152-161). */
int sobfu_hip_solver_create(sobfu_hip_solver** out, int X, int Y, int Z, const sobfu_hip_solver_params* params);
int sobfu_hip_solver_destroy(sobfu_hip_solver* s);
int sobfu_hip_solver_set_params(sobfu_hip_solver* s, const sobfu_hip_solver_params* params);
/* Bytes of device memory held by the handle. */
size_t sobfu_hip_solver_workspace_bytes(const sobfu_hip_solver* s);
/* Solver::estimate_psi (src/sobfu/solver.cpp:69-101 -> src/sobfu/cuda/solver.cu:85-205).  Mutates psi,
 * psi_inv, phi_n_psi, phi_global_psi_inv; reads phi_global, phi_n.  Synchronises `stream` before returning
 * (the reference synchronises every iteration).  per_iter_max_norm (host, may be NULL): max_iter floats. */
int sobfu_hip_solver_estimate_psi(sobfu_hip_solver* s, const float* d_phi_global, float* d_phi_global_psi_inv,
                                  const float* d_phi_n, float* d_phi_n_psi, float* d_psi, float* d_psi_inv,
                                  sobfu_hip_solver_report* report, float* per_iter_max_norm, void* stream);
/* Only the gradient-descent loop (solver.cu:106-193), no inverse / canonical warp: the unit bench.py times.
 * Always runs exactly n_iters iterations when max_update_norm < 0. */
int sobfu_hip_solver_iterate(sobfu_hip_solver* s, const float* d_phi_global, const float* d_phi_n,
                             float* d_phi_n_psi, float* d_psi, int n_iters, sobfu_hip_solver_report* report,
                             float* per_iter_max_norm, void* stream);
/* The same loop in pieces (quiet solves only; SOBFU_E_UNSUPPORTED when verbosity > 0).  begin: the warp of solver.cu:106 and the
 * entry into the iteration format, room for max_iters iterations.  step: ENQUEUES n_iters more iterations and returns without
 * synchronising (the device-side gate turns every launch after the reference's `break` into a no-op).  end: synchronises, finds
 * the iteration the reference stops at, rebuilds psi / phi_n_psi exactly as iterate() leaves them, fills report and
 * per_iter_max_norm (max_iters floats, may be NULL).  One open session per handle; the four buffers must stay valid until end. */
int sobfu_hip_solver_begin(sobfu_hip_solver* s, const float* d_phi_global, const float* d_phi_n, float* d_phi_n_psi,
                           float* d_psi, int max_iters, void* stream);
int sobfu_hip_solver_step(sobfu_hip_solver* s, int n_iters, void* stream);
int sobfu_hip_solver_end(sobfu_hip_solver* s, sobfu_hip_solver_report* report, float* per_iter_max_norm, void* stream);
/* Pointer to the `updates` buffer (Reductor::updates, src/sobfu/reductor.cpp:26); valid until destroy.  Holds
 * the last iteration's updates only when verbosity > 0 or keep_updates was set. */
float* sobfu_hip_solver_updates(sobfu_hip_solver* s);
int sobfu_hip_solver_keep_updates(sobfu_hip_solver* s, int keep);
/* Quiet solves (verbosity 0) iterate by default on a private compact copy of the state -- psi / nabla_U as 12-byte xyz
 * triples, tsdf-only 4-byte phi_global / phi_n / phi_n o psi -- and rebuild the caller's buffers after the loop
 * (76 instead of 112 bytes per voxel-iteration, identical results).  enable = 0 iterates directly on the API buffers. */
int sobfu_hip_solver_set_compact(sobfu_hip_solver* s, int enable);
/* Per-kernel timing of the quiet path: HIP events recorded on the solver's stream around the pass A / pass B launches of
 * every stride-th iteration (0 = off, 1 = every iteration).  An event between two kernels drains the pipeline (a few
 * microseconds), so time a profiled run for its kernel split, not for its wall time.  Totals and the number of timed
 * iterations accumulate until reset; call get_profile only after the stream has been synchronised. */
int sobfu_hip_solver_set_profiling(sobfu_hip_solver* s, int stride);
int sobfu_hip_solver_get_profile(sobfu_hip_solver* s, float* ms_pass_a, float* ms_pass_b, int* launches, int reset);
/* Callback invoked by estimate_psi for every line the reference prints with std::cout (solver.cu:115-190);
 * NULL (default) = print to stdout like the reference. */
typedef void (*sobfu_hip_log_fn)(const char* line, void* user);
int sobfu_hip_solver_set_logger(sobfu_hip_solver* s, sobfu_hip_log_fn fn, void* user);

/* ------------------------------------------------------------------------------------------------------
 * marching cubes -- include/kfusion/internal.hpp:213-225, src/kfusion/cuda/marching_cubes.cu
 * `occupied`: 3 rows of `stride` ints on the device -- voxel index, vertex count, vertex offset (the reference's
 * DeviceArray2D<int>(3, cols)).  Unlike the reference (atomic append, run-dependent order) cells come out in ascending
 * voxel-index order.  Vertices / normals are float4 (x, -y, -z, 1), three per triangle, one normal per triangle.
 * ---------------------------------------------------------------------------------------------------- */
/* d_workspace / workspace_bytes: optional device scratch of >= sobfu_hip_mc_workspace_bytes(X, Y, Z) bytes that the caller
 * keeps between calls (kfusion::cuda::MarchingCubes owns one); NULL or too small = the scratch is allocated and freed inside
 * the call (two implicit device synchronisations per call). */
size_t sobfu_hip_mc_workspace_bytes(int X, int Y, int Z);
/* getOccupiedVoxels (marching_cubes.cu:143-163): rows 0 and 1; *h_count = min(active cells, max_size).  Synchronises.
 * X*Y*Z > INT32_MAX: SOBFU_E_UNSUPPORTED (int voxel indices, as in the reference). */
int sobfu_hip_mc_occupied_voxels(void* stream, const float* d_vol, int X, int Y, int Z, int* d_occupied, int stride, int max_size,
                                 int* h_count, void* d_workspace, size_t workspace_bytes);
/* computeOffsetsAndTotalVertices (marching_cubes.cu:165-181): row 2 = exclusive scan of row 1.  Synchronises. */
int sobfu_hip_mc_offsets(void* stream, int* d_occupied, int stride, int count, int* h_total_vertices, void* d_workspace,
                         size_t workspace_bytes);
/* generateTriangles (marching_cubes.cu:275-313); pose = volume -> world (R row-major, t).  Triangles that would end beyond
 * max_vertices are dropped (the reference does not check). */
int sobfu_hip_mc_generate_triangles(void* stream, const float* d_vol, int X, int Y, int Z, const int* d_occupied, int stride, int count,
                                    float size_x, float size_y, float size_z, const float R[9], const float t[3], float* d_vertices,
                                    float* d_normals, int max_vertices);

/* ------------------------------------------------------------------------------------------------------
 * native multi-GPU loop: one rank per z-slab, RCCL halo exchange issued from C++ and overlapped with the interior
 * compute (no reference counterpart; schedule documented in sobfu_amd/tiled.py)
 * ---------------------------------------------------------------------------------------------------- */
#define SOBFU_E_RCCL (-4) /* RCCL not loaded / an RCCL call failed (details on stderr) */
typedef struct sobfu_hip_tiled sobfu_hip_tiled; /* opaque */
/* dlopen()s the RCCL library the host process already uses (e.g. <torch>/lib/librccl.so); must precede the rest. */
int sobfu_hip_tiled_load_rccl(const char* librccl_path);
/* ncclGetUniqueId on one rank; the caller broadcasts the 128 bytes to all ranks (any transport). */
int sobfu_hip_tiled_unique_id(char out[128]);
/* Collective over all `world` ranks (ncclCommInitRank).  The volume's Z planes are split as evenly as possible.
 * An all-zero unique_id creates a communicator-less handle (no collective): the slab layout and launch schedule of
 * (world, rank) with the transport left to sobfu_hip_tiled_set_transport -- or to nobody, for compute-only timing. */
int sobfu_hip_tiled_create(sobfu_hip_tiled** out, int X, int Y, int Z, int world, int rank, const char unique_id[128],
                           const sobfu_hip_solver_params* params);
int sobfu_hip_tiled_destroy(sobfu_hip_tiled* t);
/* The same for a Px x Py x Pz grid of tiles (rank = cx + Px * (cy + Py * cz); the cells of every axis are split as evenly as
 * possible; every split axis needs >= 4 cells per tile).  Px = Py = 1 is the z-slab layout of sobfu_hip_tiled_create. */
int sobfu_hip_tiled_create3(sobfu_hip_tiled** out, int X, int Y, int Z, int Px, int Py, int Pz, int rank, const char unique_id[128],
                            const sobfu_hip_solver_params* params);
/* out[24] = per axis (x, y, z): tile grid P, tile coordinates c, owned global range [g0, g1), halo cells lo / hi, local extent L,
 * global coordinate `base` of local cell 0 -- eight triples in that order. */
int sobfu_hip_tiled_layout3(const sobfu_hip_tiled* t, int out[24]);
/* owned planes [z0, z1) of this rank, halo planes below / above, slab thickness Lz, global z of local plane 0 */
int sobfu_hip_tiled_layout(const sobfu_hip_tiled* t, int* z0, int* z1, int* lo, int* hi, int* Lz, int* zbase);
/* n_iters iterations on this rank's slab (collective).  Local slabs: phi_global / phi_n o psi float2 (X, Y, Lz), psi
 * float4 (X, Y, Lz) exact on owned +-1 planes on entry and exit; phi_n: the whole float2 (X, Y, Z) volume. */
int sobfu_hip_tiled_iterate(sobfu_hip_tiled* t, const float* d_phi_global_local, const float* d_phi_n_full,
                            float* d_phi_n_psi_local, float* d_psi_local, int n_iters, sobfu_hip_solver_report* report,
                            float* per_iter_max_norm, void* stream);
/* The same loop in pieces, as sobfu_hip_solver_begin / step / end (collective; step ENQUEUES n_iters more iterations and returns
 * without synchronising; end synchronises, reduces the max-norm rows the loop has not yet made global, finds the iteration the
 * reference stops at and rebuilds the caller's arrays). */
int sobfu_hip_tiled_begin(sobfu_hip_tiled* t, const float* d_phi_global_local, const float* d_phi_n_full, float* d_phi_n_psi_local,
                          float* d_psi_local, int max_iters, void* stream);
int sobfu_hip_tiled_step(sobfu_hip_tiled* t, int n_iters, void* stream);
int sobfu_hip_tiled_end(sobfu_hip_tiled* t, sobfu_hip_solver_report* report, float* per_iter_max_norm, void* stream);
/* Optional: a second communicator (a second ncclGetUniqueId, broadcast like the first) and a stream of its own for the max-norm
 * all-reduce of a live threshold.  With it the reduction of iteration k's row is issued right after that iteration's pass B and
 * runs beside iteration k+1 (the late gate only needs it by pass B of k+2) without ever queueing behind a halo exchange on the
 * main communicator; without it the reduction shares the exchange's communicator (comm stream, or in line when serial). */
int sobfu_hip_tiled_add_reduce_comm(sobfu_hip_tiled* t, const char unique_id[128]);
/* How a Z-SLAB iteration is issued on the RCCL / callback transports (the results never depend on it; 3-D tiles and the direct
 * transport have one way): 0 = built-in heuristic (by slab thickness), 1 = exchange overlapped with the interior compute, pass A
 * split into boundary + interior launches, 2 = overlapped, pass A in one launch, 3 = serial (pass A, exchange, pass B in line on one
 * stream: no cross-stream events -- the better choice when the exchange is fast).  Which one wins depends on the machine's
 * exchange latency; sobfu_amd.tiled.NativeTiledSolver.autotune times them. */
int sobfu_hip_tiled_set_schedule(sobfu_hip_tiled* t, int schedule);
/* diagnostics: host microseconds per iteration the last sobfu_hip_tiled_iterate spent ISSUING its loop (launches, events,
 * RCCL calls) -- against the measured time per iteration it tells whether a thin slab is host-bound */
double sobfu_hip_tiled_last_enqueue_us(const sobfu_hip_tiled* t);
/* One halo message: `count` floats from d_send + send_off of this rank to d_recv + recv_off of rank `peer`'s matching message
 * (every pair of neighbours exchanges exactly one message in each direction per exchange). */
typedef struct {
    int peer;
    size_t send_off, recv_off, count; /* in floats */
} sobfu_hip_tiled_msg;
/* The messages of one exchange and the boxes (6 ints each: x0, x1, y0, y1, z0, z1, local cells) they are packed from / scattered
 * to; returns their number (z-slabs on the RCCL transport send the same cells as plane ranges of the field itself).
 * WHAT ACTUALLY TRAVELS on the RCCL / callback transports of a 3-D tile: only the first *n_packed of these messages (x / y faces and
 * all edge strips) go through the packed send / receive buffers with the offsets and counts listed here.  The z-face messages that
 * follow them in this list describe the cells (their boxes are right), but they are NOT packed: they travel IN PLACE as 4 whole
 * padded planes of the nabla_U array (Lx x Ly x 4 cells) -- the list sobfu_hip_tiled_messages_inplace returns, with offsets into the
 * field itself.  A transport that pre-posts its requests (MPI persistent requests, ...) must build them from BOTH lists. */
int sobfu_hip_tiled_messages(const sobfu_hip_tiled* t, sobfu_hip_tiled_msg* msgs, int* send_boxes, int* recv_boxes, int max_msgs);
/* The in-place list of one exchange (3-D tiles: the z faces; empty for z-slabs, whose plane messages are built per call) and, in
 * *n_packed, how many messages of sobfu_hip_tiled_messages go through the buffers; returns the length of the in-place list. */
int sobfu_hip_tiled_messages_inplace(const sobfu_hip_tiled* t, sobfu_hip_tiled_msg* msgs, int max_msgs, int* n_packed);
/* Timing of the SERIAL schedule's pieces with HIP events on the loop's stream around every stride-th iteration (0 = off;
 * max_samples events sets are created here, outside any timed region): ms[0] pass A, ms[1] the exchange (pack + transfer +
 * unpack, including the wait for the peers), ms[2] pass B -- sums over `samples` iterations since the last reset.  Call
 * get_profile after the stream has been synchronised.  An event between two kernels drains the pipeline: time a profiled run for
 * its split, not for its wall time. */
int sobfu_hip_tiled_set_profiling(sobfu_hip_tiled* t, int stride, int max_samples);
int sobfu_hip_tiled_get_profile(sobfu_hip_tiled* t, float ms[3], int* samples, int reset);
/* Pluggable transport for communicator-less handles (MPI, an in-process loopback for tests, ...).  `exchange` must deliver every
 * message: msgs[i].count floats at d_send + msgs[i].send_off arrive at d_recv + recv_off of the message rank msgs[i].peer posts
 * for this rank; `allreduce_max` must leave the element-wise maximum over all ranks in d_buf.  Both are called on the host in
 * launch order and must order their work after everything already enqueued on `stream` and before anything enqueued on it later.
 * CALLS PER EXCHANGE: z-slabs: one, with d_send == d_recv == the field (planes in place).  3-D tiles: UP TO TWO -- first the packed
 * list (d_send = the send buffer, d_recv = the receive buffer, the first n_packed messages of sobfu_hip_tiled_messages), then the
 * in-place list (d_send == d_recv == the nabla_U array, the messages of sobfu_hip_tiled_messages_inplace: whole-plane offsets and
 * counts); a list that is empty is not called.  A call never carries more than one message per peer; the two calls of one exchange
 * may name the same peer.  Every rank makes the same sequence of calls, so a transport may match them by call order. */
typedef int (*sobfu_hip_tiled_exchange_fn)(void* ctx, int rank, const float* d_send, float* d_recv, const sobfu_hip_tiled_msg* msgs,
                                           int n_msgs, void* stream);
typedef int (*sobfu_hip_tiled_allreduce_fn)(void* ctx, int rank, uint32_t* d_buf, size_t n, void* stream);
int sobfu_hip_tiled_set_transport(sobfu_hip_tiled* t, sobfu_hip_tiled_exchange_fn exchange, sobfu_hip_tiled_allreduce_fn allreduce_max,
                                  void* ctx);
/* DIRECT transport (communicator-less handles): the halo cells travel as plain stores from pass A's launch into the neighbours'
 * own nabla_U arrays, peer-mapped over xGMI -- no pack / unpack kernels, no communication launch, no collective in the loop; arrival
 * flags and the max-norm rows travel the same way (see sobfu_amd/csrc/tiled_capi.hip).  Every rank exports two device
 * allocations (sobfu_hip_tiled_exports_get; across processes: sobfu_hip_ipc_export -> 64-byte handles -> sobfu_hip_ipc_open on the
 * other side), hands the pointers of ALL other ranks -- valid in ITS process -- to sobfu_hip_tiled_connect, and every rank must have
 * connected (a barrier of the caller's) before any begins a solve.  A peer whose flag does not arrive within
 * SOBFU_TILED_DEADLINE_S (default 30 s) is recorded instead of waited for: _end / _status return SOBFU_E_TIMEOUT. */
#define SOBFU_E_TIMEOUT (-5) /* a peer did not answer within the deadline; the handle is dead (destroy it) */
typedef struct {
    void* arena;           /* ONE allocation (what crosses the process boundary): the two halves of the double-buffered nabla_U tile
                              (12-byte cells, local extents) and the global max-norm rows (256 uint32 per iteration) ... */
    size_t nabla_u_off[2]; /* ... at these byte offsets */
    size_t rows_off;
    void* flags;           /* a second allocation: arrival flags, one uint32 per rank (uncached) */
} sobfu_hip_tiled_exports;
int sobfu_hip_tiled_exports_get(const sobfu_hip_tiled* t, sobfu_hip_tiled_exports* out);
int sobfu_hip_tiled_connect(sobfu_hip_tiled* t, int n_peers, const int* peer_ranks, const sobfu_hip_tiled_exports* peers);
int sobfu_hip_ipc_export(const void* d_ptr, char handle[64]);
int sobfu_hip_ipc_open(const char handle[64], void** d_ptr);
int sobfu_hip_ipc_close(void* d_ptr);
/* 0, or SOBFU_E_TIMEOUT with the rank that was missing (-1: unknown) */
int sobfu_hip_tiled_status(sobfu_hip_tiled* t, int* missing_peer);
/* Iterations ONE solve may run on this handle: the direct transport's max-norm rows are mapped by the peers, so their number is
 * fixed when the handle is created (16384); other transports grow their rows on demand. */
int sobfu_hip_tiled_max_iterations(const sobfu_hip_tiled* t);
/* What the runtime says about the path from `device` to `peer_device`: out = {hipDeviceCanAccessPeer, link type
 * (hipExtGetLinkTypeAndHopCount: 0 HyperTransport, 1 QPI, 2 PCIe, 3 InfiniBand, 4 xGMI), hops, hipDevP2PAttrPerformanceRank}; -1 = unknown. */
int sobfu_hip_p2p_info(int device, int peer_device, int out[4]);
/* Diagnostics of the direct transport (every rank of the world makes the same calls in the same order, outside a solve):
 *   wait_stats  microseconds the signalling workgroup of pass A has spent waiting for its peers' arrival flags, and how many waits --
 *               the part of the exchange an iteration did NOT hide behind its own compute
 *   pingpong    `reps` flag round trips between rank_a and rank_b inside one launch (the other ranks only advance their sequence numbers)
 *   probe_push  pass A's push boxes alone (the production store path into the peers' halo cells + signalling), `reps` launches */
int sobfu_hip_tiled_wait_stats(sobfu_hip_tiled* t, double* wait_us, int* waits, int reset);
int sobfu_hip_tiled_pingpong(sobfu_hip_tiled* t, int rank_a, int rank_b, int reps, void* stream);
int sobfu_hip_tiled_probe_push(sobfu_hip_tiled* t, int reps, void* stream);
/* test hooks: wait = 0 turns the in-kernel wait for the peers' flags off (a harness that drives N ranks from ONE process steps
 * them phase by phase with host barriers instead: phase 0 = pass A incl. the pushes, phase 1 = pass B; after the last iteration
 * phase 2 = the end-of-solve handshake, then sobfu_hip_tiled_end) */
int sobfu_hip_tiled_set_wait(sobfu_hip_tiled* t, int wait);
int sobfu_hip_tiled_step_phase(sobfu_hip_tiled* t, int phase, void* stream);
/* bring-up helpers: the loop's halo exchange on a caller-provided 12-byte tile field (planes < 4: z-slabs only); a self
 * send/recv and a MAX all-reduce through the same RCCL entry points (usable with a single rank) */
int sobfu_hip_tiled_exchange(sobfu_hip_tiled* t, float* d_field3, int planes, void* stream);
int sobfu_hip_tiled_self_sendrecv(sobfu_hip_tiled* t, const float* d_src, float* d_dst, size_t n, void* stream);
int sobfu_hip_tiled_allreduce_max_u32(sobfu_hip_tiled* t, uint32_t* d_buf, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* SOBFU_HIP_H */
