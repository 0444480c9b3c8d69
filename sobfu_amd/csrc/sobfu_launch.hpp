// Internal launchers shared by solver_capi.hip / tiled_capi.hip (defined in solver_kernels.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace sobfu_hip {
// A box of cells [x0, x1) x [y0, y1) x [z0, z1) of the (local) array a fused pass produces; tr: the 64 lanes of a wave run
// along y instead of x (boxes that are thin in x).  Up to 6 boxes per launch; empty boxes are skipped.
struct LaunchBox {
    int x0, x1, y0, y1, z0, z1;
    bool tr;
};
// (X, Y, Z): extents of the field arrays; (pX, pY, pZ): extents of phi_n (the whole volume); own: the cells that enter the
// max-norm (x0, x1, y0, y1, z0, z1).
int launch_pass_a_boxes(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z, const LaunchBox* boxes,
                        int n, const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact);
int launch_pass_b_boxes(const float* nU, float* psi, const float* phi_n, float* pnp, float* updates, uint32_t* slots, const float taps[7],
                        float alpha, int X, int Y, int Z, int pX, int pY, int pZ, const int own[6], const LaunchBox* boxes, int n,
                        const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact,
                        float* psi_out = nullptr /* null: update psi in place */, int prev_rows = 1);
// Pass A / pass B over planes [z_lo, z_hi) (z_hi <= 0: the whole grid) and, optionally, a second range [z_lo2, z_hi2)
// in the same launch (both boundary regions of a multi-GPU slab).  zc <= 0: z-chunk chosen by the cost model.
int launch_pass_a(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z,
                  const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact, int z_lo = 0, int z_hi = 0,
                  int z_lo2 = 0, int z_hi2 = 0);
int launch_pass_b(const float* nU, float* psi, const float* phi_n, float* pnp, float* updates, uint32_t* slots,
                  const float taps[7], float alpha, int X, int Y, int Z, const uint32_t* prev_slots,
                  float max_update_norm, int zc, hipStream_t stream, int phi_Z, int own_lo, int own_hi, bool compact, int z_lo = 0,
                  int z_hi = 0, int z_lo2 = 0, int z_hi2 = 0, float* psi_out = nullptr /* null: update psi in place */,
                  int prev_rows = 1 /* rows the gate reads, see solver_converged */);
int launch_pack_vec(const float* src4, float* dst3, size_t N, hipStream_t stream);
int launch_unpack_vec(const float* src3, float* dst4, size_t N, hipStream_t stream);
int launch_extract_tsdf(const float* src2, float* dst1, size_t N, hipStream_t stream);
// phi_{X,Y,Z} > 0: extents of phi1 when it is the whole volume and (X, Y, Z) a tile of it
int launch_apply_tsdf_only(const float* phi1, float* out1, const float* psi3, int X, int Y, int Z, hipStream_t stream, int phi_Z = 0, int phi_X = 0,
                           int phi_Y = 0);
// halo messages of a 3-D tile: n boxes (6 ints each: x0, x1, y0, y1, z0, z1) of a 12-byte field <-> consecutive buffer segments
int launch_msg_copy(bool pack, float* field3, float* buf, int Lx, int Ly, int Lz, const int* boxes, int n, hipStream_t stream);
// whole-volume enter / leave of the compact format in one pass each (solver handle)
int launch_compact_enter(const float* psi4, const float* pg2, const float* pn2, float* c_psi, float* c_g, float* c_n, float* c_f, int X, int Y,
                         int Z, hipStream_t stream);
int launch_compact_leave(const float* c_psi, const float* pn2, float* psi4, float* pnp2, int X, int Y, int Z, hipStream_t stream);
}  // namespace sobfu_hip
