// Internal launchers shared by solver_capi.hip / tiled_capi.hip (defined in solver_kernels.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace sobfu_hip {
// A box of cells [x0, x1) x [y0, y1) x [z0, z1) of the (local) array a fused pass produces; direct: a THIN box, evaluated one
// lane per cell straight from the caches instead of by a z-march (the one-cell shells of a tile, halo messages).  Up to 6 boxes
// per launch; empty boxes are skipped.
struct LaunchBox {
    int x0, x1, y0, y1, z0, z1;
    bool direct;
};
// A box of a tile's pass A.  dst != null: a PUSH box -- the cells of one halo message, whose results go to
// dst + 3 * ((x + ox) + px * ((y + oy) + py * (z + oz))): the neighbour's halo cells (peer-mapped) or a packed send buffer.
// A MARCHING push box may be larger than its message and serve the owned block too (which then leaves those cells out): only rows
// push_y0 <= y < push_y1 travel, and cells of planes local_z0 <= z < local_z1 are ALSO stored into this rank's own nabla_U.
struct TileLaunchBox {
    LaunchBox box;
    float* dst;
    int ox, oy, oz, px, py;
    int push_y0, push_y1, local_z0, local_z1;
};
// (X, Y, Z): extents of the field arrays; (pX, pY, pZ): extents of phi_n (the whole volume); own: the cells that enter the
// max-norm (x0, x1, y0, y1, z0, z1).
int launch_pass_a_boxes(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z, const LaunchBox* boxes,
                        int n, const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact);
// Signalling state of the direct transport (device memory, one per tiled handle; filled by sobfu_hip_tiled_connect and read by
// the tail of tile_potential_gradient_kernel).
constexpr int kMaxSync = 64;
struct TileSync {
    uint32_t ticket_push, reserved;    // 0 at rest
    uint64_t wait_ticks;               // diagnostics: wall_clock64 ticks the signalling workgroup has spent waiting for peers' flags ...
    uint32_t wait_count, pad0;         // ... over this many waits (sobfu_hip_tiled_wait_stats)
    uint32_t err;                      // 0, or 1 + the rank whose flag did not arrive before the deadline
    uint32_t n_sync;                   // ranks this rank signals and waits for
    uint32_t my_rank, world;
    uint64_t timeout_ticks;            // wall_clock64 ticks (100 MHz)
    uint32_t* my_flags;                // [world] arrival flags, written by the peers
    uint32_t* my_grows;                // global max-norm rows [iteration][256], entry q written by rank q
    int sync_rank[kMaxSync];
    uint32_t* peer_flags[kMaxSync];    // the flags array of sync_rank[i]
    uint32_t* peer_grows[kMaxSync];    // its global rows
};
// sync: device pointer to the handle's TileSync (null: no signalling); seq / wait / row / row_index: see TilePassAArgs
int launch_tile_pass_a(const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, int X, int Y, int Z, const TileLaunchBox* boxes,
                       int n, TileSync* sync, uint32_t seq, int wait, const uint32_t* row, uint32_t row_index, int zc, hipStream_t stream, bool compact);
// the same launch planned once (compact format): the box list is uploaded to device memory at plan time, a launch passes ~100 bytes
struct TilePassAPlan;
int tile_pass_a_plan_create(TilePassAPlan** out, const TileLaunchBox* boxes, int n, int X, int Y, int Z);
void tile_pass_a_plan_destroy(TilePassAPlan* p);
int launch_tile_pass_a_plan(const TilePassAPlan* p, const float* pnp, const float* pg, const float* psi, float* nU, float w_reg, TileSync* sync,
                            uint32_t seq, int wait, const uint32_t* row, uint32_t row_index, hipStream_t stream);
// end of a solve on the direct transport: push the last max-norm row (row may be null), raise the flags with `seq`, wait
int launch_tile_flush(TileSync* sync, uint32_t seq, int wait, const uint32_t* row, uint32_t row_index, hipStream_t stream);
// diagnostics: `reps` flag round trips between this rank and sync-set member `q` (both sides launch it; `first` serves); sequence
// numbers seq0 .. seq0 + 2 * reps - 1
int launch_tile_pingpong(TileSync* sync, int q, int first, uint32_t seq0, int reps, hipStream_t stream);
int launch_pass_b_boxes(const float* nU, float* psi, const float* phi_n, float* pnp, float* updates, uint32_t* slots, const float taps[7],
                        float alpha, int X, int Y, int Z, int pX, int pY, int pZ, const int own[6], const LaunchBox* boxes, int n,
                        const uint32_t* prev_slots, float max_update_norm, int zc, hipStream_t stream, bool compact,
                        float* psi_out = nullptr /* null: update psi in place */, int prev_rows = 1,
                        bool sys_acquire = false /* the launch reads cells other GPUs stored: nabla_U and the max-norm rows are READ AT SYSTEM SCOPE
                                                     (sc0 sc1 loads of the pipelined march / the thin shells / the gate; there is no invalidate -- + 39 us,
                                                     measured); a launch that cannot take that march is refused with SOBFU_E_UNSUPPORTED */);
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
// the same scatter by a precomputed table (device memory): cell c of buf -> cell d_table[c] of the field
int launch_msg_scatter_table(float* field3, const float* buf, const uint32_t* d_table, unsigned n_cells, hipStream_t stream);
// whole-volume enter / leave of the compact format in one pass each (solver handle)
int launch_compact_enter(const float* psi4, const float* pg2, const float* pn2, float* c_psi, float* c_g, float* c_n, float* c_f, int X, int Y,
                         int Z, hipStream_t stream);
int launch_compact_leave(const float* c_psi, const float* pn2, float* psi4, float* pnp2, int X, int Y, int Z, hipStream_t stream);
// the two energies of solver.cu:132-142 straight from the iteration format (reduce_kernels.hip): same tree, same values as
// sobfu_hip_data_energy / sobfu_hip_reg_energy_sobolev_from_psi on the API-format arrays
int data_energy_tsdf(const float* g1, const float* f1, int n, void* d_scratch, float* out, hipStream_t stream);
int reg_energy_from_psi3(const float* psi3, int X, int Y, int Z, void* d_scratch, float* out, hipStream_t stream);
}  // namespace sobfu_hip
