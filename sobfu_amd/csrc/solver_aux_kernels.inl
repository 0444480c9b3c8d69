// Part of solver_kernels.hip (included there, inside its anonymous namespace; not a translation unit of its own): once-per-solve conversions of the compact format and the halo-message copy / scatter kernels of a 3-D tile
// clang-format off: the include order in solver_kernels.hip matters (common -> pass A -> pass B -> aux)

// --- compact-format conversions (once per solve, not per iteration) ----------------------------------------------
__global__ void __launch_bounds__(256) pack_vec_kernel(const float4* __restrict__ src, P3* __restrict__ dst, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    stv<true>(dst, i, src[i]);
}
// writes xyz back into the API float4 field; w is left untouched, as update_psi_kernel leaves it (utils.hpp:260-265)
__global__ void __launch_bounds__(256) unpack_vec_kernel(const P3* __restrict__ src, float4* __restrict__ dst, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 v = ldv<true>(src, i);
    *(v3f_u*) ((float*) (dst + i)) = v3f{v.x, v.y, v.z};
}
__global__ void __launch_bounds__(256) extract_tsdf_kernel(const float2* __restrict__ src, float* __restrict__ dst, size_t N) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) dst[i] = src[i].x;
}
__global__ void __launch_bounds__(256) apply_tsdf_only_kernel(const float* __restrict__ phi, float* __restrict__ out,
                                                              const P3* __restrict__ psi, Dims d, Dims pd) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z;
    if (x >= d.x || y >= d.y) return;
    size_t i = vidx(d, x, y, z);
    float4 p = ldv<true>(psi, i);
    out[i]   = interp_tsdf_only(phi, pd, p.x, p.y, p.z);
}

// Entering / leaving the compact format in ONE pass each (the solver handle's whole-volume case; the slab loop keeps the
// separate kernels because its phi_n is a different, larger array than its slab fields):
//   enter: psi float4 -> 12-byte psi, tsdf channels of phi_global / phi_n, F = interpolate_tsdf(phi_n, psi).tsdf (solver.cu:106)
//   leave: 12-byte psi -> psi.xyz (w untouched), phi_n o psi = interpolate_tsdf(phi_n, psi) (the state solver.cu:168 leaves)
__global__ void __launch_bounds__(256) compact_enter_kernel(const float4* __restrict__ psi4, const float2* __restrict__ pg2,
                                                            const float2* __restrict__ pn2, P3* __restrict__ c_psi, float* __restrict__ c_g,
                                                            float* __restrict__ c_n, float* __restrict__ c_f, Dims d) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z;
    if (x >= d.x || y >= d.y) return;
    const size_t i = vidx(d, x, y, z);
    const float4 p = psi4[i];
    stv<true>(c_psi, i, p);
    c_g[i] = pg2[i].x;
    c_n[i] = pn2[i].x;
    c_f[i] = interp_tsdf(pn2, d, p.x, p.y, p.z).x;  // same lerp chain on the same tsdf values as interp_tsdf_only on c_n
}
__global__ void __launch_bounds__(256) compact_leave_kernel(const P3* __restrict__ c_psi, const float2* __restrict__ pn2,
                                                            float4* __restrict__ psi4, float2* __restrict__ pnp2, Dims d) {
    const int x = blockIdx.x * kBX + threadIdx.x, y = blockIdx.y * kBY + threadIdx.y, z = blockIdx.z;
    if (x >= d.x || y >= d.y) return;
    const size_t i = vidx(d, x, y, z);
    const float4 p = ldv<true>(c_psi, i);
    *(v3f_u*) ((float*) (psi4 + i)) = v3f{p.x, p.y, p.z};
    pnp2[i] = interp_tsdf(pn2, d, p.x, p.y, p.z);
}

// --- halo messages of a 3-D tile ------------------------------------------------------------------------------------
// A message is a box of cells of a 12-byte field, laid out x fastest in a contiguous buffer segment.  One launch packs (or
// unpacks) all messages of an exchange: one thread per cell, the message found by a scan of <= 18 prefix entries.
struct MsgBoxes {
    int n;
    int x0[kMaxMsgs], y0[kMaxMsgs], z0[kMaxMsgs], nx[kMaxMsgs], ny[kMaxMsgs];
    unsigned first[kMaxMsgs + 1];  // first cell of message i in the buffer; first[n] = cells in all messages
};
template <bool PACK>
__global__ void __launch_bounds__(256) msg_copy_kernel(float* __restrict__ field3, float* __restrict__ buf, Dims d, MsgBoxes m) {
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= m.first[m.n]) return;
    int x0 = m.x0[0], y0 = m.y0[0], z0 = m.z0[0], nx = m.nx[0], ny = m.ny[0];
    unsigned first = 0;
#pragma unroll
    for (int k = 1; k < kMaxMsgs; ++k)
        if (k < m.n && c >= m.first[k]) {
            x0 = m.x0[k]; y0 = m.y0[k]; z0 = m.z0[k]; nx = m.nx[k]; ny = m.ny[k];
            first = m.first[k];
        }
    const unsigned e = c - first;
    const int ix = (int) (e % (unsigned) nx), iy = (int) ((e / (unsigned) nx) % (unsigned) ny), iz = (int) (e / ((unsigned) nx * (unsigned) ny));
    const size_t i = vidx(d, x0 + ix, y0 + iy, z0 + iz);
    if (PACK) stv<true>(buf, c, ldv<true>(field3, i));
    else stv<true>(field3, i, ldv<true>(buf, c));
}

// The scatter of an exchange's packed messages by a precomputed TABLE: cell c of the receive buffer goes to cell table[c] of the field.
// The loop issues the same scatter every iteration, so the message scan, the three integer divisions per cell and the 500-byte argument
// block of msg_copy_kernel are paid once, at handle creation: what is left is two independent loads and a store per cell.
__global__ void __launch_bounds__(256) msg_scatter_table_kernel(float* __restrict__ field3, const float* __restrict__ buf, const uint32_t* __restrict__ table,
                                                                unsigned n) {
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n) return;
    stv<true>(field3, table[c], ldv<true>(buf, c));
}
