// Part of solver_kernels.hip (included there, inside its anonymous namespace; not a translation unit of its own): pass B -- fused smoothing + psi update + warp + max-norm: arguments, thin-box (lane per cell) evaluation, the pipelined march of cache-resident sizes, the kernel
// clang-format off: the include order in solver_kernels.hip matters (common -> pass A -> pass B -> aux)

// --- pass B ----------------------------------------------------------------------------------------------------
struct PassBArgs {
    const void* nU;
    void* psi;
    const void* phi_n;
    void* pnp;        // phi_n o psi (output)
    float4* updates;  // may be null (always float4)
    uint32_t* slots;  // 256 x uint32, atomic max of ||u||^2 bit patterns
    Dims d;
    Taps S;
    float alpha;
    BoxList boxes;  // the cells this launch produces
    const uint32_t* prev_slots;
    float max_update_norm;
    // multi-GPU tiles: the fields are local arrays (d) that carry halo cells, phi_n is the whole volume (pd); only the
    // cells of `own` belong to this rank and enter the max-norm.  Single GPU: pd == d, own = everything.
    Dims pd;
    int own[6];     // x0, x1, y0, y1, z0, z1
    int prev_rows;  // rows the gate looks at (see solver_converged)
    void* psi_out;  // where the updated psi goes: == psi (in place) or the other half of a ping-pong pair (native tiled loop)
    int sys_acquire;  // direct transport: halo cells and max-norm entries of this launch's inputs were stored by OTHER GPUs (see the kernel's entry)
};

#ifndef SOBFU_PAIR_B
#define SOBFU_PAIR_B 1  // cache-resident launches of the pipelined pass B: z-chunks march in alternating directions (Box::pair)
#endif
#ifndef SOBFU_HLEAD
#define SOBFU_HLEAD 3  // planes the halo requests of pass B run ahead on long marches (0: never; one plane ahead, straight from registers)
#endif
#ifndef SOBFU_HLEAD_MIN_ZC
#define SOBFU_HLEAD_MIN_ZC 24  // shortest march (planes) that uses the halo lead
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef SOBFU_MINW_B
#define SOBFU_MINW_B 6  // waves/SIMD the register allocator must leave room for: <= 80 VGPR -> 3 workgroups of 8 waves per CU
#endif
#ifndef SOBFU_MINW_PIPE
#define SOBFU_MINW_PIPE 4  // the pipelined march: <= 128 VGPR -> 2 workgroups of 8 waves per CU
#endif
// HL: planes the halo requests run ahead of the plane they are staged for (0: one plane ahead, straight from registers -- short
// marches, where the extra prologue round trip costs more than the re-fetched halo lines).
// max ||u||^2 over the voxels a workgroup owns -> one atomicMax on one of 256 slots
template <int WY>
SOBFU_DEV void maxnorm_tail(float msq, uint32_t* slots, uint32_t* s_max) {
    const int lx = threadIdx.x, wy = threadIdx.y;
    uint32_t m = __float_as_uint(msq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
    if (lx == 0) s_max[wy] = m;
    __syncthreads();
    if (lx == 0 && wy == 0) {
#pragma unroll
        for (int w = 1; w < WY; ++w) m = max(m, s_max[w]);
        atomicMax(slots + (blockIdx.x & 255u), m);
    }
}

// DIRECT evaluation of one cell of pass B (thin boxes: the one-cell x / y shells of a tile): 19 nabla_U loads + psi + the
// phi_n gather, op for op the marching path's arithmetic (sum = 0; taps ascending j; (Sx + Sy) + Sz)
template <bool COMPACT, bool SYS>
SOBFU_DEV void direct_taps(const PassBArgs& a, int x, int y, int z, float& slx, float& sly, float& slz, float& srx, float& sry, float& srz, float& szx,
                           float& szy, float& szz) {
    const Dims d = a.d;
    const __amdgpu_buffer_rsrc_t r_nu = buf_rsrc(a.nU, SYS ? (uint32_t) ((size_t) d.x * d.y * d.z * 12) : 0u);
    auto ld_nu = [&](size_t i) { return SYS ? buf_ld3_scope(r_nu, (uint32_t) (i * 12), 0u, true) : ldv<COMPACT>(a.nU, i); };
    slx = sly = slz = srx = sry = srz = szx = szy = szz = 0.f;
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        const float s = a.S.s[3 - j];
        const float4 vl = ld_nu(vidx(d, min(max(x + j, 0), d.x - 1), y, z));
        slx += vl.x * s;
        sly += vl.y * s;
        slz += vl.z * s;
    }
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        const float s = a.S.s[3 - j];
        const float4 vr = ld_nu(vidx(d, x, min(max(y + j, 0), d.y - 1), z));
        srx += vr.x * s;
        sry += vr.y * s;
        srz += vr.z * s;
    }
#pragma unroll
    for (int j = -3; j <= 3; ++j) {
        const float s = a.S.s[3 - j];
        const float4 vz = ld_nu(vidx(d, x, y, min(max(z + j, 0), d.z - 1)));
        szx += vz.x * s;
        szy += vz.y * s;
        szz += vz.z * s;
    }
}
template <bool WRITE_UPDATES, bool COMPACT, bool IDX32>
SOBFU_DEV float pass_b_direct_cell(const PassBArgs& a, int x, int y, int z) {
    const Dims d = a.d;
    float slx, sly, slz, srx, sry, srz, szx, szy, szz;
    // nabla_U cells of the halo rims may have been stored by other GPUs (direct transport): the taps are read at system scope then
    if (COMPACT && a.sys_acquire != 0) direct_taps<COMPACT, COMPACT>(a, x, y, z, slx, sly, slz, srx, sry, srz, szx, szy, szz);
    else direct_taps<COMPACT, false>(a, x, y, z, slx, sly, slz, srx, sry, srz, szx, szy, szz);
    const float tx = (slx + srx) + szx, ty = (sly + sry) + szy, tz = (slz + srz) + szz;
    const size_t i = vidx(d, x, y, z);
    const float4 uu = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
    float4 p = ldv<COMPACT>(a.psi, i);
    p.x -= uu.x;
    p.y -= uu.y;
    p.z -= uu.z;
    stv<COMPACT>(a.psi_out, i, p);
    if (WRITE_UPDATES) a.updates[i] = uu;
    if (COMPACT) ((float*) a.pnp)[i] = IDX32 ? interp_tsdf_only32((const float*) a.phi_n, a.pd, p.x, p.y, p.z) : interp_tsdf_only((const float*) a.phi_n, a.pd, p.x, p.y, p.z);
    else ((float2*) a.pnp)[i] = interp_tsdf((const float2*) a.phi_n, a.pd, p.x, p.y, p.z);
    const bool owned = x >= a.own[0] && x < a.own[1] && y >= a.own[2] && y < a.own[3] && z >= a.own[4] && z < a.own[5];
    return owned ? norm_sq4(uu) : 0.f;
}

// The SOFTWARE-PIPELINED march of pass B (compact solver format, one row per lane).  In the plain march a plane's dependent
// chain ends in a memory round trip nothing hides when few workgroups share a CU (multi-GPU tiles, small grids: fewer workgroups
// than the chip has slots for): the eight phi_n corners are gathered at the coordinates the psi update has just produced, and the
// step waits for them.  Here the gather of plane z-1 is ISSUED AT THE TOP of step z, together with the step's other requests
// (psi(z), nabla_U plane z+4, the halo of plane z+1), and everything is awaited once, behind the barrier and the 63 taps:
//     requests | barrier | taps of plane z (LDS + the 7 register planes) | -- await --
//     fold the corners into phi_n o psi(z-1), psi(z) -= alpha * t, store both, shift the z pipeline, stage plane z+1 (centre +
//     halo) into the OTHER LDS buffer.
// No request is in flight across the loop's back edge (the compiler would wait for all of them there anyway, to copy the
// loop-carried registers), one barrier per plane as before (a wave writes buffer b^1 only behind the barrier that followed the
// last reads of b^1).  Same arithmetic, same bits.
template <int WY, int NTL, bool DOWN>
SOBFU_DEV void pass_b_march_pipe(const PassBArgs& a, const TileGeom& tg, const GateRegs& gate, float4 (*tile)[WY + 6][TX + 8], uint32_t* s_max) {
    constexpr int R = 3, TY = WY;
    constexpr int NXH = (2 * R * TY + TX - 1) / TX, NTASK = 2 * R + NXH, TPW = (NTASK + WY - 1) / WY;
    constexpr uint32_t VB = 12u, TB = 4u;
    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    const int u0 = tg.u0, v0 = tg.v0, zb = tg.zb, ze = tg.ze;
    const int u = u0 + lx, uc = min(u, tg.DU - 1), v = v0 + wy;
    const size_t plane = (size_t) d.x * d.y, sv = (size_t) d.x;
    const uint32_t cell = (uint32_t) ((size_t) uc + sv * (size_t) min(v, tg.DV - 1));
    const uint32_t off = cell * VB, offT = cell * TB;
    const bool mine = u < tg.u_hi && v < tg.v_hi;
    const bool owned = u >= a.own[0] && u < a.own[1] && v >= a.own[2] && v < a.own[3];
    int h_lr[TPW], h_lc[TPW];
    uint32_t h_off[TPW];
    bool h_on[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int task = wy + k * WY;
        h_on[k] = task < NTASK;
        int lr = 0, lc = 0;
        if (task < R) { lr = task; lc = lx + R; }
        else if (task < 2 * R) { lr = TY + task; lc = lx + R; }
        else {
            const int e = (task - 2 * R) * TX + lx;
            h_on[k] = h_on[k] && e < 2 * R * TY;
            const int row = e / (2 * R), c = e % (2 * R);
            lr = R + row;
            lc = c < R ? c : TX + c;
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        const int gu = min(max(u0 - R + lc, 0), tg.DU - 1), gv = min(max(v0 - R + lr, 0), tg.DV - 1);
        h_off[k] = (uint32_t) ((size_t) gu + (size_t) gv * sv) * VB;
    }
    // buffer addressing: a resource per array (SGPRs), the lane's 32-bit byte offset in the plane, the plane as a scalar byte offset
    const uint32_t plane4 = (uint32_t) plane * TB, cells = (uint32_t) plane * (uint32_t) d.z;
    const __amdgpu_buffer_rsrc_t r_nu = buf_rsrc(a.nU, cells * VB), r_psi = buf_rsrc(a.psi, cells * VB), r_out = buf_rsrc(a.psi_out, cells * VB),
                                 r_f = buf_rsrc(a.pnp, cells * TB);
    auto nU_plane = [&](int z) { return 3u * (uint32_t) min(max(z, 0), d.z - 1) * plane4; };
    // direct transport: cells of nabla_U's halo rims were stored by other GPUs -> this launch reads nabla_U at system scope (measured on
    // one GPU with every load of the march so marked: pass B 24.7 -> 24.2 us, the same fabric bytes: free)
    const bool sys = a.sys_acquire != 0;
    // The z taps live in EIGHT register slots that rotate (the loop is unrolled eight times: slot indices are constants, nothing is
    // shifted -- the seven-plane shift of the plain march is 18 register moves per plane, 9 % of the loop's vector instructions).
    // Slot m % 8 holds the m-th plane of the march's window: m = st .. st + 6 at step st, i.e. planes z - 3 .. z + 3 going up
    // (plane = z_first - 3 + m) or z + 3 .. z - 3 going DOWN (plane = z_first + 3 - m, see Box::pair); the plane requested at step st,
    // m = st + 7, takes the slot the window left a step ago.
    constexpr int DZ = DOWN ? -1 : 1;
    const int z_first = DOWN ? ze - 1 : zb, n_steps = ze - zb;
    float4 Q[8], hq[TPW];
#pragma unroll
    for (int m = 0; m < 7; ++m) Q[m] = buf_ld3_scope(r_nu, off, nU_plane(z_first + DZ * (m - 3)), sys);
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        if (h_on[k]) hq[k] = buf_ld3_scope(r_nu, h_off[k], nU_plane(z_first), sys);
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;
    tile[0][wy + R][lx + R] = Q[3];
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        if (h_on[k]) tile[0][h_lr[k]][h_lc[k]] = hq[k];
    float4 p_prev = make_float4(0.f, 0.f, 0.f, 0.f);
    float msq = 0.f;
    for (int s0 = 0; s0 < n_steps; s0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int st = s0 + i;
            if (st >= n_steps) break;
            const int z = z_first + DZ * st;
            const int buf = st & 1;
            const uint32_t zcur4 = (uint32_t) z * plane4;
            const float4 qc = Q[(i + 3) % 8];  // plane z
            // this step's requests
            const float4 pv = buf_ld3(r_psi, off, 3u * zcur4, NTL >= 2);
            if (st + 1 < n_steps) {
                Q[(i + 7) % 8] = buf_ld3_scope(r_nu, off, nU_plane(z + 4 * DZ), sys);
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (h_on[k]) hq[k] = buf_ld3_scope(r_nu, h_off[k], nU_plane(z + DZ), sys);
            }
            Gather8 g;
            if (mine && st > 0) g = gather_issue32((const float*) a.phi_n, a.pd, p_prev.x, p_prev.y, p_prev.z);
            __syncthreads();
            // the taps of plane z: x and y from the LDS tile, z from the register planes (sum = 0; ascending j; products not contracted)
            v2f l01 = {0.f, 0.f}, l23 = {0.f, 0.f}, r01 = {0.f, 0.f}, r23 = {0.f, 0.f}, z01 = {0.f, 0.f}, z23 = {0.f, 0.f};
#pragma unroll
            for (int j = -R; j <= R; ++j) {
                const v2f s2 = {a.S.s[R - j], a.S.s[R - j]};
                const float4 vl = (j == 0) ? qc : tile[buf][wy + R][lx + R + j];
                l01 += v2f{vl.x, vl.y} * s2;
                l23 += v2f{vl.z, vl.w} * s2;
                const float4 vr = (j == 0) ? qc : tile[buf][wy + R + j][lx + R];
                r01 += v2f{vr.x, vr.y} * s2;
                r23 += v2f{vr.z, vr.w} * s2;
                const float4 vz = Q[(i + 3 + DZ * j + 8) % 8];  // plane z + j
                z01 += v2f{vz.x, vz.y} * s2;
                z23 += v2f{vz.z, vz.w} * s2;
            }
            const v2f t01 = (l01 + r01) + z01;
            const float tx = t01.x, ty = t01.y, tz = (l23.x + r23.x) + z23.x;
            // update_psi_kernel (solver.cu:64-67)
            const float4 uu = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
            float4 p = pv;
            p.x -= uu.x;
            p.y -= uu.y;
            p.z -= uu.z;
            pin3(p);
            if (mine) {
                if (owned && z >= a.own[4] && z < a.own[5]) msq = fmaxf(msq, norm_sq4(uu));
                if (st > 0) {  // apply_kernel (vector_fields.cu:95-98) of the plane of the step before
                    buf_st1(r_f, offT, DOWN ? zcur4 + plane4 : zcur4 - plane4, gather_finish(g), NTL >= 1);
                }
                buf_st3(r_out, off, 3u * zcur4, p, NTL >= 1);
            }
            p_prev = p;
            if (st + 1 < n_steps) {  // stage the next plane of the march into the other buffer
                tile[buf ^ 1][wy + R][lx + R] = Q[(i + 4) % 8];
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (h_on[k]) tile[buf ^ 1][h_lr[k]][h_lc[k]] = hq[k];
            }
        }
    }
    if (ze > zb && mine) {  // the last plane's warp
        buf_st1(r_f, offT, (uint32_t) (DOWN ? zb : ze - 1) * plane4, interp_tsdf_only32((const float*) a.phi_n, a.pd, p_prev.x, p_prev.y, p_prev.z), NTL >= 1);
    }
    maxnorm_tail<WY>(msq, a.slots, s_max);
}

// DIRECT_OK: the launch may hold direct boxes (multi-GPU tiles)
// NTL: streaming hints (see pass_a_march).  PIPE: the software-pipelined march (pass_b_march_pipe).
// NTBUF (plain march, compact format, arrays below 4 GiB -- the launcher checks): the streaming hint of the 12-byte psi load / store is
// REAL.  hipcc drops the nontemporal flag of __builtin_nontemporal_load / _store on the 4-byte-aligned 12-byte vector type (found in the
// ISA in round 5: `global_load_dwordx3 ... off` without `nt`, while the 4-byte phi_n o psi store carries it); the buffer instructions
// take the hint as an operand.
template <int RPT, int WY, bool WRITE_UPDATES, bool COMPACT, bool DIRECT_OK, bool IDX32 = false, int HL = 0, int NTL = kNT, bool PIPE = false, bool NTBUF = false>
// (the API-format instantiations -- 16-byte psi / nabla_U, 8-byte volumes: the launcher-level entry point and set_compact(0) -- get the
// 128-VGPR budget: at 80 they spilled 12 - 28 B/lane to scratch)
__global__ void __launch_bounds__(TX* WY, (PIPE || !COMPACT) ? SOBFU_MINW_PIPE : SOBFU_MINW_B) fused_smooth_update_apply_kernel(PassBArgs a) {
    static_assert(!NTBUF || (COMPACT && !PIPE && NTL >= 1), "NTBUF: the plain march of the compact format with streaming hints");
    static_assert(!PIPE || (RPT == 1 && COMPACT && IDX32 && !WRITE_UPDATES && HL == 0), "the pipelined march exists for the compact solver format");
    constexpr int R = 3, TY = RPT * WY, LW = TX + 2 * R, LH = TY + 2 * R;
    static_assert(HL == 0 || HL >= 2, "the halo-lead FIFO needs a lead of >= 2 planes (a lead of 1 is the register path, HL = 0)");
    constexpr int NXH = (2 * R * TY + TX - 1) / TX;  // row-tasks for the 2R lane-halo columns
    constexpr int NTASK = 2 * R + NXH, TPW = (NTASK + WY - 1) / WY;
    __shared__ float4 tile[2][LH][LW + 2];
    __shared__ uint32_t s_max[WY];
    __shared__ P3 hfifo[HL > 0 ? HL : 1][HL > 0 ? NTASK * TX : 1];  // 12-byte entries: with the 32 KB tile, 3 workgroups still fit a CU's 160 KB

    // Direct transport (a.sys_acquire): the 4-cell halo rims of nabla_U were stored into this GPU's memory by kernels of OTHER GPUs
    // (write-through at system scope, acknowledged before their arrival flag went out; the flag was seen by this rank's pass A before
    // it retired: DESIGN.md section 6.2).  An invalidate at this kernel's entry (`buffer_inv sc0 sc1` by every wave) was built and
    // measured: + 39 us per launch on a 128^3 tile -- waves start at different times and every late invalidate throws away what the
    // early waves had fetched.  Instead the pipelined march reads nabla_U at system scope on such handles (buf_ld3_scope).
    const GateRegs gate = gate_load(a.prev_slots, a.prev_rows, a.sys_acquire != 0);

    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    // marching workgroups are XCD-swizzled among themselves; direct ones (numbered behind them) keep the dispatch order, which
    // spreads them over all XCDs -- a thin box concentrated on one XCD's 32 CUs is bound by their address units
    const bool marching_wg = (int) blockIdx.x >= a.boxes.m0 && (int) blockIdx.x < a.boxes.m1;
    const unsigned wg = (SOBFU_SWIZZLE_B && marching_wg)
                            ? (unsigned) a.boxes.m0 + xcd_swizzle(blockIdx.x - (unsigned) a.boxes.m0, (unsigned) (a.boxes.m1 - a.boxes.m0)) : blockIdx.x;
    int first_wg, count_wg;
    const Box box = find_box(a.boxes, wg, first_wg, &count_wg);
    if (DIRECT_OK && box.kind != 0) {  // a thin box: one lane per cell
        if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;
        int x, y, z;
        float msq = 0.f;
        const unsigned wd = (SOBFU_BOX_XCD & 2) ? (unsigned) first_wg + box_xcd_order(wg, (unsigned) first_wg, (unsigned) count_wg) : wg;
        if (direct_cell(box, wd, first_wg, x, y, z)) msq = pass_b_direct_cell<WRITE_UPDATES, COMPACT, IDX32>(a, x, y, z);
        maxnorm_tail<WY>(msq, a.slots, s_max);
        return;
    }
    const TileGeom tg = geom_in_box(box, wg, first_wg, d, TY);
    if constexpr (PIPE) {
        if (tg.down) pass_b_march_pipe<WY, NTL, true>(a, tg, gate, tile, s_max);
        else pass_b_march_pipe<WY, NTL, false>(a, tg, gate, tile, s_max);
        return;
    }
    const int u0 = tg.u0, v0 = tg.v0, zb = tg.zb, ze = tg.ze;
    const int u = u0 + lx, uc = min(u, tg.DU - 1);
    const size_t plane = (size_t) d.x * d.y, sv = (size_t) d.x;

    // in-plane BYTE offsets of the lane's cells (a plane of a vector field is < 4 GiB: checked at launch); every plane base is a
    // uniform 64-bit value, so an address costs one scalar pair + one lane register
    constexpr uint32_t VB = COMPACT ? 12u : 16u, TB = COMPACT ? 4u : 8u;  // bytes per cell of a vector field / a TSDF volume
    uint32_t off[RPT], offT[RPT];  // ... in a vector field / in a TSDF volume
    bool mine[RPT];  // the cell is stored by this launch / belongs to this rank (x, y part of the test)
    bool owned[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int v = v0 + wy * RPT + r;
        const uint32_t cell = (uint32_t) ((size_t) uc + sv * (size_t) min(v, tg.DV - 1));
        off[r]      = cell * VB;
        offT[r]     = cell * TB;
        mine[r]     = u < tg.u_hi && v < tg.v_hi;
        owned[r]    = u >= a.own[0] && u < a.own[1] && v >= a.own[2] && v < a.own[3];
    }

    // halo tasks: 0..R-1 rows above, R..2R-1 rows below, then lane-halo cells (2R per tile row)
    int h_lr[TPW], h_lc[TPW];
    uint32_t h_off[TPW];
    bool h_on[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        int task = wy + k * WY;
        h_on[k]  = task < NTASK;
        int lr = 0, lc = 0;
        if (task < R) { lr = task; lc = lx + R; }
        else if (task < 2 * R) { lr = TY + task; lc = lx + R; }  // TY + R + (task - R)
        else {
            int e = (task - 2 * R) * TX + lx;  // 0 .. 2R*TY-1
            h_on[k] = h_on[k] && e < 2 * R * TY;
            int row = e / (2 * R), c = e % (2 * R);
            lr = R + row;
            lc = c < R ? c : TX + c;  // R..2R-1 -> TX+R .. TX+2R-1
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        int gu = min(max(u0 - R + lc, 0), tg.DU - 1), gv = min(max(v0 - R + lr, 0), tg.DV - 1);
        h_off[k] = (uint32_t) ((size_t) gu + (size_t) gv * sv) * VB;
    }

    float4 hq[TPW];
    // Halo cells run HL planes ahead of the plane they are staged for, like the main cells of the z pipeline (which must be 4
    // ahead): a neighbour tile's halo request then meets the owner's own request for the same lines in the L2 instead of coming
    // 3 plane-steps (~5 MB of traffic through a 4 MB L2) later -- 86 of the 111 MB pass B read beyond its minimum at 256^3 were
    // halo lines fetched twice (PMC attribution).  In between a cell waits in a per-lane LDS FIFO (only its own lane
    // ever touches an entry: no barrier involved).  Planes zb+1 .. of the first steps are requested -- and parked -- before the
    // z pipeline's seven planes are, so that their registers are free again by then.
    if (HL > 0) {
        float4 hpre[TPW][HL > 1 ? HL - 1 : 1];
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
#pragma unroll
                for (int p = 1; p < HL; ++p) hpre[k][p - 1] = ldvb<COMPACT>((const char*) a.nU + (size_t) min(zb + p, d.z - 1) * plane * VB, h_off[k]);
            }
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
#pragma unroll
                for (int p = 1; p < HL; ++p) {
                    P3& e = hfifo[(zb + p) % (HL > 0 ? HL : 1)][(wy + k * WY) * TX + lx];
                    e.x = hpre[k][p - 1].x; e.y = hpre[k][p - 1].y; e.z = hpre[k][p - 1].z;
                }
            }
    }
    // z register pipeline q[r][0..6] = planes clamp(z-3 .. z+3)  (clamp-to-edge, solver.cu:396-424)
    float4 q[RPT][7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const char* nUz = (const char*) a.nU + (size_t) min(max(zb - 3 + k, 0), d.z - 1) * plane * VB;
#pragma unroll
        for (int r = 0; r < RPT; ++r) q[r][k] = ldvb<COMPACT>(nUz, off[r]);
    }
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        if (h_on[k]) hq[k] = ldvb<COMPACT>((const char*) a.nU + (size_t) zb * plane * VB, h_off[k]);
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;
    int hslot = HL > 0 ? zb % (HL > 0 ? HL : 1) : 0;  // FIFO slot of plane z

    float msq = 0.f;
    for (int z = zb; z < ze; ++z) {
        const int buf = (z - zb) & 1;
#pragma unroll
        for (int r = 0; r < RPT; ++r) tile[buf][wy * RPT + r + R][lx + R] = q[r][3];
        if (HL > 0) {
        const int hprev = hslot == 0 ? HL - 1 : hslot - 1;  // slot of plane z-1 == slot of plane z-1+HL
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
                const int hi = (wy + k * WY) * TX + lx;
                if (z == zb) {  // plane zb's halo came straight from the prologue's request
                    tile[buf][h_lr[k]][h_lc[k]] = hq[k];
                } else {
                    const P3 e = hfifo[hslot][hi];
                    tile[buf][h_lr[k]][h_lc[k]] = make_float4(e.x, e.y, e.z, 0.f);
                    if (z - 1 + HL < ze) {  // the cell requested during the previous step (plane z-1+HL) takes the slot plane z-1 left
                        P3& w = hfifo[hprev][hi];
                        w.x = hq[k].x; w.y = hq[k].y; w.z = hq[k].z;
                    }
                }
            }
        hslot = hslot + 1 == HL ? 0 : hslot + 1;
        } else {
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) tile[buf][h_lr[k]][h_lc[k]] = hq[k];
        }

        const size_t zcur = (size_t) z * plane;
        float4 pv[RPT], nq[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if constexpr (NTBUF && NTL >= 2) pv[r] = buf_ld3(buf_rsrc(a.psi, (uint32_t) (plane * (size_t) d.z) * VB), off[r], (uint32_t) zcur * VB, true);
            else pv[r] = ldvb<COMPACT>((const char*) a.psi + zcur * VB, off[r], NTL >= 2);
        }
        if (z + 1 < ze) {
            const char* nU4 = (const char*) a.nU + (size_t) min(z + 4, d.z - 1) * plane * VB;
#pragma unroll
            for (int r = 0; r < RPT; ++r) nq[r] = ldvb<COMPACT>(nU4, off[r]);
            if (HL == 0) {
                const char* nU1 = (const char*) a.nU + (size_t) (z + 1) * plane * VB;
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (h_on[k]) hq[k] = ldvb<COMPACT>(nU1, h_off[k]);
            }
        }
        if (HL > 0 && z + HL < ze) {
            const char* nUh = (const char*) a.nU + (size_t) (z + HL) * plane * VB;
#pragma unroll
            for (int k = 0; k < TPW; ++k)
                if (h_on[k]) hq[k] = ldvb<COMPACT>(nUh, h_off[k]);
        }
        __syncthreads();
        // row-axis taps outside this lane's strip
        float4 yt[R], yb[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            yt[j] = tile[buf][wy * RPT + j][lx + R];                 // strip rows -3, -2, -1
            yb[j] = tile[buf][wy * RPT + RPT + R + j][lx + R];       // strip rows RPT, RPT+1, RPT+2
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            // the lane-axis sum (l*) and the row-axis sum (r*) are the x and y convolutions
            // packed fp32 math: the {x, y} and {z, w} halves of a cell are adjacent register pairs (ds_read_b128), so each tap is
            // 2 v_pk_mul_f32 + 2 v_pk_add_f32 instead of 3 + 3 scalar ops (the w lane rides along; products are not contracted)
            v2f l01 = {0.f, 0.f}, l23 = {0.f, 0.f}, r01 = {0.f, 0.f}, r23 = {0.f, 0.f}, z01 = {0.f, 0.f}, z23 = {0.f, 0.f};
#pragma unroll
            for (int j = -R; j <= R; ++j) {
                const v2f s2 = {a.S.s[R - j], a.S.s[R - j]};
                const float4 vl = (j == 0) ? q[r][3] : tile[buf][wy * RPT + r + R][lx + R + j];
                l01 += v2f{vl.x, vl.y} * s2;
                l23 += v2f{vl.z, vl.w} * s2;
                const int rr = r + j;
                const float4 vr = rr < 0 ? yt[rr + R < 0 ? 0 : (rr + R > R - 1 ? R - 1 : rr + R)]
                                         : (rr >= RPT ? yb[rr - RPT > R - 1 ? R - 1 : (rr - RPT < 0 ? 0 : rr - RPT)]
                                                      : q[rr < 0 ? 0 : (rr >= RPT ? RPT - 1 : rr)][3]);
                r01 += v2f{vr.x, vr.y} * s2;
                r23 += v2f{vr.z, vr.w} * s2;
                const float4 vz = q[r][3 + j];
                z01 += v2f{vz.x, vz.y} * s2;
                z23 += v2f{vz.z, vz.w} * s2;
            }
            const v2f t01 = (l01 + r01) + z01;
            float tx = t01.x, ty = t01.y, tz = (l23.x + r23.x) + z23.x;
            // = ((Sx*src) + (Sy*src)) + (Sz*src)  (rows assign, columns +=, depth +=)
            // update_psi_kernel (solver.cu:64-67)
            float4 uu = f4(tx * a.alpha, ty * a.alpha, tz * a.alpha);
            float4 p  = pv[r];
            p.x -= uu.x;
            p.y -= uu.y;
            p.z -= uu.z;
            pin3(p);
            if (mine[r]) {
                if (owned[r] && z >= a.own[4] && z < a.own[5]) msq = fmaxf(msq, norm_sq4(uu));
                // inside the box no clamp was active: off[r] is the cell itself
                if constexpr (NTBUF) buf_st3(buf_rsrc(a.psi_out, (uint32_t) (plane * (size_t) d.z) * VB), off[r], (uint32_t) zcur * VB, p, true);
                else stvb<COMPACT>((char*) a.psi_out + zcur * VB, off[r], p, NTL >= 1);
                if (WRITE_UPDATES) *(float4*) ((char*) a.updates + zcur * 16 + (size_t) (offT[r] / TB * 16u)) = uu;
                // apply_kernel (vector_fields.cu:95-98)
                if (COMPACT) {
                    const float f = IDX32 ? interp_tsdf_only32((const float*) a.phi_n, a.pd, p.x, p.y, p.z)
                                          : interp_tsdf_only((const float*) a.phi_n, a.pd, p.x, p.y, p.z);
                    float* fo = (float*) ((char*) a.pnp + zcur * TB + (size_t) offT[r]);
                    if (NTL >= 1) __builtin_nontemporal_store(f, fo);
                    else *fo = f;
                }
                else *(float2*) ((char*) a.pnp + zcur * TB + (size_t) offT[r]) = interp_tsdf((const float2*) a.phi_n, a.pd, p.x, p.y, p.z);
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                q[r][k] = q[r][k + 1];
                // keep the shift as plain register moves (hipcc otherwise SLP-vectorises the 7-deep shift of the compact
                // variant into a <7 x float> shuffle that it lowers through 64 B of scratch per lane)
                asm volatile("" : "+v"(q[r][k].x), "+v"(q[r][k].y), "+v"(q[r][k].z));
            }
            q[r][6] = nq[r];
        }
    }
    maxnorm_tail<WY>(msq, a.slots, s_max);  // max ||u||^2 over the voxels this workgroup owns
}
