// Part of solver_kernels.hip (included there, inside its anonymous namespace; not a translation unit of its own): pass A -- fused potential gradient: the z-march, the single-GPU kernel, and the multi-GPU tile kernel whose launch contains the halo exchange (push boxes, system-scope stores, arrival flags)
// clang-format off: the include order in solver_kernels.hip matters (common -> pass A -> pass B -> aux)

// --- pass A ----------------------------------------------------------------------------------------------------
struct PassACore {
    const void* pnp;  // phi_n o psi
    const void* pg;   // phi_global
    const void* psi;
    void* nU;
    Dims d;  // extents of the (local) arrays
    float w_reg;
    const uint32_t* prev_slots;
    float max_update_norm;
};
struct PassAArgs {
    PassACore c;
    BoxList boxes;  // the cells this launch produces
};

// Pass B's marching loops store under a per-lane "this cell is mine" test, and the compiler sinks everything that consumes the step's
// loads into that branch with the stores.  On the path around the branch it must then assume the loads still in flight, so at
// the join -- the pipeline shift, the next step's halo staging -- it waits for vmcnt(0), which on the path that DID store also
// waits for the stores' acknowledgement: once per plane per wave, on the critical chain.  Pinning the value about to be stored
// in front of the branch makes the wait for its loads unconditional (same place: behind the arithmetic), the join then knows
// that every load has landed, and the stores drain behind the next plane's work.  (Pass A gains nothing from the same pin: measured,
// profiles/LABBOOK.md round 4.)
SOBFU_DEV void pin3(const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z)); }

// --- buffer addressing (cache-resident launches) -------------------------------------------------------------------------------
// A 128-bit buffer resource in SGPRs (base, bytes) + a 32-bit lane byte offset + a scalar byte offset (the plane): an address costs
// no vector instruction and no 64-bit lane register pair.  Arrays below 4 GiB only (checked at launch).
typedef unsigned v3u __attribute__((ext_vector_type(3)));
SOBFU_DEV __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int) bytes, 0x00020000);
}
// nt: the streaming (nontemporal) hint, bit 1 of the cache-policy operand on gfx94x / gfx950
SOBFU_DEV float4 buf_ld3(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, bool nt = false) {
    const v3u t = nt ? __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 2) : __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), 0.f);
}
// the same load at SYSTEM scope (sc0 sc1: bits 0 and 4 of the cache-policy operand) when `sys` (wave-uniform) says so: cells another
// GPU stored -- the halo rims of nabla_U on the direct transport
SOBFU_DEV float4 buf_ld3_scope(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, bool sys) {
    const v3u t = sys ? __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 17) : __builtin_amdgcn_raw_buffer_load_b96(r, (int) voff, (int) soff, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), 0.f);
}
SOBFU_DEV float buf_ld1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int) voff, (int) soff, 0));
}
SOBFU_DEV void buf_st3(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, const float4& v, bool nt = false) {
    const v3u t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
    if (nt) __builtin_amdgcn_raw_buffer_store_b96(t, r, (int) voff, (int) soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b96(t, r, (int) voff, (int) soff, 0);
}
SOBFU_DEV void buf_st1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float v, bool nt = false) {
    if (nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int) voff, (int) soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int) voff, (int) soff, 0);
}

// One cell of pass A from its centre c = psi, fc = (phi_n o psi).tsdf, bg = phi_global.tsdf and the six RAW neighbours of psi
// (p**) and of F (f**) along x (l), y (r), z -- loaded with clamped indices; the boundary rules are applied here.  Shared by
// the marching and the direct path: the same operations in the same order.
SOBFU_DEV float4 potential_gradient_cell(const float4& c, float fc, float bg, float4 plp, float4 plm, float4 prp, float4 prm, float4 pzp,
                                         float4 pzm, float flp, float flm, float frp, float frm, float fzp, float fzm, bool ulo, bool uhi,
                                         bool vlo, bool vhi, bool zlo, bool zhi, float w_reg) {
    // TsdfDifferentiator boundary rule (vector_fields.cu:165-191): mirror the missing neighbour
    const float gl1 = uhi ? flm : flp, gl2 = ulo ? flp : flm;
    const float gr1 = vhi ? frm : frp, gr2 = vlo ? frp : frm;
    const float gz1 = zhi ? fzm : fzp, gz2 = zlo ? fzp : fzm;
    const float4 g = f4((gl1 - gl2) / 2.f, (gr1 - gr2) / 2.f, (gz1 - gz2) / 2.f);
    // SecondOrderDifferentiator boundary rule (vector_fields.cu:299-331): both neighbours <- centre
    if (ulo || uhi) { plp = c; plm = c; }
    if (vlo || vhi) { prp = c; prm = c; }
    if (zlo || zhi) { pzp = c; pzm = c; }
    // the reference adds x+, x-, y+, y-, z+, z- in that order
    float4 vv = mul4(c, -6.f);
    vv = add4(vv, plp);
    vv = add4(vv, plm);
    vv = add4(vv, prp);
    vv = add4(vv, prm);
    vv = add4(vv, pzp);
    vv = add4(vv, pzm);
    const float4 L = mul4(vv, -1.f);
    // calculate_potential_gradient_kernel (solver.cu:28-31)
    const float diff = fc - bg;
    return add4(mul4(g, diff), mul4(L, w_reg));
}

// DIRECT evaluation of one cell of pass A (thin boxes): 7 psi + 7 F + 1 G loads, all but a few of them cache hits
template <bool COMPACT>
SOBFU_DEV float4 pass_a_direct_cell(const PassACore& a, int x, int y, int z) {
    const Dims d = a.d;
    const int xm = max(x - 1, 0), xp = min(x + 1, d.x - 1), ym = max(y - 1, 0), yp = min(y + 1, d.y - 1), zm = max(z - 1, 0), zp = min(z + 1, d.z - 1);
    const size_t i = vidx(d, x, y, z), ixm = vidx(d, xm, y, z), ixp = vidx(d, xp, y, z), iym = vidx(d, x, ym, z), iyp = vidx(d, x, yp, z),
                 izm = vidx(d, x, y, zm), izp = vidx(d, x, y, zp);
    const float4 c = ldv<COMPACT>(a.psi, i);
    const float4 plp = ldv<COMPACT>(a.psi, ixp), plm = ldv<COMPACT>(a.psi, ixm), prp = ldv<COMPACT>(a.psi, iyp), prm = ldv<COMPACT>(a.psi, iym),
                 pzp = ldv<COMPACT>(a.psi, izp), pzm = ldv<COMPACT>(a.psi, izm);
    const float fc = ldt<COMPACT>(a.pnp, i), flp = ldt<COMPACT>(a.pnp, ixp), flm = ldt<COMPACT>(a.pnp, ixm), frp = ldt<COMPACT>(a.pnp, iyp),
                frm = ldt<COMPACT>(a.pnp, iym), fzp = ldt<COMPACT>(a.pnp, izp), fzm = ldt<COMPACT>(a.pnp, izm);
    const float bg = ldt<COMPACT>(a.pg, i);
    return potential_gradient_cell(c, fc, bg, plp, plm, prp, prm, pzp, pzm, flp, flm, frp, frm, fzp, fzm, x == 0, x == d.x - 1, y == 0,
                                   y == d.y - 1, z == 0, z == d.z - 1, a.w_reg);
}

// the MARCHING path of pass A for the tile tg (a z-chunk of a 64 x TY tile)
// where the cells of a PUSH box go (pass A of a multi-GPU tile: see tile_potential_gradient_kernel)
struct PushDst {
    float* base;             // null: the box is stored locally
    int ox, oy, oz, px, py;  // cell (x, y, z) -> base + 3 * ((x + ox) + px * ((y + oy) + py * (z + oz)))
    int y0, y1, lz0, lz1;    // marching push boxes: rows [y0, y1) travel; planes [lz0, lz1) are stored locally as well
};
SOBFU_DEV void st3_system(float* p, const float4& v);

// NTL: streaming (nontemporal) hints, kNT or 0 -- for grids whose state exceeds the 256 MiB Infinity Cache; 0 for cache-resident
// ones (multi-GPU tiles, small grids), where the hints keep the data the NEXT launch reads out of the cache (2 x 2 x 2 tile of
// 256^3: 54.4 -> 48.8 us per iteration without them)
template <int RPT, int WY, bool COMPACT, int NTL, bool PUSHABLE = false>
SOBFU_DEV void pass_a_march(const PassACore& a, const TileGeom& tg, const GateRegs& gate, const PushDst* pd = nullptr) {
    constexpr int TY = RPT * WY, LW = TX + 2, LH = TY + 2;
    constexpr int NXH = (2 * TY + TX - 1) / TX;  // row-tasks for the two lane-halo columns
    constexpr int NTASK = 2 + NXH, TPW = (NTASK + WY - 1) / WY;
    __shared__ float4 t_psi[2][LH][LW + 2];  // {psi.xyz, F = (phi_n o psi).tsdf} -- psi.w is never read

    const Dims d = a.d;
    const int lx = threadIdx.x, wy = threadIdx.y;
    const int u0 = tg.u0, v0 = tg.v0, zb = tg.zb, ze = tg.ze;
    const int u = u0 + lx, uc = min(u, tg.DU - 1);
    const size_t plane = (size_t) d.x * d.y, sv = (size_t) d.x;

    size_t off[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) off[r] = (size_t) uc + sv * (size_t) min(v0 + wy * RPT + r, tg.DV - 1);
    // halo tasks: task 0 = row above the tile, task 1 = row below, tasks 2.. = lane-halo cells (col -1 / col TX)
    int h_lr[TPW], h_lc[TPW];  // LDS cell
    size_t h_off[TPW];
    bool h_on[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        int task = wy + k * WY;
        h_on[k]  = task < NTASK;
        int lr = 0, lc = 0;
        if (task == 0) { lr = 0; lc = lx + 1; }
        else if (task == 1) { lr = LH - 1; lc = lx + 1; }
        else {
            int e = (task - 2) * TX + lx;  // 0 .. 2*TY-1
            h_on[k] = h_on[k] && e < 2 * TY;
            lr = 1 + (e >> 1);
            lc = (e & 1) ? LW - 1 : 0;
        }
        h_lr[k] = lr;
        h_lc[k] = lc;
        int gu = min(max(u0 - 1 + lc, 0), tg.DU - 1), gv = min(max(v0 - 1 + lr, 0), tg.DV - 1);
        h_off[k] = (size_t) gu + (size_t) gv * sv;
    }

    // z register pipeline: m = z-1, c = z, n = z+1 (clamped loads; boundary rules applied at use)
    float4 pm[RPT], pc[RPT], pn[RPT];
    float fm[RPT], fc[RPT], fn[RPT];
    float4 hp[TPW];
    float hf[TPW];
    float bg[RPT], bgn[RPT];  // phi_global of plane z, requested one step ahead like everything else (no same-step round trip)
    auto ld_bg = [&](size_t i) { return (NTL >= 3 && COMPACT) ? __builtin_nontemporal_load((const float*) a.pg + i) : ldt<COMPACT>(a.pg, i); };
    {
        const size_t zm = (size_t) max(zb - 1, 0) * plane, zc0 = (size_t) zb * plane;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pm[r] = ldv<COMPACT>(a.psi, zm + off[r]);
            fm[r] = ldt<COMPACT>(a.pnp, zm + off[r]);
            pc[r] = ldv<COMPACT>(a.psi, zc0 + off[r]);
            fc[r] = ldt<COMPACT>(a.pnp, zc0 + off[r]);
            bg[r] = ld_bg(zc0 + off[r]);
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) {
                hp[k] = ldv<COMPACT>(a.psi, zc0 + h_off[k]);
                hf[k] = ldt<COMPACT>(a.pnp, zc0 + h_off[k]);
            }
    }
    if (gate_decide(gate, a.prev_slots, a.max_update_norm)) return;

    const bool ulo = (u == 0), uhi = (u == tg.DU - 1);
    for (int z = zb; z < ze; ++z) {
        const int buf = (z - zb) & 1;
        // stage plane z
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            t_psi[buf][wy * RPT + r + 1][lx + 1] = make_float4(pc[r].x, pc[r].y, pc[r].z, fc[r]);
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k)
            if (h_on[k]) t_psi[buf][h_lr[k]][h_lc[k]] = make_float4(hp[k].x, hp[k].y, hp[k].z, hf[k]);
        // prefetch plane z+1 (main) and the halo of plane z+1
        const size_t zn = (size_t) min(z + 1, d.z - 1) * plane, zcur = (size_t) z * plane;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pn[r] = ldv<COMPACT>(a.psi, zn + off[r]);
            fn[r] = ldt<COMPACT>(a.pnp, zn + off[r]);
            if (z + 1 < ze) bgn[r] = ld_bg(zn + off[r]);
        }
        if (z + 1 < ze) {
#pragma unroll
            for (int k = 0; k < TPW; ++k)
                if (h_on[k]) {
                    hp[k] = ldv<COMPACT>(a.psi, zn + h_off[k]);
                    hf[k] = ldt<COMPACT>(a.pnp, zn + h_off[k]);
                }
        }
        __syncthreads();

        const bool zlo = (z == 0), zhi = (z == d.z - 1);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int v  = v0 + wy * RPT + r;
            const int lr = wy * RPT + r + 1;
            const bool vlo = (v == 0), vhi = (v == tg.DV - 1);
            // raw neighbours along x (l*) and y (r*)
            const float4 plp = t_psi[buf][lr][lx + 2], plm = t_psi[buf][lr][lx];
            float4 prp, prm;
            float frp, frm;
            if (r + 1 < RPT) { prp = pc[r + 1 < RPT ? r + 1 : r]; frp = fc[r + 1 < RPT ? r + 1 : r]; }
            else { prp = t_psi[buf][lr + 1][lx + 1]; frp = prp.w; }
            if (r > 0) { prm = pc[r > 0 ? r - 1 : r]; frm = fc[r > 0 ? r - 1 : r]; }
            else { prm = t_psi[buf][lr - 1][lx + 1]; frm = prm.w; }
            const float4 o = potential_gradient_cell(pc[r], fc[r], bg[r], plp, plm, prp, prm, pn[r], pm[r], plp.w, plm.w, frp, frm, fn[r], fm[r], ulo,
                                                     uhi, vlo, vhi, zlo, zhi, a.w_reg);
            if (u < tg.u_hi && v < tg.v_hi) {
                const size_t i = zcur + off[r];  // inside the box no clamp was active: off[r] is the cell itself
                if (PUSHABLE && pd->base != nullptr) {  // a marching PUSH box: the rows of the message go to their destination ...
                    if (v >= pd->y0 && v < pd->y1) {
                        const size_t j = (size_t) (u + pd->ox) + (size_t) pd->px * ((size_t) (v + pd->oy) + (size_t) pd->py * (size_t) (z + pd->oz));
                        st3_system(pd->base + 3 * j, o);
                    }
                    if (z >= pd->lz0 && z < pd->lz1) stv<COMPACT>(a.nU, i, o);  // ... and where the box stands in for the owned block, home too
                } else stv<COMPACT>(a.nU, i, o);
            }
        }
        // shift the z pipeline
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            pm[r] = pc[r];
            pc[r] = pn[r];
            fm[r] = fc[r];
            fc[r] = fn[r];
            bg[r] = bgn[r];
        }
    }
}


template <int RPT, int WY, bool COMPACT, int NTL>
__global__ void __launch_bounds__(TX* WY) fused_potential_gradient_kernel(PassAArgs a) {
    const GateRegs gate = gate_load(a.c.prev_slots, 1);
    const unsigned t    = xcd_swizzle(blockIdx.x, (unsigned) a.boxes.first[a.boxes.n]);
    int first;
    const Box b = find_box(a.boxes, t, first);
    pass_a_march<RPT, WY, COMPACT, NTL>(a.c, geom_in_box(b, t, first, a.c.d, RPT * WY), gate);
}

// ---- pass A of a multi-GPU TILE: the halo exchange is part of the launch ---------------------------------------------------
// Boxes of a tile launch, in workgroup order:
//   PUSH boxes: the cells of one halo message (a 4-cell face or a 4 x 4 edge strip of the owned block) are evaluated a second
//       time -- lane per cell where the box is thin in x, by a short march where its rows are wide -- and stored STRAIGHT INTO
//       THE DESTINATION -- the halo cells of the neighbour's nabla_U
//       array, peer-mapped over xGMI (direct transport), or this rank's packed send buffer (RCCL / callback transports): no
//       pack kernel, no unpack kernel and, with the direct transport, no communication launch at all.  They are numbered
//       first, so they leave while the owned block is still being computed.
//   the owned block (marching), stored locally.
// Synchronisation of the direct transport, in the kernel's tail (TileSync): every workgroup that pushed waits for its stores'
// acknowledgements and takes a ticket; the LAST of them writes this rank's arrival flag (= the iteration's sequence number) at
// every rank of the sync set, then waits until the flags of all those ranks have reached the sequence number (with a
// deadline): a launch retires when all its workgroups have, so when this one does every halo cell of the iteration has landed
// and pass B -- a separate launch, whose start invalidates the caches -- reads it.  nabla_U is
// double-buffered by iteration parity, which orders a neighbour's stores of iteration k+1 behind this rank's reads of
// iteration k without a second handshake (see tiled_capi.hip).
constexpr int kMaxTileBoxes = 20;  // 18 messages + the owned block + one spare
struct TileBox {
    Box b;
    PushDst push;
};
struct TileBoxList {
    int n, n_push_wgs;  // workgroups [0, n_push_wgs) belong to push boxes
    TileBox b[kMaxTileBoxes];
    int first[kMaxTileBoxes + 1];
};
// ---- stores that leave the GPU ----------------------------------------------------------------------------------------------
// What travels to a peer (message cells, row maxima, flags) is stored WRITE-THROUGH at system scope (sc0 sc1): it never sits
// dirty in this GPU's write-back L2, so "everything I sent has arrived" is `s_waitcnt vmcnt(0)` -- the stores' acknowledgements --
// and not the L2 write-back a system-scope release fence would do (pass A is filling that L2 with nabla_U at the time: one such
// fence per push workgroup cost 4x the whole iteration).  The flag goes out after the wait, so it cannot overtake the data.
SOBFU_DEV void st3_system(float* p, const float4& v) {
    const v3f o = {v.x, v.y, v.z};
    asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" ::"v"(p), "v"(o) : "memory");
}
SOBFU_DEV void st1_system(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
SOBFU_DEV uint32_t ld1_system(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
SOBFU_DEV void stores_acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wave 0 of one workgroup: the maximum of this rank's slot row -> entry `my_rank` of that row at every rank of the sync set (and
// here); the next signal covers these stores
SOBFU_DEV void tile_row_push(const TileSync* sy, const uint32_t* row, uint32_t row_index) {
    const int l = threadIdx.x;
    uint32_t m = max(max(row[l], row[l + 64]), max(row[l + 128], row[l + 192]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
    const size_t e = (size_t) row_index * 256u + sy->my_rank;
    if (l == 0) sy->my_grows[e] = m;
    for (int q = l; q < (int) sy->n_sync; q += 64) st1_system(sy->peer_grows[q] + e, m);
}
// one lane: raise this rank's arrival flag at every rank of the sync set (after stores_acknowledged() on everything it covers)
SOBFU_DEV void tile_signal(const TileSync* sy, uint32_t seq) {
    for (uint32_t q = 0; q < sy->n_sync; ++q) st1_system(sy->peer_flags[q] + sy->my_rank, seq);
}
// one lane: wait until every rank of the sync set has raised its flag to `seq` -- with a deadline: a missing peer is recorded
// (err = 1 + its rank) and every later wait returns at once, so a wedged neighbour never hangs this GPU.  The cells the flags
// announce are read by the NEXT launch (whose start invalidates the caches), never by this one.
SOBFU_DEV void tile_wait(TileSync* sy, uint32_t seq) {
    if (__hip_atomic_load(&sy->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const uint64_t t0 = wall_clock64();
    for (uint32_t q = 0; q < sy->n_sync; ++q) {
        const uint32_t* f = sy->my_flags + sy->sync_rank[q];
        while ((int32_t) (ld1_system(f) - seq) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > sy->timeout_ticks) {
                __hip_atomic_store(&sy->err, 1u + (uint32_t) sy->sync_rank[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
    // diagnostics (one lane per launch gets here): how long the launch sat waiting for its peers -- what of the exchange was NOT hidden
    sy->wait_ticks += wall_clock64() - t0;
    sy->wait_count += 1u;
}
// diagnostics: flag round trips with ONE peer (sync-set member q), `reps` of them inside one launch: the side that serves stores seq,
// the other answers seq + 1, ...; one lane each.  Both sides observe the same deadline as every other wait.
__global__ void __launch_bounds__(64) tile_pingpong_kernel(TileSync* sy, int q, int first, uint32_t seq0, int reps) {
    if (threadIdx.x != 0) return;
    uint32_t* theirs = sy->peer_flags[q] + sy->my_rank;
    const uint32_t* mine = sy->my_flags + sy->sync_rank[q];
    const uint64_t t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        const uint32_t s_ping = seq0 + 2u * (uint32_t) r, s_pong = s_ping + 1u;
        if (first) st1_system(theirs, s_ping);
        while ((int32_t) (ld1_system(mine) - (first ? s_pong : s_ping)) < 0) {
            if (wall_clock64() - t0 > sy->timeout_ticks) {
                __hip_atomic_store(&sy->err, 1u + (uint32_t) sy->sync_rank[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        if (!first) st1_system(theirs, s_pong);
    }
}
__global__ void __launch_bounds__(64) tile_flush_kernel(TileSync* sy, uint32_t seq, int wait, const uint32_t* row, uint32_t row_index) {
    if (row != nullptr) tile_row_push(sy, row, row_index);
    stores_acknowledged();
    if (threadIdx.x != 0) return;
    tile_signal(sy, seq);
    if (wait) tile_wait(sy, seq);
}

// the signalling part of a tile launch's arguments
struct TileSignal {
    TileSync* sync;       // null: no signalling (single-box launches, RCCL / callback transports)
    uint32_t seq;         // sequence number of this iteration
    int wait;             // the last workgroup waits for the peers' flags
    const uint32_t* row;  // this rank's max-norm slot row of the PREVIOUS iteration (null: none) ...
    uint32_t row_index;   // ... which is row `row_index` of the global rows
};
// The box list of a launch lives in DEVICE memory (a list is fixed for the life of a handle: uploaded once -- launch_tile_pass_a keeps
// every distinct list it has seen -- and read through the scalar cache), not in the kernel-argument segment: a 1.5 KB argument block
// costs a launch 0.6 us (tools/calib/launch_cost.hip: 2.9 -> 3.5 us back to back), and handed on by reference it ended up copied to
// 1.8 KB of scratch per lane (pass A 17 -> 129 us: found with SOBFU_TILED_DEBUG_SKIP=1)
struct TilePassAArgsP {
    PassACore c;
    const TileBoxList* boxes;
    TileSignal s;
};
template <int RPT, int WY, bool COMPACT, int NTL>
SOBFU_DEV void tile_potential_gradient_body(const PassACore& core, const TileBoxList& L, const TileSignal& sg) {
    const unsigned nb = (unsigned) L.first[L.n];
    // push workgroups keep their launch order (they go out first); the others are XCD-swizzled among themselves
    unsigned t = blockIdx.x;
    const bool push_wg = (int) t < L.n_push_wgs;
    if (!push_wg) t = (unsigned) L.n_push_wgs + xcd_swizzle(t - (unsigned) L.n_push_wgs, nb - (unsigned) L.n_push_wgs);
    Box b     = L.b[0].b;
    PushDst pd = L.b[0].push;
    int first = 0, next = L.first[1];
#pragma unroll
    for (int k = 1; k < kMaxTileBoxes; ++k)
        if (k < L.n && (int) t >= L.first[k]) {
            b     = L.b[k].b;
            pd    = L.b[k].push;
            first = L.first[k];
            next  = L.first[k + 1];
        }
    // inside a push box every XCD takes a contiguous run of the box's cells (see box_xcd_order)
    if (push_wg && (SOBFU_BOX_XCD & 1)) t = (unsigned) first + box_xcd_order(t, (unsigned) first, (unsigned) (next - first));
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    // the max-norm of the previous iteration, made global without a collective (workgroup 0 is a push workgroup: the signal
    // below covers these stores)
    if (sg.sync != nullptr && sg.row != nullptr && blockIdx.x == 0 && threadIdx.y == 0) tile_row_push(sg.sync, sg.row, sg.row_index);
    if (b.kind != 0) {
        int x, y, z;
        if (direct_cell(b, t, first, x, y, z)) {
            const float4 o = pass_a_direct_cell<COMPACT>(core, x, y, z);
            if (pd.base != nullptr) {
                const size_t i = (size_t) (x + pd.ox) + (size_t) pd.px * ((size_t) (y + pd.oy) + (size_t) pd.py * (size_t) (z + pd.oz));
                st3_system(pd.base + 3 * i, o);  // messages are always 12-byte cells
            } else {
                stv<COMPACT>(core.nU, vidx(core.d, x, y, z), o);
            }
        }
    } else {
        GateRegs gate;
#pragma unroll
        for (int k = 0; k < 8; ++k) gate.v[k] = 0xffffffffu;  // pass A of a tile writes scratch only: never gated
        pass_a_march<RPT, WY, COMPACT, NTL, true>(core, geom_in_box(b, t, first, core.d, RPT * WY), gate, &pd);
    }
    if (sg.sync == nullptr || !push_wg) return;
    // the push workgroups count themselves out; the LAST one raises this rank's flag at its peers and then waits for theirs: a
    // launch retires when all its workgroups have, so pass B cannot start before every neighbour's cells have landed -- while the
    // owned block's workgroups never touch the synchronisation at all
    TileSync* sy = sg.sync;
    stores_acknowledged();  // every lane: what it stored at the peers has arrived ...
    __syncthreads();        // ... before lane 0 takes the workgroup's ticket
    if (tid != 0) return;
    const uint32_t k = __hip_atomic_fetch_add(&sy->ticket_push, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == (uint32_t) L.n_push_wgs - 1u) {
        __hip_atomic_store(&sy->ticket_push, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tile_signal(sy, sg.seq);
        if (sg.wait) tile_wait(sy, sg.seq);
    }
}

template <int RPT, int WY, bool COMPACT, int NTL>
__global__ void __launch_bounds__(TX* WY) tile_potential_gradient_kernel(TilePassAArgsP a) {
    tile_potential_gradient_body<RPT, WY, COMPACT, NTL>(a.c, *a.boxes, a.s);
}
