// Part of solver_kernels.hip (included there, inside its anonymous namespace; not a translation unit of its own): storage formats and their loads / stores, the workgroup -> tile map (boxes, XCD order), the convergence gate
// clang-format off: the include order in solver_kernels.hip matters (common -> pass A -> pass B -> aux)

// ============================================================================================================
// Part 2: fused two-pass iteration
// ============================================================================================================

constexpr int TX = 64;  // tile width in lanes: one wave per tile row (32-wide tiles measured slower: profiles/LABBOOK.md, round 5)

// --- storage formats ------------------------------------------------------------------------------------------------
// API format (COMPACT = false): the reference's layouts -- psi / nabla_U float4 (w == 0), TSDF volumes float2.
// Compact format (COMPACT = true), private to the solver handle while it iterates: psi / nabla_U as packed 12-byte
// xyz triples (the w lane is a constant 0 that would cost 25 % of their traffic) and tsdf-only 4-byte copies of
// phi_global, phi_n and phi_n o psi (the weight lane is not read by the iteration; it is rebuilt once after the
// loop).  Same arithmetic on the same values => bit-identical results, 76 instead of 112 bytes per voxel-iteration.
struct P3 {  // 12-byte element of the compact fields (only used for pointer arithmetic / sizeof)
    float x, y, z;
};
typedef float v3f __attribute__((ext_vector_type(3)));
typedef v3f __attribute__((aligned(4))) v3f_u;  // 12-byte access, 4-byte aligned -> global_load/store_dwordx3
template <bool C>
SOBFU_DEV float4 ldv(const void* base, size_t i) {
    if (C) {
        v3f v = *(const v3f_u*) ((const float*) base + 3 * i);
        return make_float4(v.x, v.y, v.z, 0.f);
    }
    return ((const float4*) base)[i];
}
template <bool C>
SOBFU_DEV void stv(void* base, size_t i, const float4& v) {
    if (C) {
        v3f o = {v.x, v.y, v.z};
        *(v3f_u*) ((float*) base + 3 * i) = o;
    } else {
        ((float4*) base)[i] = v;
    }
}
// streaming variants (nontemporal hint) for data a launch touches exactly once: they should not displace the lines that
// neighbouring workgroups re-read from the XCD's L2 (nabla_U halo rows, phi_n corners)
// kNT = 3: pass B's stores of psi / phi_n o psi, its load of psi, and pass A's load of phi_global carry the hint on grids beyond the
// Infinity Cache (template NTL = kNT; 0 on cache-resident grids).  Hinting the nabla_U store / pass A's inner rows as well was
// within run-to-run noise (profiles/LABBOOK.md, round 5) and is gone.
// NB (found in the ISA in round 5): hipcc keeps the hint of __builtin_nontemporal_load / _store on 4-byte accesses (`global_store_dword
// ... nt`: phi_n o psi, phi_global, F) but DROPS it on the 4-byte-aligned 12-byte vector type -- the 12-byte variants below compile to
// plain `global_load/store_dwordx3`.  Where the hint on a 12-byte access matters it goes through a buffer instruction, whose cache-policy
// operand carries it (buf_ld3 / buf_st3: the pipelined march, and the plain march's psi load / store: template NTBUF, + 3.4 %
// iterations/s at 256^3).
constexpr int kNT = 3;
// The same accesses as (uniform plane pointer) + (32-bit byte offset of the lane's cell in the plane): the address is a scalar base
// plus one 32-bit lane register (global_load ... v_off, s[base]) instead of a 64-bit lane address per stream.
template <bool C>
SOBFU_DEV float4 ldvb(const char* plane_ptr, uint32_t byte_off, bool nt = false) {
    const char* p = plane_ptr + (size_t) byte_off;
    if (C) {
        v3f v = nt ? __builtin_nontemporal_load((const v3f_u*) p) : *(const v3f_u*) p;
        return make_float4(v.x, v.y, v.z, 0.f);
    }
    typedef float v4f __attribute__((ext_vector_type(4)));
    if (nt) {
        v4f v = __builtin_nontemporal_load((const v4f*) p);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *(const float4*) p;
}
template <bool C>
SOBFU_DEV void stvb(char* plane_ptr, uint32_t byte_off, const float4& v, bool nt = false) {
    char* p = plane_ptr + (size_t) byte_off;
    if (C) {
        v3f o = {v.x, v.y, v.z};
        if (nt) __builtin_nontemporal_store(o, (v3f_u*) p);
        else *(v3f_u*) p = o;
    } else {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f o = {v.x, v.y, v.z, v.w};
        if (nt) __builtin_nontemporal_store(o, (v4f*) p);
        else *(v4f*) p = o;
    }
}
template <bool C>
SOBFU_DEV float ldt(const void* base, size_t i) {  // tsdf of voxel i
    return C ? ((const float*) base)[i] : ((const float2*) base)[i].x;
}
// interpolate_tsdf on a tsdf-only volume (utils.hpp:50-86 without the weight fetch)
SOBFU_DEV float interp_tsdf_only(const float* __restrict__ v, const Dims& d, float px, float py, float pz) {
    Tri a = tri_setup(px, d.x), b = tri_setup(py, d.y), c = tri_setup(pz, d.z);
    const size_t sy = (size_t) d.x, sz = (size_t) d.x * d.y;
    const float* pg = v + (size_t) b.g * sy + (size_t) c.g * sz;
    const size_t dy = (size_t) (b.h - b.g) * sy, dz = (size_t) (c.h - c.g) * sz;
    float hhh = pg[a.h + dy + dz], hhg = pg[a.h + dy], hgh = pg[a.h + dz], hgg = pg[a.h];
    float ghh = pg[a.g + dy + dz], ghg = pg[a.g + dy], ggh = pg[a.g + dz], ggg = pg[a.g];
    return lerp1(lerp1(lerp1(hhh, hhg, c.t), lerp1(hgh, hgg, c.t), b.t), lerp1(lerp1(ghh, ghg, c.t), lerp1(ggh, ggg, c.t), b.t), a.t);
}

// The same sampler with 32-bit BYTE offsets from the (uniform) volume base: one scalar base + a 32-bit lane offset per corner
// (global_load_dword v, v_off, s[base]) instead of eight 64-bit lane addresses -- ~20 VALU fewer per voxel.  Valid while the
// tsdf-only volume is < 4 GiB (< 2^30 voxels); same loads, same lerp chain, same bits.
SOBFU_DEV float interp_tsdf_only32(const float* __restrict__ v, const Dims& d, float px, float py, float pz) {
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

// interp_tsdf_only32 in two halves, for the software-pipelined pass B: the eight corner loads are ISSUED when a plane's psi is
// known and CONSUMED one plane later (the gather's round trip then overlaps the next plane's barrier and taps instead of ending
// every plane's dependent chain).  Same loads, same lerp chain, same bits.
struct Gather8 {
    float c[8];  // hhh, hhg, hgh, hgg, ghh, ghg, ggh, ggg
    float ta, tb, tc;
};
SOBFU_DEV Gather8 gather_issue32(const float* __restrict__ v, const Dims& d, float px, float py, float pz) {
    const Tri a = tri_setup(px, d.x), b = tri_setup(py, d.y), c = tri_setup(pz, d.z);
    const uint32_t sy = 4u * (uint32_t) d.x, sz = sy * (uint32_t) d.y;
    const uint32_t o  = 4u * (uint32_t) a.g + sy * (uint32_t) b.g + sz * (uint32_t) c.g;
    const uint32_t ox = a.h != a.g ? 4u : 0u, oy = b.h != b.g ? sy : 0u, oz = c.h != c.g ? sz : 0u;
    const char* base = (const char*) v;
    auto at = [&](uint32_t off) { return *(const float*) (base + (size_t) off); };
    Gather8 g;
    g.c[0] = at(o + ox + oy + oz); g.c[1] = at(o + ox + oy); g.c[2] = at(o + ox + oz); g.c[3] = at(o + ox);
    g.c[4] = at(o + oy + oz); g.c[5] = at(o + oy); g.c[6] = at(o + oz); g.c[7] = at(o);
    g.ta = a.t; g.tb = b.t; g.tc = c.t;
    return g;
}
SOBFU_DEV float gather_finish(const Gather8& g) {
    return lerp1(lerp1(lerp1(g.c[0], g.c[1], g.tc), lerp1(g.c[2], g.c[3], g.tc), g.tb), lerp1(lerp1(g.c[4], g.c[5], g.tc), lerp1(g.c[6], g.c[7], g.tc), g.tb), g.ta);
}

// --- workgroup -> tile map ---------------------------------------------------------------------------------------
// A launch produces up to kMaxBoxes BOXES of cells of the (local) array: the whole volume on a single GPU; on a multi-GPU
// tile, the owned cells (plus the one-cell shells pass B refreshes) or the boundary / interior regions of an overlapped
// schedule.  Workgroups are numbered box after box.
//   MARCHING box (kind 0): x-tile (64 lanes) fastest, then y-tile, then z-chunk; a workgroup marches its z-chunk with the
//       register / LDS pipeline described above.
//   DIRECT box (kind 1): one lane per cell, every tap read straight from the L1 / L2 -- for THIN regions (the one-cell x / y
//       shells of a tile, the 4-cell faces and 4 x 4 edge strips that travel to the neighbours), where a march would either
//       leave 63 of 64 lanes idle (regions thin in x), waste most of an 8-row tile (thin in y) or be all prologue (thin in
//       z).  A wave covers a (wx x 64/wx) patch of an x-y plane, wx = min(64, pow2ceil(x extent)): coalesced along x as far
//       as the box allows; no LDS, no barrier, one round trip.  Such boxes hold a few per cent of the cells, so the ~20
//       cached loads a cell costs this way do not matter; the arithmetic is op for op the marching path's.
//
// With the XCD swizzle the linear id is first remapped so that each XCD (workgroup b runs on XCD b % 8 -- observed, used
// for speed only) owns a contiguous run of tiles and serves neighbour-tile halos from its own L2.  PMC (256^3): fabric
// bytes per launch drop 1.013 -> 0.821 GB for pass A and 1.406 -> 1.286 GB for pass B.
constexpr int kMaxBoxes = 6;
struct Box {
    int x0, x1, y0, y1, z0, z1;  // cells [x0, x1) x [y0, y1) x [z0, z1)
    int zc;                      // marching: planes per march (z-chunk); direct: wx, the lanes of a wave that run along x
    int kind;                    // 0 marching, 1 direct
    int wpg;                     // direct: waves of a workgroup that take cells (the others leave at once) -- see direct_wpg()
    int rem;                     // marching: the first `rem` z-chunks march zc + 1 planes (an even split of the planes over a chosen NUMBER of chunks)
    int pair;                    // marching, pass B: z-chunks march in alternating directions (even chunks top-down, odd ones bottom-up), so that
                                 // two neighbours start at -- or arrive at -- their common boundary TOGETHER: the 6 planes either side of it,
                                 // which both read, are fetched once where the two share an XCD (and its L2) instead of a march apart
};
struct BoxList {
    int n;
    int m0, m1;   // workgroups [m0, m1) belong to marching boxes, the rest to direct boxes
    Box b[kMaxBoxes];
    int first[kMaxBoxes + 1];  // first workgroup of box i; first[n] = workgroups in the launch
};
struct TileGeom {
    int u0, v0, zb, ze;  // tile origin along x / y, planes [zb, ze) of this march
    int u_hi, v_hi;      // cells with x >= u_hi or y >= v_hi are outside the box (computed, not stored)
    int DU, DV;          // array extents along x / y
    bool down;           // the march runs from plane ze - 1 down to zb (Box::pair)
};
SOBFU_DEV unsigned xcd_swizzle(unsigned t, unsigned nb) {
    const unsigned q = nb / 8u, rem = nb % 8u, xcd = t % 8u, slot = t / 8u;
    return xcd * q + min(xcd, rem) + slot;  // bijective for any nb
}
// Position of workgroup t inside a box of n workgroups numbered [first, first + n) in DISPATCH order -- which hands consecutive
// workgroups to consecutive XCDs -- such that every XCD gets a CONTIGUOUS run of the box's own order (x fastest, then y, then z):
// the workgroups of a thin box that share cache lines (the same rows one plane up or down, the rows next door) then share an L2
// too, while the box as a whole stays spread over all eight XCDs and over time exactly as before.
SOBFU_DEV unsigned box_xcd_order(unsigned t, unsigned first, unsigned n) {
    const unsigned d = (t - first) % 8u, slot = (t - first) / 8u;  // d: which of the box's eight interleaved streams; same XCD <=> same d
    unsigned pre = 0;
#pragma unroll
    for (unsigned k = 0; k < 7u; ++k)
        if (k < d) pre += n > k ? (n - k + 7u) / 8u : 0u;  // members of stream k
    return pre + slot;
}
#ifndef SOBFU_BOX_XCD
#define SOBFU_BOX_XCD 3  // bit 0: push / direct boxes of a tile's pass A, bit 1: direct boxes of pass B take the XCD-contiguous order
#endif
// marching geometry of workgroup t inside box b whose first workgroup is `first` (all scalar)
SOBFU_DEV TileGeom geom_in_box(const Box& b, unsigned t, int first, const Dims& d, int ty) {
    t -= (unsigned) first;
    TileGeom g;
    g.u_hi = b.x1;
    g.v_hi = b.y1;
    g.DU   = d.x;
    g.DV   = d.y;
    const unsigned ntu = (unsigned) ((b.x1 - b.x0 + TX - 1) / TX), ntv = (unsigned) ((b.y1 - b.y0 + ty - 1) / ty);
    g.u0 = b.x0 + (int) (t % ntu) * TX;
    g.v0 = b.y0 + (int) ((t / ntu) % ntv) * ty;
    const int ck = (int) (t / (ntu * ntv));
    g.zb = b.z0 + ck * b.zc + min(ck, b.rem);
    g.ze = min(g.zb + b.zc + (ck < b.rem ? 1 : 0), b.z1);
    g.down = b.pair != 0 && (ck & 1) == 0;
    return g;
}
// the cell of this lane in a DIRECT box; false: the lane has none
SOBFU_DEV bool direct_cell(const Box& b, unsigned t, int first, int& x, int& y, int& z) {
    const int wx = b.zc, wyl = 64 / wx;
    const unsigned ntx = (unsigned) ((b.x1 - b.x0 + wx - 1) / wx), nty = (unsigned) ((b.y1 - b.y0 + wyl - 1) / wyl);
    const unsigned wv = (unsigned) __builtin_amdgcn_readfirstlane((int) threadIdx.y);
    if (wv >= (unsigned) b.wpg) return false;
    const unsigned w = (t - (unsigned) first) * (unsigned) b.wpg + wv;  // wave of the box
    const int lane = threadIdx.x;
    x = b.x0 + (int) (w % ntx) * wx + (lane & (wx - 1));
    y = b.y0 + (int) ((w / ntx) % nty) * wyl + lane / wx;
    z = b.z0 + (int) (w / (ntx * nty));
    return z < b.z1 && x < b.x1 && y < b.y1;
}
// box of workgroup t (constant indices only: a dynamically indexed by-value argument would be copied to scratch)
SOBFU_DEV Box find_box(const BoxList& L, unsigned t, int& first, int* count = nullptr) {
    Box b = L.b[0];
    first = 0;
    int next = L.first[1];
#pragma unroll
    for (int k = 1; k < kMaxBoxes; ++k)
        if (k < L.n && (int) t >= L.first[k]) {
            b     = L.b[k];
            first = L.first[k];
            next  = L.first[k + 1];
        }
    if (count) *count = next - first;
    return b;
}

// --- convergence gate ----------------------------------------------------------------------------------------
// Pass B folds max ||u||^2 of iteration k into 256 uint32 slots (non-negative floats order like their bit
// patterns).  A kernel of iteration k+1 receives the slots of iteration k and returns immediately when
// sqrt_rd(max) <= max_update_norm -- the reference's `break` (solver.cu:183) without a host round trip.
//
// prev_rows = 2 (native tiled loop, late gate): the gate is true when the row at prev_slots OR the row before it says
// "converged".  That loop gates iteration j on row j-2, runs iteration k+1 speculatively after the threshold fired at k, and
// a gated launch leaves its own row at the all-zero (= converged) state it was cleared to -- looking at rows j-2 and j-3
// makes the stop sticky for both parities whatever the speculative row k+1 holds.
SOBFU_DEV bool solver_converged(const uint32_t* __restrict__ prev_slots, float max_update_norm, int prev_rows = 1) {
    if (prev_slots == nullptr) return false;
    __shared__ int s_flag;
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    if (tid < 64) {
        bool conv = false;
        for (int r = 0; r < prev_rows; ++r) {
            const uint32_t* row = prev_slots - (size_t) r * 256;
            uint32_t m = max(max(row[tid], row[tid + 64]), max(row[tid + 128], row[tid + 192]));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
            conv = conv || sqrt_rd(__uint_as_float(m)) <= max_update_norm;
        }
        if (tid == 0) s_flag = conv ? 1 : 0;
    }
    __syncthreads();
    return s_flag != 0;
}

// The same gate in two halves for the marching kernels: the slot loads are issued FIRST, the z-pipeline prologue loads
// behind them, and the verdict is formed after that -- the gate's L2 round trip overlaps the prologue's instead of
// preceding it (the marching loop itself is untouched).
struct GateRegs {
    uint32_t v[8];
};
SOBFU_DEV GateRegs gate_load(const uint32_t* __restrict__ prev_slots, int prev_rows, bool sys = false /* the rows hold entries other GPUs stored */) {
    GateRegs g;
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
#pragma unroll
    for (int k = 0; k < 8; ++k) g.v[k] = 0xffffffffu;  // "not converged" filler for rows that are not looked at
    if (prev_slots != nullptr && tid < 64) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < prev_rows) {
                const uint32_t* row = prev_slots - (size_t) r * 256;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    g.v[4 * r + k] = sys ? __hip_atomic_load(row + tid + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : row[tid + 64 * k];
            }
    }
    return g;
}
SOBFU_DEV bool gate_decide(const GateRegs& g, const uint32_t* __restrict__ prev_slots, float max_update_norm) {
    if (prev_slots == nullptr) return false;
    __shared__ int s_flag;
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    if (tid < 64) {
        bool conv = false;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            uint32_t m = max(max(g.v[4 * r], g.v[4 * r + 1]), max(g.v[4 * r + 2], g.v[4 * r + 3]));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t) __shfl_xor((int) m, o, 64));
            conv = conv || sqrt_rd(__uint_as_float(m)) <= max_update_norm;  // 0xffffffff is a NaN pattern: never <=
        }
        if (tid == 0) s_flag = conv ? 1 : 0;
    }
    __syncthreads();
    return s_flag != 0;
}
