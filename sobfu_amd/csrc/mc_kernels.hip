// Marching cubes over a TSDF volume on gfx950 -- counterpart of src/kfusion/cuda/marching_cubes.cu + the host wrapper
// src/kfusion/marching_cubes.cpp.  Not on the solver's hot path: once per extracted mesh.
//
// The reference compacts active cells with a global atomic counter, so its voxel (and therefore triangle) ORDER differs
// from run to run.  Here the order is deterministic -- ascending voxel index, the canonical member of that family -- by
// compacting with a three-kernel exclusive scan (per-block sums -> one-block scan of the sums -> per-block scatter)
// instead of atomics; the same scan then turns the per-voxel vertex counts into vertex offsets (the reference uses
// thrust::exclusive_scan).  Classification is one lane per cell with x fastest: the 8 corner reads of a wave are 4 pairs
// of neighbouring 512-byte row segments, served by L1/L2.
//
// Arithmetic conventions: see oracle/sobfu_oracle.c (marching-cubes block) -- IEEE / and sqrt, no contraction, fma only in
// dot() and pose * vertex.
#include <cstdlib>

#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"

using namespace sobfu_hip;

namespace {

__device__ const uint64_t kTri[256] = {
#include "mc_table.inc"
};

SOBFU_DEV int tri_edge(uint64_t row, int k) { return (int) ((row >> (4 * k)) & 15u); }
SOBFU_DEV int num_verts(uint64_t row) {  // entries before the first 0xF nibble (numVertsTable)
    // a nibble is 0xF iff all four bits are set: AND the four bit planes, find the first set nibble
    uint64_t m = row & (row >> 1) & (row >> 2) & (row >> 3) & 0x1111111111111111ull;
    return m ? (int) (__builtin_ctzll(m) >> 2) : 16;
}

// CubeIndexEstimator::computeCubeIndex (marching_cubes.cu:38-79), isoValue = 0
SOBFU_DEV int cube_index(const float2* __restrict__ vol, const Dims& d, int x, int y, int z, float f[8]) {
    const size_t sy = (size_t) d.x, sz = (size_t) d.x * d.y, o = vidx(d, x, y, z);
    const size_t off[8] = {o, o + 1, o + 1 + sy, o + sy, o + sz, o + 1 + sz, o + 1 + sy + sz, o + sy + sz};
    bool seen = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float2 v = vol[off[c]];
        f[c]     = v.x;
        seen     = seen && v.y != 0.f;
    }
    if (!seen) return 0;
    int cube = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) cube += (f[c] < 0.f) << c;
    return cube;
}

constexpr int kBlock = 256, kItems = 8, kChunk = kBlock * kItems;  // cells per workgroup

// block-wide exclusive prefix of one int per lane; returns the prefix, *total = block sum (valid in every lane)
SOBFU_DEV int block_exclusive(int v, int* total, int* s_wave /* kBlock / 64 + 1 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    __syncthreads();  // s_wave may still be read from the previous call
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        int t = s_wave[w];
        if (w < wave) base += t;
        sum += t;
    }
    *total = sum;
    return base + incl - v;
}

// pass 1: vertex count of every cell (uint8 scratch) + per-workgroup number of active cells
__global__ void __launch_bounds__(kBlock) classify_kernel(const float2* __restrict__ vol, Dims d, uint8_t* __restrict__ nv_out,
                                                          int* __restrict__ block_cnt) {
    __shared__ int s_wave[kBlock / 64 + 1];
    const size_t N = (size_t) d.x * d.y * d.z, base = (size_t) blockIdx.x * kChunk;
    int active = 0;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const size_t i = base + (size_t) it * kBlock + threadIdx.x;
        int nv = 0;
        if (i < N) {
            const int x = (int) (i % d.x), y = (int) ((i / d.x) % d.y), z = (int) (i / ((size_t) d.x * d.y));
            if (x + 1 < d.x && y + 1 < d.y && z + 1 < d.z) {
                float f[8];
                const int cube = cube_index(vol, d, x, y, z, f);
                nv = (cube == 0 || cube == 255) ? 0 : num_verts(kTri[cube]);
            }
            nv_out[i] = (uint8_t) nv;
        }
        active += nv > 0;
    }
    int total;
    block_exclusive(active, &total, s_wave);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

// pass 2: exclusive scan of n ints by ONE workgroup (n = number of workgroups of pass 1 / pass 3: a few thousand)
__global__ void __launch_bounds__(1024) scan_blocks_kernel(int* __restrict__ v, int n, int* __restrict__ total_out) {
    __shared__ int s_wave[17];
    __shared__ int s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b = 0; b < n; b += 1024) {
        const int i = b + threadIdx.x, x = i < n ? v[i] : 0;
        int incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int base = s_carry;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        if (i < n) v[i] = base + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

// pass 3: scatter the active cells of each workgroup in ascending index order
__global__ void __launch_bounds__(kBlock) compact_kernel(const uint8_t* __restrict__ nv_in, size_t N, const int* __restrict__ block_off,
                                                         int* __restrict__ voxel_idx, int* __restrict__ voxel_nv, int max_size) {
    __shared__ int s_wave[kBlock / 64 + 1];
    const size_t base = (size_t) blockIdx.x * kChunk;
    int run = block_off[blockIdx.x];
#pragma unroll 1
    for (int it = 0; it < kItems; ++it) {
        const size_t i = base + (size_t) it * kBlock + threadIdx.x;
        const int nv = i < N ? nv_in[i] : 0;
        int total;
        const int pos = run + block_exclusive(nv > 0, &total, s_wave);
        if (nv > 0 && pos < max_size) {
            voxel_idx[pos] = (int) i;
            voxel_nv[pos]  = nv;
        }
        run += total;
    }
}

// generic int exclusive scan, same three passes: sums of kChunk-element blocks, scan of the sums, local scan + offset
__global__ void __launch_bounds__(kBlock) chunk_sum_kernel(const int* __restrict__ in, int n, int* __restrict__ block_sum) {
    __shared__ int s_wave[kBlock / 64 + 1];
    int s = 0;
    for (int it = 0; it < kItems; ++it) {
        const size_t i = (size_t) blockIdx.x * kChunk + (size_t) it * kBlock + threadIdx.x;
        s += i < (size_t) n ? in[i] : 0;
    }
    int total;
    block_exclusive(s, &total, s_wave);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kBlock) chunk_scan_kernel(const int* __restrict__ in, int n, const int* __restrict__ block_off,
                                                            int* __restrict__ out) {
    __shared__ int s_wave[kBlock / 64 + 1];
    int run = block_off[blockIdx.x];
#pragma unroll 1
    for (int it = 0; it < kItems; ++it) {
        const size_t i = (size_t) blockIdx.x * kChunk + (size_t) it * kBlock + threadIdx.x;
        const int x = i < (size_t) n ? in[i] : 0;
        int total;
        const int pos = run + block_exclusive(x, &total, s_wave);
        if (i < (size_t) n) out[i] = pos;
        run += total;
    }
}

struct Pose {
    float R[9], t[3];
};

SOBFU_DEV void interp(const float p0[3], const float p1[3], float f0, float f1, float out[3]) {  // vertex_interp :193-199
    const float t = (0.f - f0) / (f1 - f0 + 1e-15f);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = p0[k] + t * (p1[k] - p0[k]);
}

// TrianglesGenerator::operator() (marching_cubes.cu:201-268): one lane per active cell
__global__ void __launch_bounds__(256) triangles_kernel(const float2* __restrict__ vol, Dims d, const int* __restrict__ voxel_idx,
                                                        const int* __restrict__ vertex_off, int count, float csx, float csy, float csz,
                                                        Pose pose, float4* __restrict__ out_v, float4* __restrict__ out_n, int max_vertices) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    const int voxel = voxel_idx[idx];
    const int z = voxel / (d.x * d.y), y = (voxel - z * d.x * d.y) / d.x, x = (voxel - z * d.x * d.y) - y * d.x;
    float f[8];
    const int cube = cube_index(vol, d, x, y, z, f);
    const uint64_t row = kTri[cube];
    const int nv = num_verts(row), first = vertex_off[idx];
    const float cx[2] = {((float) x + 0.5f) * csx, ((float) (x + 1) + 0.5f) * csx};  // get_node_coo :183-191
    const float cy[2] = {((float) y + 0.5f) * csy, ((float) (y + 1) + 0.5f) * csy};
    const float cz[2] = {((float) z + 0.5f) * csz, ((float) (z + 1) + 0.5f) * csz};
    for (int i = 0; i < nv; i += 3) {
        if (first + i + 3 > max_vertices) break;
        float p[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // edge e joins corners (ea, eb): 0-3 bottom ring, 4-7 top ring, 8-11 verticals (:232-243)
            const int e = tri_edge(row, i + k);
            const int ea = e < 8 ? e : e - 8, eb = e < 8 ? ((e & 3) == 3 ? e - 3 : e + 1) : e - 4;
            // corner c: x offset = ((c & 3) == 1 || (c & 3) == 2), y offset = (c & 3) >= 2, z offset = c >> 2
            const float a[3] = {cx[((ea & 3) == 1) | ((ea & 3) == 2)], cy[(ea & 3) >> 1], cz[ea >> 2]};
            const float b[3] = {cx[((eb & 3) == 1) | ((eb & 3) == 2)], cy[(eb & 3) >> 1], cz[eb >> 2]};
            float fa = 0.f, fb = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {  // select without dynamic indexing of f[] (keeps it in registers)
                fa = c == ea ? f[c] : fa;
                fb = c == eb ? f[c] : fb;
            }
            interp(a, b, fa, fb, p[k]);
        }
        // normalized(cross(v3 - v1, v2 - v1)) (:258), one normal per triangle
        const float ax = p[2][0] - p[0][0], ay = p[2][1] - p[0][1], az = p[2][2] - p[0][2];
        const float bx = p[1][0] - p[0][0], by = p[1][1] - p[0][1], bz = p[1][2] - p[0][2];
        const float c[3] = {ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx};
        const float inv  = 1.f / __builtin_sqrtf(dot3(c, c[0], c[1], c[2]));
        const float4 n   = make_float4(c[0] * inv, -(c[1] * inv), -(c[2] * inv), 1.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // pose * vertex, store_point (:260-273)
            const float wx = dot3(pose.R + 0, p[k][0], p[k][1], p[k][2]) + pose.t[0];
            const float wy = dot3(pose.R + 3, p[k][0], p[k][1], p[k][2]) + pose.t[1];
            const float wz = dot3(pose.R + 6, p[k][0], p[k][1], p[k][2]) + pose.t[2];
            out_v[first + i + k] = make_float4(wx, -wy, -wz, 1.f);
            out_n[first + i + k] = n;
        }
    }
}

int scan_in_place_sums(int* d_sums, int nb, int* d_total, hipStream_t st) {
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, d_sums, nb, d_total);
    return (int) hipGetLastError();
}

}  // namespace

extern "C" {

// Scratch of the two scan-based steps: one vertex-count byte per voxel + one int per 1024-voxel chunk (+ the total).
// A caller-provided workspace (sobfu_hip_mc_workspace_bytes) avoids the hipMalloc / hipFree pair -- two implicit device
// synchronisations -- per call; without one (NULL / too small) the scratch is allocated for the call.
static size_t nv_bytes(size_t N) { return (N + 255) / 256 * 256; }

size_t sobfu_hip_mc_workspace_bytes(int X, int Y, int Z) {
    if (X <= 0 || Y <= 0 || Z <= 0) return 0;
    const size_t N = (size_t) X * Y * Z;
    return nv_bytes(N) + ((N + kChunk - 1) / kChunk + 1) * sizeof(int);
}

int sobfu_hip_mc_occupied_voxels(void* stream, const float* d_vol, int X, int Y, int Z, int* d_occupied, int stride, int max_size,
                                 int* h_count, void* d_workspace, size_t workspace_bytes) {
    SOBFU_CHECK_ARGS(d_vol && d_occupied && h_count && X > 0 && Y > 0 && Z > 0 && max_size > 0 && stride >= max_size);
    const size_t N = (size_t) X * Y * Z;
    if (N > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;  // int32 voxel indices in `occupied`, like the reference
    hipStream_t st = (hipStream_t) stream;
    const int nb = (int) ((N + kChunk - 1) / kChunk);
    const bool own = !d_workspace || workspace_bytes < sobfu_hip_mc_workspace_bytes(X, Y, Z);
    uint8_t* d_nv = (uint8_t*) d_workspace;
    if (own) SOBFU_HIP_TRY(hipMalloc((void**) &d_nv, sobfu_hip_mc_workspace_bytes(X, Y, Z)));
    int* d_blk = (int*) (d_nv + nv_bytes(N));  // nb block counts + 1 total
    hipLaunchKernelGGL(classify_kernel, dim3(nb), dim3(kBlock), 0, st, (const float2*) d_vol, Dims{X, Y, Z}, d_nv, d_blk);
    int rc = (int) hipGetLastError();
    if (rc == 0) rc = scan_in_place_sums(d_blk, nb, d_blk + nb, st);
    if (rc == 0) {
        hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(kBlock), 0, st, d_nv, N, d_blk, d_occupied, d_occupied + stride, max_size);
        rc = (int) hipGetLastError();
    }
    int found = 0;
    if (rc == 0) rc = (int) hipMemcpyAsync(&found, d_blk + nb, sizeof(int), hipMemcpyDeviceToHost, st);
    if (rc == 0) rc = (int) hipStreamSynchronize(st);
    if (own) (void) hipFree(d_nv);
    if (rc == 0) *h_count = found < max_size ? found : max_size;
    return rc;
}

int sobfu_hip_mc_offsets(void* stream, int* d_occupied, int stride, int count, int* h_total_vertices, void* d_workspace,
                         size_t workspace_bytes) {
    SOBFU_CHECK_ARGS(d_occupied && h_total_vertices && count >= 0 && stride >= count);
    if (count == 0) { *h_total_vertices = 0; return 0; }
    hipStream_t st = (hipStream_t) stream;
    const int nb = (count + kChunk - 1) / kChunk;
    const bool own = !d_workspace || workspace_bytes < (size_t) (nb + 1) * sizeof(int);
    int* d_blk = (int*) d_workspace;
    if (own) SOBFU_HIP_TRY(hipMalloc((void**) &d_blk, (size_t) (nb + 1) * sizeof(int)));
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(nb), dim3(kBlock), 0, st, d_occupied + stride, count, d_blk);
    int rc = (int) hipGetLastError();
    if (rc == 0) rc = scan_in_place_sums(d_blk, nb, d_blk + nb, st);
    if (rc == 0) {
        hipLaunchKernelGGL(chunk_scan_kernel, dim3(nb), dim3(kBlock), 0, st, d_occupied + stride, count, d_blk, d_occupied + 2 * (size_t) stride);
        rc = (int) hipGetLastError();
    }
    if (rc == 0) rc = (int) hipMemcpyAsync(h_total_vertices, d_blk + nb, sizeof(int), hipMemcpyDeviceToHost, st);
    if (rc == 0) rc = (int) hipStreamSynchronize(st);
    if (own) (void) hipFree(d_blk);
    return rc;
}

int sobfu_hip_mc_generate_triangles(void* stream, const float* d_vol, int X, int Y, int Z, const int* d_occupied, int stride, int count,
                                    float size_x, float size_y, float size_z, const float R[9], const float t[3], float* d_vertices,
                                    float* d_normals, int max_vertices) {
    SOBFU_CHECK_ARGS(d_vol && d_occupied && R && t && d_vertices && d_normals && X > 0 && Y > 0 && Z > 0 && count >= 0 && stride >= count &&
                     max_vertices >= 0);
    if (count == 0) return 0;
    Pose p;
    for (int i = 0; i < 9; ++i) p.R[i] = R[i];
    for (int i = 0; i < 3; ++i) p.t[i] = t[i];
    hipLaunchKernelGGL(triangles_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t) stream, (const float2*) d_vol, Dims{X, Y, Z},
                       d_occupied, d_occupied + 2 * (size_t) stride, count, size_x / X, size_y / Y, size_z / Z, p, (float4*) d_vertices,
                       (float4*) d_normals, max_vertices);
    return (int) hipGetLastError();
}

}  // extern "C"
