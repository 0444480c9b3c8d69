// Energy / max-norm reductions with the reference's exact tree shape, for gfx950.
// Reference: src/sobfu/cuda/reductor.cu (kernels), src/sobfu/reductor.cpp:38-94 (host finish),
// src/sobfu/precomp.cpp:20-43 (launch sizing).
//
// The float results of data_energy / reg_energy_sobolev depend on the association order, so the order is
// reproduced: per thread ((0 + e(i)) + e(i+T)) per grid stride, stride-halving pairwise tree T/2..1, block
// partials summed sequentially on the host.  Like the reference's sm>=30 path (reductor.cu:60-69) the tree runs
// through LDS only down to one wavefront; its last six levels (h = 32..1) are wave64 __shfl_down steps, which pair
// exactly the elements the shared-memory tail pairs (lane t takes lane t+h's value from before the level), so the
// bits do not depend on which tail runs.
#include <cmath>
#include <vector>

#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"
#include "sobfu_launch.hpp"

using namespace sobfu_hip;

namespace {

struct DataEl {  // reduce_data_kernel element (reductor.cu:26)
    const float2 *g, *n;
    SOBFU_DEV float operator()(size_t i) const {
        float d = g[i].x - n[i].x;
        return d * d;
    }
};
struct RegEl {  // reduce_reg_sobolev_kernel element (reductor.cu:129)
    const float4* J;
    SOBFU_DEV float operator()(size_t i) const {
        const float4* r = J + 4 * i;
        return norm_sq4(r[0]) + norm_sq4(r[1]) + norm_sq4(r[2]);
    }
};
struct RegFromPsiEl {  // same value, Jacobian (mode 1, vector_fields.cu:415-472) rebuilt in registers
    const float4* psi;
    Dims d;
    SOBFU_DEV float operator()(size_t i) const {
        int x = (int) (i % d.x), y = (int) ((i / d.x) % d.y), z = (int) (i / ((size_t) d.x * d.y));
        int x1 = x + 1, x2 = x - 1, y1 = y + 1, y2 = y - 1, z1 = z + 1, z2 = z - 1;
        if (x == 0) x2 = x + 1; else if (x == d.x - 1) x1 = x - 1;
        if (y == 0) y2 = y + 1; else if (y == d.y - 1) y1 = y - 1;
        if (z == 0) z2 = z + 1; else if (z == d.z - 1) z1 = z - 1;
        float4 jx = half4(sub4(disp_at(psi, d, x1, y, z), disp_at(psi, d, x2, y, z)));
        float4 jy = half4(sub4(disp_at(psi, d, x, y1, z), disp_at(psi, d, x, y2, z)));
        float4 jz = half4(sub4(disp_at(psi, d, x, y, z1), disp_at(psi, d, x, y, z2)));
        return norm_sq4(f4(jx.x, jy.x, jz.x)) + norm_sq4(f4(jx.y, jy.y, jz.y)) + norm_sq4(f4(jx.z, jy.z, jz.z));
    }
};

// the same two elements on the solver's iteration format (tsdf-only 4-byte volumes, 12-byte psi): identical values, so the
// energies printed at verbosity 1 / 2 do not need the API-format arrays
struct DataElC {
    const float *g, *n;
    SOBFU_DEV float operator()(size_t i) const {
        float d = g[i] - n[i];
        return d * d;
    }
};
struct RegFromPsi3El {
    const float* psi;  // 3 floats per voxel
    Dims d;
    SOBFU_DEV float4 disp(int x, int y, int z) const {
        const float* p = psi + 3 * vidx(d, x, y, z);
        return sub4(f4(p[0], p[1], p[2]), f4((float) x, (float) y, (float) z));
    }
    SOBFU_DEV float operator()(size_t i) const {
        int x = (int) (i % d.x), y = (int) ((i / d.x) % d.y), z = (int) (i / ((size_t) d.x * d.y));
        int x1 = x + 1, x2 = x - 1, y1 = y + 1, y2 = y - 1, z1 = z + 1, z2 = z - 1;
        if (x == 0) x2 = x + 1; else if (x == d.x - 1) x1 = x - 1;
        if (y == 0) y2 = y + 1; else if (y == d.y - 1) y1 = y - 1;
        if (z == 0) z2 = z + 1; else if (z == d.z - 1) z1 = z - 1;
        float4 jx = half4(sub4(disp(x1, y, z), disp(x2, y, z)));
        float4 jy = half4(sub4(disp(x, y1, z), disp(x, y2, z)));
        float4 jz = half4(sub4(disp(x, y, z1), disp(x, y, z2)));
        return norm_sq4(f4(jx.x, jy.x, jz.x)) + norm_sq4(f4(jx.y, jy.y, jz.y)) + norm_sq4(f4(jx.z, jy.z, jz.z));
    }
};

template <class El>
__global__ void __launch_bounds__(512) tree_sum_kernel(El el, float* __restrict__ partials, size_t n) {
    __shared__ float s[512];
    const unsigned T = blockDim.x, tid = threadIdx.x;
    const size_t grid = (size_t) T * 2 * gridDim.x;
    float my = 0.f;
    for (size_t i = (size_t) blockIdx.x * T * 2 + tid; i < n; i += grid) {
        my += el(i);
        if (i + T < n) my += el(i + T);
    }
    s[tid] = my;
    __syncthreads();
    for (unsigned h = T / 2; h >= 64; h >>= 1) {  // levels that span more than one wavefront: through LDS
        if (tid < h) s[tid] = my = my + s[tid + h];
        __syncthreads();
    }
    if (tid < 64) {  // wavefront tail (reductor.cu:60-69)
#pragma unroll
        for (unsigned h = 32; h >= 1; h >>= 1) {
            const float o = __shfl_down(my, h, 64);
            if (h < T && tid < h) my = my + o;
        }
    }
    if (tid == 0) partials[blockIdx.x] = my;
}

// reduce_max_kernel -- reductor.cu:342-456 (strict '>' everywhere: first in scan order wins)
__global__ void __launch_bounds__(512) tree_max_kernel(const float4* __restrict__ updates, float2* __restrict__ partials, size_t n) {
    __shared__ float2 s[512];
    const unsigned T = blockDim.x, tid = threadIdx.x;
    const size_t grid = (size_t) T * 2 * gridDim.x;
    float2 lm = make_float2(0.f, 0.f);
    for (size_t i = (size_t) blockIdx.x * T * 2 + tid; i < n; i += grid) {
        float v = norm4(updates[i]);
        if (v > lm.x) lm = make_float2(v, (float) (unsigned) i);
        if (i + T < n) {
            float w = norm4(updates[i + T]);
            if (w > lm.x) lm = make_float2(w, (float) (unsigned) i + T);  // reductor.cu:371
        }
    }
    s[tid] = lm;
    __syncthreads();
    for (unsigned h = T / 2; h >= 64; h >>= 1) {
        if (tid < h && s[tid + h].x > lm.x) s[tid] = lm = s[tid + h];
        __syncthreads();
    }
    if (tid < 64) {  // wavefront tail: same pairs, same strict '>' (the lower lane keeps ties)
#pragma unroll
        for (unsigned h = 32; h >= 1; h >>= 1) {
            const float ox = __shfl_down(lm.x, h, 64), oy = __shfl_down(lm.y, h, 64);
            if (h < T && tid < h && ox > lm.x) lm = make_float2(ox, oy);
        }
    }
    if (tid == 0) partials[blockIdx.x] = lm;
}

int next_pow2(int x) {  // precomp.cpp:8-18
    if (x < 0) return 0;
    --x;
    x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
    return x + 1;
}

template <class El>
int run_sum(El el, int n, void* d_scratch, float* out, void* stream) {
    int blocks, threads;
    SOBFU_TRY(sobfu_hip_reduce_config(n, &blocks, &threads));
    hipLaunchKernelGGL(tree_sum_kernel<El>, dim3(blocks), dim3(threads), 0, (hipStream_t) stream, el, (float*) d_scratch, (size_t) n);
    SOBFU_HIP_TRY(hipGetLastError());
    std::vector<float> h(blocks);
    SOBFU_HIP_TRY(hipMemcpyAsync(h.data(), d_scratch, sizeof(float) * blocks, hipMemcpyDeviceToHost, (hipStream_t) stream));
    SOBFU_HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    volatile float r = 0.f;  // sequential float sum, reductor.cpp:68-79
    for (int i = 0; i < blocks; ++i) r = r + h[i];
    *out = 0.5f * r;  // reductor.cpp:42,49
    return 0;
}

}  // namespace

namespace sobfu_hip {
int data_energy_tsdf(const float* g1, const float* f1, int n, void* d_scratch, float* out, hipStream_t stream) {
    return run_sum(DataElC{g1, f1}, n, d_scratch, out, stream);
}
int reg_energy_from_psi3(const float* psi3, int X, int Y, int Z, void* d_scratch, float* out, hipStream_t stream) {
    return run_sum(RegFromPsi3El{psi3, Dims{X, Y, Z}}, X * Y * Z, d_scratch, out, stream);
}
}  // namespace sobfu_hip

extern "C" {

int sobfu_hip_reduce_config(int n, int* blocks, int* threads) {
    SOBFU_CHECK_ARGS(n > 0 && blocks && threads);
    const int maxThreads = 512, maxBlocks = 65536;  // reductor.cpp:16
    int t = (n < maxThreads * 2) ? next_pow2((n + 1) / 2) : maxThreads;
    int b = (n + (t * 2 - 1)) / (t * 2);
    *blocks  = b < maxBlocks ? b : maxBlocks;
    *threads = t;
    return 0;
}

int sobfu_hip_data_energy(const float* d_phi_global, const float* d_phi_n, int n, void* d_scratch, float* out, void* stream) {
    SOBFU_CHECK_ARGS(d_phi_global && d_phi_n && d_scratch && out && n > 0);
    return run_sum(DataEl{(const float2*) d_phi_global, (const float2*) d_phi_n}, n, d_scratch, out, stream);
}

int sobfu_hip_reg_energy_sobolev(const float* d_J, int n, void* d_scratch, float* out, void* stream) {
    SOBFU_CHECK_ARGS(d_J && d_scratch && out && n > 0);
    return run_sum(RegEl{(const float4*) d_J}, n, d_scratch, out, stream);
}

int sobfu_hip_reg_energy_sobolev_from_psi(const float* d_psi, int X, int Y, int Z, void* d_scratch, float* out, void* stream) {
    SOBFU_CHECK_ARGS(d_psi && d_scratch && out && X > 1 && Y > 1 && Z > 1);
    return run_sum(RegFromPsiEl{(const float4*) d_psi, Dims{X, Y, Z}}, X * Y * Z, d_scratch, out, stream);
}

int sobfu_hip_max_update_norm(const float* d_updates, int n, void* d_scratch, float out[2], void* stream) {
    SOBFU_CHECK_ARGS(d_updates && d_scratch && out && n > 0);
    int blocks, threads;
    SOBFU_TRY(sobfu_hip_reduce_config(n, &blocks, &threads));
    hipLaunchKernelGGL(tree_max_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t) stream, (const float4*) d_updates,
                       (float2*) d_scratch, (size_t) n);
    SOBFU_HIP_TRY(hipGetLastError());
    std::vector<float2> h(blocks);
    SOBFU_HIP_TRY(hipMemcpyAsync(h.data(), d_scratch, sizeof(float2) * blocks, hipMemcpyDeviceToHost, (hipStream_t) stream));
    SOBFU_HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    float2 r = make_float2(0.f, 0.f);  // final_reduce_max, reductor.cpp:81-94
    for (int i = 0; i < blocks; ++i)
        if (h[i].x > r.x) r = h[i];
    out[0] = r.x;
    out[1] = r.y;
    return 0;
}

}  // extern "C"
