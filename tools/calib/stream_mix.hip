// What a plain streaming kernel reaches on this part with the SAME read / write mix and access widths as the fused passes --
// the practical HBM ceiling their launch times are compared with in DESIGN.md (the 8 TB/s of the data sheet is not reachable by
// any kernel; a 1 : 1 copy is not the passes' mix either).  One lane per voxel, x fastest, no reuse, no arithmetic:
//   mix_a   reads 12 + 4 + 4 B (psi, F, phi_global)   writes 12 B (nabla_U)            = pass A's compulsory streams
//   mix_b   reads 12 + 12 B (nabla_U, psi)            writes 12 + 4 B (psi, F)         = pass B's compulsory streams
//   *_nt    the same with the nontemporal hints the passes use
//   copy_x3 / read_x3 / write_x3  one stream per direction, for reference
// Timed with HIP events around `reps` back-to-back launches of one pattern; prints GB/s (bytes moved / time).
//   hipcc --offload-arch=gfx950 -O3 tools/calib/stream_mix.hip -o build/stream_mix && build/stream_mix [voxels] [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v3f __attribute__((ext_vector_type(3)));
typedef v3f __attribute__((aligned(4))) v3f_u;

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                              \
        }                                                              \
    } while (0)

struct Arrays {
    float *v0, *v1, *v2, *v3;  // 12-byte fields
    float *f0, *f1, *f2;       // 4-byte fields
};

template <bool NT>
__device__ __forceinline__ v3f ld3(const float* p, size_t i) {
    return NT ? __builtin_nontemporal_load((const v3f_u*) (p + 3 * i)) : *(const v3f_u*) (p + 3 * i);
}
template <bool NT>
__device__ __forceinline__ void st3(float* p, size_t i, v3f v) {
    if (NT) __builtin_nontemporal_store(v, (v3f_u*) (p + 3 * i));
    else *(v3f_u*) (p + 3 * i) = v;
}
template <bool NT>
__device__ __forceinline__ float ld1(const float* p, size_t i) {
    return NT ? __builtin_nontemporal_load(p + i) : p[i];
}
template <bool NT>
__device__ __forceinline__ void st1(float* p, size_t i, float v) {
    if (NT) __builtin_nontemporal_store(v, p + i);
    else p[i] = v;
}

template <bool NT>
__global__ void __launch_bounds__(256) mix_a(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v3f psi = ld3<false>(a.v0, i);
    const float f = ld1<false>(a.f0, i), g = ld1<NT>(a.f1, i);  // pass A: phi_global is the nontemporal one
    st3<false>(a.v1, i, psi * (f - g));
}
template <bool NT, bool REV = false>
__global__ void __launch_bounds__(256) mix_b(Arrays a, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (REV) i = (size_t) (gridDim.x - 1 - blockIdx.x) * 256 + threadIdx.x;  // highest addresses first (n is a multiple of 256)
    const v3f u = ld3<false>(a.v1, i), psi = ld3<NT>(a.v0, i);  // pass B: psi load, psi store and F store are nontemporal
    const v3f p = psi - u;
    st3<NT>(a.v2, i, p);
    st1<NT>(a.f2, i, p.x + p.y + p.z);
}
__global__ void __launch_bounds__(256) copy_x3(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) st3<false>(a.v2, i, ld3<false>(a.v0, i));
}
__global__ void __launch_bounds__(256) read_x3(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v3f v = ld3<false>(a.v0, i);
    if (v.x + v.y + v.z == 12345.678f) a.f2[0] = 1.f;  // never true: keeps the load alive
}
__global__ void __launch_bounds__(256) write_x3(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) st3<false>(a.v2, i, v3f{1.f, 2.f, 3.f});
}

// ---- the same with hints that REACH the hardware (round 6).  hipcc drops the nontemporal flag of __builtin_nontemporal_load/_store on the
// 4-byte-aligned 12-byte vector type (found in the ISA in round 5), so the `_nt` patterns above stream their 12-byte accesses plainly.  Below they
// go through buffer instructions, whose cache-policy operand carries the hint -- as the solver's pass B does (NTBUF) -- and a 16-byte copy shows
// what the launcher-shaped kernels' float4 streams reach.
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int) bytes, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ v3f bld3(__amdgpu_buffer_rsrc_t r, size_t i) {
    const v3u t = __builtin_amdgcn_raw_buffer_load_b96(r, (int) (12 * i), 0, NT ? 2 : 0);
    return v3f{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z)};
}
template <bool NT>
__device__ __forceinline__ void bst3(__amdgpu_buffer_rsrc_t r, size_t i, v3f v) {
    const v3u t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
    __builtin_amdgcn_raw_buffer_store_b96(t, r, (int) (12 * i), 0, NT ? 2 : 0);
}
// LOADS: which of pass B's two 12-byte loads stream (bit 0: nabla_U, bit 1: psi); STORES: psi and F stores stream
template <int LOADS, bool STORES>
__global__ void __launch_bounds__(256) mix_b_buf(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const __amdgpu_buffer_rsrc_t r0 = rsrc(a.v0, 12 * n), r1 = rsrc(a.v1, 12 * n), r2 = rsrc(a.v2, 12 * n);
    const v3f u = bld3<(LOADS & 1) != 0>(r1, i), psi = bld3<(LOADS & 2) != 0>(r0, i);
    const v3f p = psi - u;
    bst3<STORES>(r2, i, p);
    st1<STORES>(a.f2, i, p.x + p.y + p.z);
}
template <bool NT_LOADS, bool NT_STORE>
__global__ void __launch_bounds__(256) mix_a_buf(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const __amdgpu_buffer_rsrc_t r0 = rsrc(a.v0, 12 * n), r1 = rsrc(a.v1, 12 * n);
    const v3f psi = bld3<NT_LOADS>(r0, i);
    const float f = ld1<NT_LOADS>(a.f0, i), g = ld1<true>(a.f1, i);
    bst3<NT_STORE>(r1, i, psi * (f - g));
}
template <bool NT_LOAD, bool NT_STORE>
__global__ void __launch_bounds__(256) copy_x3_buf(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) bst3<NT_STORE>(rsrc(a.v2, 12 * n), i, bld3<NT_LOAD>(rsrc(a.v0, 12 * n), i));
}
// 16-byte elements over the first 3/4 of the same arrays (n * 12 bytes hold 3 n / 4 float4s)
template <bool NT_LOAD, bool NT_STORE>
__global__ void __launch_bounds__(256) copy_x4(Arrays a, size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n / 4 * 3) return;
    const v4f v = NT_LOAD ? __builtin_nontemporal_load((const v4f*) a.v0 + i) : ((const v4f*) a.v0)[i];
    if (NT_STORE) __builtin_nontemporal_store(v, (v4f*) a.v2 + i);
    else ((v4f*) a.v2)[i] = v;
}

template <class K>
static void run(const char* name, K kernel, const Arrays& a, size_t n, int reps, double bytes_per_voxel) {
    const dim3 grid((unsigned) ((n + 255) / 256)), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> us;
    for (int round = 0; round < 5; ++round) {
        hipLaunchKernelGGL(kernel, grid, block, 0, 0, a, n);  // warm
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, grid, block, 0, 0, a, n);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(1e3f * ms / reps);
    }
    std::sort(us.begin(), us.end());
    const double t = us[us.size() / 2];
    std::printf("%-22s %6.1f B/voxel  %7.1f us/launch  %7.1f GB/s  (min %.1f, max %.1f us)\n", name, bytes_per_voxel, t,
                bytes_per_voxel * n / t * 1e-3, us.front(), us.back());
}

// a pass-A-like launch followed by a pass-B-like launch that reads what it wrote (v1), as in the solver's iteration: does the
// consumer find the producer's output in the Infinity Cache, and does it matter which end it starts from?
template <class KA, class KB>
static void run_pair(const char* name, KA ka, KB kb, const Arrays& a, size_t n, int reps, double bytes_per_voxel) {
    const dim3 grid((unsigned) ((n + 255) / 256)), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> us;
    for (int round = 0; round < 5; ++round) {
        hipLaunchKernelGGL(ka, grid, block, 0, 0, a, n);
        hipLaunchKernelGGL(kb, grid, block, 0, 0, a, n);
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(ka, grid, block, 0, 0, a, n);
            hipLaunchKernelGGL(kb, grid, block, 0, 0, a, n);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(1e3f * ms / reps);
    }
    std::sort(us.begin(), us.end());
    const double t = us[us.size() / 2];
    std::printf("%-22s %6.1f B/voxel  %7.1f us/pair    %7.1f GB/s  (min %.1f, max %.1f us)\n", name, bytes_per_voxel, t,
                bytes_per_voxel * n / t * 1e-3, us.front(), us.back());
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t) std::atoll(argv[1]) : ((size_t) 1 << 24);  // 256^3 voxels
    const int reps = argc > 2 ? std::atoi(argv[2]) : 20;
    Arrays a{};
    float** v[4] = {&a.v0, &a.v1, &a.v2, &a.v3};
    float** f[3] = {&a.f0, &a.f1, &a.f2};
    for (auto p : v) {
        CK(hipMalloc((void**) p, n * 12));
        CK(hipMemset(*p, 0, n * 12));
    }
    for (auto p : f) {
        CK(hipMalloc((void**) p, n * 4));
        CK(hipMemset(*p, 0, n * 4));
    }
    CK(hipDeviceSynchronize());
    std::printf("stream_mix: %zu voxels, %d launches per timing, median of 5 timings\n", n, reps);
    run("copy_x3", copy_x3, a, n, reps, 24);
    run("read_x3", read_x3, a, n, reps, 12);
    run("write_x3", write_x3, a, n, reps, 12);
    run("mix_a", mix_a<false>, a, n, reps, 32);
    run("mix_a_nt", mix_a<true>, a, n, reps, 32);
    run("mix_b", mix_b<false>, a, n, reps, 40);
    run("mix_b_nt", mix_b<true>, a, n, reps, 40);
    run("mix_b_nt_rev", mix_b<true, true>, a, n, reps, 40);
    std::printf("-- hints that reach the hardware: 12-byte accesses through buffer instructions, 16-byte through the builtin\n");
    run("copy_x3 buf plain", copy_x3_buf<false, false>, a, n, reps, 24);
    run("copy_x3 buf nt st", copy_x3_buf<false, true>, a, n, reps, 24);
    run("copy_x3 buf nt ld+st", copy_x3_buf<true, true>, a, n, reps, 24);
    run("copy_x4 plain", copy_x4<false, false>, a, n, reps, 24);  // (3/4 n float4s = the same 24 B per voxel of n)
    run("copy_x4 nt st", copy_x4<false, true>, a, n, reps, 24);
    run("copy_x4 nt ld+st", copy_x4<true, true>, a, n, reps, 24);
    run("mix_a buf plain", mix_a_buf<false, false>, a, n, reps, 32);
    run("mix_a buf nt st", mix_a_buf<false, true>, a, n, reps, 32);
    run("mix_a buf nt all", mix_a_buf<true, true>, a, n, reps, 32);
    run("mix_b buf plain", mix_b_buf<0, false>, a, n, reps, 40);
    run("mix_b buf as pass B", mix_b_buf<2, true>, a, n, reps, 40);  // psi load + both stores stream, nabla_U load plain
    run("mix_b buf nt all", mix_b_buf<3, true>, a, n, reps, 40);
    run_pair("mix_a_nt + mix_b_nt", mix_a<true>, mix_b<true>, a, n, reps, 72);
    run_pair("mix_a_nt + mix_b_rev", mix_a<true>, mix_b<true, true>, a, n, reps, 72);
    return 0;
}
