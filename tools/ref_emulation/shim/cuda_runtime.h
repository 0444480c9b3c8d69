// Host stand-in for the CUDA runtime + device language, written for this repo (nothing here comes from CUDA or from the
// reference).  It exists so that tools/ref_emulation/build.py can compile the reference's own hot-path sources with g++
// and run them on the CPU to produce arrays (tests/golden/make_reference_fixtures.py).
//
// THIS IS SHIM EVIDENCE, NOT A REFERENCE BUILD: the headers below are stand-ins for a toolkit the image lacks, so by the
// task's rules nothing produced through them pins the oracle.  What it does give: the reference's own source lines
// (solver.cu, vector_fields.cu, reductor.cu, tsdf_volume.cu, imgproc.cu, utils.hpp, the host .cpp files) executing on the
// same inputs as the oracle and the HIP path, array for array.
//
// Arithmetic of the stand-ins (IEEE binary32, round-to-nearest-even, no flush-to-zero, nothing contracted; g++ is run with
// -ffp-contract=off):  __fadd_rn/__fsub_rn/__fmul_rn = the plain operation; __fmaf_rn = fmaf; __fdividef and '/' = IEEE
// divide (nvcc --prec-div=false would give <= 2 ulp); sqrtf/__fsqrt_rn = correctly rounded; __fsqrt_rd = round-down;
// __expf/powf/expf = glibc.  The distance of these choices to a real nvcc build is bounded separately by the oracle's
// SO_NVCC_MODE.
#pragma once
#include <math.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <type_traits>

#define __CUDACC__ 1
#ifdef CUEMU_ARCH
#define __CUDA_ARCH__ CUEMU_ARCH  // selects the branch an sm_61 build takes (shuffle tails in the reductions)
#endif
#define CUDART_VERSION 9000

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __constant__ static
#define __shared__ static thread_local  // a host thread runs one block at a time, so its function-local static IS the block's shared array
#define __launch_bounds__(...)

using std::isnan;
using std::max;
using std::min;
// CUDA's min/max also take mixed int / unsigned / enum arguments
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// ---------------------------------------------------------------- vector types
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint3 { unsigned int x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }

// ---------------------------------------------------------------- execution model (tools/ref_emulation/cuemu.cpp)
typedef struct cuemu_stream* cudaStream_t;
namespace cuemu {
extern thread_local uint3 tIdx, bIdx;  // blocks of a launch are spread over host threads (cuemu.cpp)
extern dim3 bDim, gDim;
extern int max_threads;                // host threads a launch may use (CUEMU_THREADS; 1 = blocks in launch order)
struct cfg {
    dim3 grid, block;
    size_t smem;
    cfg(dim3 g, dim3 b, size_t s = 0, cudaStream_t = 0) : grid(g), block(b), smem(s) {}
};
struct body {  // type-erased kernel call; avoids std::function so that the callee stays a plain function pointer + closure
    void (*fn)(void*);
    void* closure;
};
void launch_impl(const cfg& c, body b);
template <class F>
inline void launch(const cfg& c, F&& f) {
    launch_impl(c, body{[](void* p) { (*static_cast<typename std::remove_reference<F>::type*>(p))(); }, (void*) &f});
}
void syncthreads();                                 // a barrier = switch back to the block scheduler
void* dynamic_smem();                               // the launch's dynamic shared buffer
unsigned long long exchange(unsigned long long v, int src_lane);  // warp exchange: returns lane src_lane's v (own v if out of range)
unsigned int ballot(int predicate);
unsigned int ptx_special(const char* text);         // %laneid / %lanemask_lt
}  // namespace cuemu
#define threadIdx (cuemu::tIdx)
#define blockIdx (cuemu::bIdx)
#define blockDim (cuemu::bDim)
#define gridDim (cuemu::gDim)
static const int warpSize = 32;
#define CUEMU_LAUNCH(K, CFG, ARGS) cuemu::launch(cuemu::cfg CFG, [&] { K ARGS; })
static inline void __syncthreads() { cuemu::syncthreads(); }

template <class T>
static inline T __shfl_down_sync(unsigned, T v, unsigned int delta, int = 32) {
    static_assert(sizeof(T) <= 8, "");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    int lane = (int) ((threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x) & 31;
    bits     = cuemu::exchange(bits, lane + (int) delta);
    memcpy(&v, &bits, sizeof(T));
    return v;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int = 32) {
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    int lane = (int) ((threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x) & 31;
    bits     = cuemu::exchange(bits, lane ^ mask);
    memcpy(&v, &bits, sizeof(T));
    return v;
}
static inline unsigned int __ballot_sync(unsigned, int p) { return cuemu::ballot(p); }
static inline int __all_sync(unsigned, int p) { return cuemu::ballot(!p) == 0u; }
static inline int __popc(unsigned int v) { return __builtin_popcount(v); }
// the two PTX special-register reads in kfusion/cuda/temp_utils.hpp both assign to a local called 'ret'
#define asm(...) ret = cuemu::ptx_special(#__VA_ARGS__)

// ---------------------------------------------------------------- device math
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __fsqrt_rd(float a) {
    float r = sqrtf(a);  // correctly rounded to nearest; r*r is exact in double
    if ((double) r * (double) r > (double) a) r = nextafterf(r, 0.f);
    return r;
}
static inline int __float2int_rd(float a) { return (int) floorf(a); }
static inline int __float2int_rn(float a) { return (int) lrintf(a); }
static inline float __int_as_float(int v) {
    float f;
    memcpy(&f, &v, 4);
    return f;
}
#define __expf(x) expf(x)  // FUNCTIONS of these names collide with glibc's internal ones
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
static inline float rsqrtf(float a) { return 1.f / sqrtf(a); }
static inline float rsqrt(float a) { return 1.f / sqrtf(a); }
// CUDA resolves fma/sqrt/fabs/... on floats to the single-precision overloads; <math.h> under libstdc++ brings std::'s
// float overloads into the global namespace, which is what the reference's unqualified calls then pick up.
static_assert(std::is_same<decltype(fma(1.f, 1.f, 1.f)), float>::value, "fma(float...) must be single precision");
static_assert(std::is_same<decltype(sqrt(1.f)), float>::value, "sqrt(float) must be single precision");
static_assert(std::is_same<decltype(fabs(1.f)), float>::value, "fabs(float) must be single precision");
static_assert(std::is_same<decltype(fmax(1.f, 1.f)), float>::value, "fmax(float, float) must be single precision");

// ---------------------------------------------------------------- runtime API
enum cudaError_t { cudaSuccess = 0, cudaErrorUnknown = 1 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
enum cudaFuncCache { cudaFuncCachePreferNone, cudaFuncCachePreferShared, cudaFuncCachePreferL1 };
static inline const char* cudaGetErrorString(cudaError_t) { return "cuemu"; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaStreamCreate(cudaStream_t* s) { return *s = 0, cudaSuccess; }
static inline cudaError_t cudaStreamDestroy(cudaStream_t) { return cudaSuccess; }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
static inline cudaError_t cudaGetDevice(int* d) { return *d = 0, cudaSuccess; }
static inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
struct cudaDeviceProp {
    char name[64];
    int maxGridSize[3], maxThreadsPerBlock, major, minor, multiProcessorCount, warpSize;
    size_t totalGlobalMem, sharedMemPerBlock;
};
static inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->name, "cuemu");
    p->maxGridSize[0] = 2147483647;  // sm >= 3.0
    p->maxGridSize[1] = p->maxGridSize[2] = 65535;
    p->maxThreadsPerBlock = 1024;
    p->major = 6, p->minor = 1, p->warpSize = 32, p->multiProcessorCount = 1;
    return cudaSuccess;
}
template <class F>
static inline cudaError_t cudaFuncSetCacheConfig(F, cudaFuncCache) { return cudaSuccess; }

// The reference's convolution kernels read (never write) a few rows past the end of their source for dims that are not
// multiples of their tiles (solver.cu:243,254-257 precede the guard at :277), so every allocation gets zeroed slack.
namespace cuemu {
enum { kGuard = 4 << 20 };
void* dev_alloc(size_t bytes);
void dev_free(void* p);
}  // namespace cuemu
template <class T>
static inline cudaError_t cudaMalloc(T** p, size_t bytes) { return *p = (T*) cuemu::dev_alloc(bytes), cudaSuccess; }
static inline cudaError_t cudaFree(void* p) { return cuemu::dev_free(p), cudaSuccess; }
static inline cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t width_bytes, size_t rows) {
    *pitch = (width_bytes + 511) / 512 * 512;
    *p     = cuemu::dev_alloc(*pitch * rows);
    return cudaSuccess;
}
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { return memcpy(d, s, n), cudaSuccess; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { return memset(d, v, n), cudaSuccess; }
static inline cudaError_t cudaMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, cudaMemcpyKind) {
    for (size_t r = 0; r < h; ++r) memcpy((char*) d + r * dp, (const char*) s + r * sp, w);
    return cudaSuccess;
}
template <class T>
static inline cudaError_t cudaMemcpyToSymbol(T& sym, const void* s, size_t n, size_t off = 0, cudaMemcpyKind = cudaMemcpyHostToDevice) {
    return memcpy((char*) &sym + off, s, n), cudaSuccess;
}

template <class T>
static inline cudaError_t cudaMemcpyFromSymbol(void* d, const T& sym, size_t n, size_t off = 0, cudaMemcpyKind = cudaMemcpyDeviceToHost) {
    return memcpy(d, (const char*) &sym + off, n), cudaSuccess;
}
// blocks may run on different host threads: real atomics
static inline int atomicAdd(int* a, int v) { return __atomic_fetch_add(a, v, __ATOMIC_SEQ_CST); }
static inline unsigned int atomicInc(unsigned int* a, unsigned int limit) {
    unsigned int old = __atomic_load_n(a, __ATOMIC_SEQ_CST);
    while (!__atomic_compare_exchange_n(a, &old, old >= limit ? 0u : old + 1u, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

// ---------------------------------------------------------------- legacy texture references (point-sampled, zero border)
enum cudaTextureReadMode { cudaReadModeElementType, cudaReadModeNormalizedFloat };
enum cudaTextureFilterMode { cudaFilterModePoint, cudaFilterModeLinear };
enum cudaTextureAddressMode { cudaAddressModeWrap, cudaAddressModeClamp, cudaAddressModeMirror, cudaAddressModeBorder };
enum cudaChannelFormatKind { cudaChannelFormatKindSigned, cudaChannelFormatKindUnsigned, cudaChannelFormatKindFloat, cudaChannelFormatKindNone };
struct cudaChannelFormatDesc { int x, y, z, w; cudaChannelFormatKind f; };
static inline cudaChannelFormatDesc cudaCreateChannelDesc(int x, int y, int z, int w, cudaChannelFormatKind f) { return cudaChannelFormatDesc{x, y, z, w, f}; }
template <class T>
static inline cudaChannelFormatDesc cudaCreateChannelDesc() { return cudaChannelFormatDesc{(int) sizeof(T) * 8, 0, 0, 0, cudaChannelFormatKindNone}; }
struct textureReference {
    int normalized;
    cudaTextureFilterMode filterMode;
    cudaTextureAddressMode addressMode[3];
    cudaChannelFormatDesc channelDesc;
    mutable const void* ptr;
    mutable size_t width, height, pitch;
};
template <class T, int dim = 1, cudaTextureReadMode mode = cudaReadModeElementType>
struct texture : textureReference {
    texture(int norm = 0, cudaTextureFilterMode fm = cudaFilterModePoint, cudaTextureAddressMode am = cudaAddressModeClamp) {
        normalized = norm, filterMode = fm, addressMode[0] = addressMode[1] = addressMode[2] = am, ptr = 0, width = height = pitch = 0;
    }
    texture(int norm, cudaTextureFilterMode fm, cudaTextureAddressMode am, cudaChannelFormatDesc d) : texture(norm, fm, am) { channelDesc = d; }
};
template <class T, int dim, cudaTextureReadMode mode>
static inline cudaError_t cudaBindTexture2D(size_t* off, const texture<T, dim, mode>& t, const void* p, const cudaChannelFormatDesc&, size_t w, size_t h, size_t pitch) {
    if (off) *off = 0;
    t.ptr = p, t.width = w, t.height = h, t.pitch = pitch;
    return cudaSuccess;
}
template <class T, int dim, cudaTextureReadMode mode>
static inline cudaError_t cudaBindTexture(size_t* off, const texture<T, dim, mode>& t, const void* p, const cudaChannelFormatDesc&, size_t bytes = UINT_MAX) {
    if (off) *off = 0;
    t.ptr = p, t.width = bytes / sizeof(T), t.height = 1, t.pitch = bytes;
    return cudaSuccess;
}
static inline cudaError_t cudaUnbindTexture(const textureReference*) { return cudaSuccess; }
static inline cudaError_t cudaUnbindTexture(const textureReference&) { return cudaSuccess; }
template <class T>
static inline T tex2D(const texture<T, 2>& t, float x, float y) {
    // unnormalised coordinates, point filter: texel (floor x, floor y); border addressing: 0 outside
    float fx = floorf(x), fy = floorf(y);
    if (!(fx >= 0.f && fy >= 0.f && fx < (float) t.width && fy < (float) t.height)) return T();
    return *(const T*) ((const char*) t.ptr + (size_t) fy * t.pitch + (size_t) fx * sizeof(T));
}
template <class T>
static inline T tex1Dfetch(const texture<T, 1>& t, int i) { return ((const T*) t.ptr)[i]; }
