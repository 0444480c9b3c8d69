// sobfu_amd/sobfu.hpp -- C++ shells that rebuild the reference's host class surface for the solver hot path on
// top of the C ABI (include/sobfu_hip.h).  Header-only, C++14, host code only (g++ or hipcc); links against
// libsobfu_hip.so and libamdhip64.so.
//
// Mirrors, with the same names, argument meaning and error behaviour:
//   Params                                   include/sobfu/params.hpp:7-37
//   kfusion::Intr                            include/kfusion/types.hpp, src/kfusion/precomp.cpp:9-16
//   kfusion::cuda::DeviceMemory (CudaData)   include/kfusion/cuda/device_memory.hpp:20-102
//   kfusion::cuda::DeviceArray2D<T>          include/kfusion/cuda/device_array.hpp (subset: create/upload/download/ptr/step)
//   kfusion::cuda::TsdfVolume                include/kfusion/cuda/tsdf_volume.hpp:17-92
//   kfusion::cuda::{depthBilateralFilter, depthTruncation, computeDists, waitAllDefaultStream}
//                                            include/kfusion/cuda/imgproc.hpp:11-28
//   kfusion::device::{TsdfVolume POD, clear_volume, integrate x2, init_*}  include/kfusion/internal.hpp:59-78,189-198
//   sobfu::device::{VectorField & typedefs, Jacobian, clear, init_identity, apply, estimate_inverse,
//                   TsdfDifferentiator, SecondOrderDifferentiator, Differentiator, Reductor, launchers}
//                                            include/sobfu/vector_fields.hpp:117-241, reductor.hpp:24-50, solver.hpp:109-136
//   sobfu::cuda::{VectorField, DeformationField, Jacobian, Solver}
//                                            include/sobfu/vector_fields.hpp:14-112, solver.hpp:52-101
//
// Errors: like the reference (src/kfusion/device_memory.cpp:7-10, include/kfusion/safe_call.hpp:12-22) any device
// error prints "error: <msg>\t<file>:<line>" and calls exit(0).  Define SOBFU_AMD_THROW to get std::runtime_error.
#pragma once

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <vector>

#include "sobfu_hip.h"

// ---- minimal stand-ins for the OpenCV types that leak into the reference's API ----------------------------------
#ifndef SOBFU_AMD_HAVE_OPENCV
namespace cv {
template <class T, int N>
struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(); }
    Vec(T a, T b, T c) { static_assert(N == 3, "3-vector ctor"); val[0] = a; val[1] = b; val[2] = c; }
    static Vec all(T v) { Vec r; for (int i = 0; i < N; ++i) r.val[i] = v; return r; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
    template <class U> operator Vec<U, N>() const { Vec<U, N> r; for (int i = 0; i < N; ++i) r.val[i] = (U) val[i]; return r; }
};
typedef Vec<int, 3> Vec3i;
typedef Vec<float, 3> Vec3f;
// rigid transform x -> R x + t (only what the path uses: translate, inv, composition)
struct Affine3f {
    float R[9];
    float t[3];
    Affine3f() { std::memset(R, 0, sizeof R); R[0] = R[4] = R[8] = 1.f; t[0] = t[1] = t[2] = 0.f; }
    static Affine3f Identity() { return Affine3f(); }
    Affine3f translate(const Vec3f& d) const { Affine3f a = *this; for (int i = 0; i < 3; ++i) a.t[i] += d[i]; return a; }
    Affine3f inv() const {  // rigid: R^T, -R^T t
        Affine3f a;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a.R[3 * i + j] = R[3 * j + i];
        for (int i = 0; i < 3; ++i) a.t[i] = -(a.R[3 * i] * t[0] + a.R[3 * i + 1] * t[1] + a.R[3 * i + 2] * t[2]);
        return a;
    }
    Affine3f operator*(const Affine3f& b) const {  // this o b
        Affine3f a;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            a.R[3 * i + j] = 0.f;
            for (int k = 0; k < 3; ++k) a.R[3 * i + j] += R[3 * i + k] * b.R[3 * k + j];
        }
        for (int i = 0; i < 3; ++i) a.t[i] = R[3 * i] * b.t[0] + R[3 * i + 1] * b.t[1] + R[3 * i + 2] * b.t[2] + t[i];
        return a;
    }
};
template <class T>
struct Ptr : std::shared_ptr<T> {
    Ptr() {}
    Ptr(T* p) : std::shared_ptr<T>(p) {}
    Ptr(const std::shared_ptr<T>& p) : std::shared_ptr<T>(p) {}
};
// dense n-D host array, as far as the reference's tests use cv::Mat (test/deformation_field_test.cpp:96-105): the n-D constructor,
// ptr<T>() as a download target, at<T>(i0, i1, i2) with the LAST index fastest
#ifndef CV_32F
#define CV_8U 0
#define CV_16U 2
#define CV_32S 4
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)
#endif
class Mat {
public:
    Mat() : dims(0), type_(0) {}
    Mat(int ndims, const int* sizes, int type) : dims(ndims), type_(type), size_(sizes, sizes + ndims) {
        size_t n = elemSize();
        for (int i = 0; i < ndims; ++i) n *= (size_t) sizes[i];
        data_ = std::make_shared<std::vector<unsigned char>>(n);
    }
    size_t elemSize() const {
        static const size_t depth_bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
        return depth_bytes[type_ & 7] * (size_t) ((type_ >> 3) + 1);
    }
    int type() const { return type_; }
    template <class T> T* ptr() { return reinterpret_cast<T*>(data_->data()); }
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(data_->data()); }
    template <class T> T& at(int i0, int i1, int i2) { return ptr<T>()[((size_t) i0 * size_[1] + i1) * size_[2] + i2]; }
    template <class T> const T& at(int i0, int i1, int i2) const { return ptr<T>()[((size_t) i0 * size_[1] + i1) * size_[2] + i2]; }
    int dims;

private:
    int type_;
    std::vector<int> size_;
    std::shared_ptr<std::vector<unsigned char>> data_;
};
}  // namespace cv
#endif

// Mat4f of the reference (include/kfusion/internal.hpp:12-14)
struct Mat4f {
    float4 data[4];
};

namespace kfusion {
typedef cv::Vec3i Vec3i;
typedef cv::Vec3f Vec3f;
typedef cv::Affine3f Affine3f;

struct Intr {
    float fx, fy, cx, cy;
    Intr() : fx(0), fy(0), cx(0), cy(0) {}
    Intr(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
    Intr operator()(int level) const { int d = 1 << level; return Intr(fx / d, fy / d, cx / d, cy / d); }
};

namespace cuda {
// error(): print and exit(0), as src/kfusion/device_memory.cpp:7-10
inline void error(const char* msg, const char* file, int line, const char* func = "") {
#ifdef SOBFU_AMD_THROW
    throw std::runtime_error(std::string(msg) + " at " + file + ":" + std::to_string(line) + " " + func);
#else
    std::printf("error: %s\t%s:%d\n", msg, file, line);
    std::fflush(stdout);
    std::exit(0);
#endif
}
}  // namespace cuda
}  // namespace kfusion

#define sobfuSafeCall(expr)                                                                        \
    do {                                                                                           \
        int _rc = (int) (expr);                                                                    \
        if (_rc != 0) ::kfusion::cuda::error(sobfu_hip_error_string(_rc), __FILE__, __LINE__, ""); \
    } while (0)
#define cudaSafeCall sobfuSafeCall  // the reference's spelling (include/kfusion/safe_call.hpp)

// ---- Params (include/sobfu/params.hpp) ---------------------------------------------------------------------------
struct Params {
    int cols = 640, rows = 480;
    cv::Vec3i volume_dims;
    cv::Vec3f volume_size;
    cv::Affine3f volume_pose;
    kfusion::Intr intr;
    float icp_truncate_depth_dist = 0.f;
    float bilateral_sigma_depth = 0.f, bilateral_sigma_spatial = 0.f;
    int bilateral_kernel_size = 0;
    float tsdf_trunc_dist = 0.f, eta = 0.f;
    float tsdf_max_weight = 0.f;
    float gradient_delta_factor = 0.f;
    int start_frame = 0;
    int verbosity = 0;
    int s = 7, max_iter = 0;
    float max_update_norm = 0.f, lambda = 0.1f, alpha = 0.f, w_reg = 0.f;
    cv::Vec3f voxel_sizes() const {
        return cv::Vec3f(volume_size[0] / volume_dims[0], volume_size[1] / volume_dims[1], volume_size[2] / volume_dims[2]);
    }
};

namespace kfusion {

// ---- ScopeTime / SampledScopeTime (include/kfusion/types.hpp:101-122, src/kfusion/core.cpp:214-234) ---------------
// Same console lines as the reference ("Time(name) = X ms", and every 34th sampled scope "avg. frame time = ...ms
// (...fps)"); the tick source is std::chrono::steady_clock instead of cv::getTickCount.
struct ScopeTime {
    const char* name;
    double start;
    static double now_ms() {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    explicit ScopeTime(const char* name_) : name(name_), start(now_ms()) {}
    ~ScopeTime() { std::cout << "Time(" << name << ") = " << now_ms() - start << "ms" << std::endl; }
};

struct SampledScopeTime {
    enum { EACH = 34 };
    explicit SampledScopeTime(double& time_ms) : time_ms_(time_ms), start(ScopeTime::now_ms()) {}
    ~SampledScopeTime() {
        static int scopes = 0;  // one counter for the process, like the reference's function-local static
        time_ms_ += ScopeTime::now_ms() - start;
        if (scopes % EACH == 0 && scopes) {
            std::cout << "avg. frame time = " << time_ms_ / EACH << "ms (" << 1000.f * EACH / time_ms_ << "fps)" << std::endl;
            time_ms_ = 0.0;
        }
        ++scopes;
    }
    SampledScopeTime(const SampledScopeTime&) = delete;
    SampledScopeTime& operator=(const SampledScopeTime&) = delete;

private:
    double& time_ms_;
    double start;
};

namespace cuda {

inline void setDevice(int device) { sobfuSafeCall(hipSetDevice(device)); }
inline void waitAllDefaultStream() { sobfuSafeCall(hipDeviceSynchronize()); }
inline void printShortCudaDeviceInfo(int device) {
    hipDeviceProp_t p;
    sobfuSafeCall(hipGetDeviceProperties(&p, device));
    std::printf("[%s] %d CUs, %.1f GB\n", p.name, p.multiProcessorCount, p.totalGlobalMem / 1e9);
}
// remaining device queries of include/kfusion/kinfu.hpp:25-30, answered by the HIP runtime
inline int getCudaEnabledDeviceCount() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
inline std::string getDeviceName(int device) {
    hipDeviceProp_t p;
    sobfuSafeCall(hipGetDeviceProperties(&p, device));
    return p.name;
}
inline bool checkIfPreFermiGPU(int) { return false; }  // the demo's "GPU too old" gate (demo.cpp:575): never true on CDNA
inline void printCudaDeviceInfo(int device) {
    hipDeviceProp_t p;
    sobfuSafeCall(hipGetDeviceProperties(&p, device));
    std::printf("Device %d: \"%s\" (%s)\n  compute units %d, wavefront %d, clock %.0f MHz\n  global memory %.1f GB, L2 %d KB, LDS per workgroup %zu KB\n",
                device, p.name, p.gcnArchName, p.multiProcessorCount, p.warpSize, p.clockRate / 1e3, p.totalGlobalMem / 1e9,
                p.l2CacheSize / 1024, p.sharedMemPerBlock / 1024);
}

// ---- DeviceMemory: a shared device allocation (public surface of include/kfusion/cuda/device_memory.hpp:20-102) ---
// Ownership is a std::shared_ptr whose deleter is hipFree: copies share the block, the last owner frees it; a block handed in by the
// caller (pointer + size) is viewed, never owned.  create(n) keeps the block when the size already matches, create(0) is a no-op.
class DeviceMemory {
public:
    DeviceMemory() = default;
    explicit DeviceMemory(size_t n) { create(n); }
    DeviceMemory(void* p, size_t n) : view_(p), bytes_(n) {}  // the caller's buffer
    void create(size_t n) {
        if (n == bytes_ || n == 0) return;
        void* p = nullptr;
        sobfuSafeCall(hipMalloc(&p, n));
        block_.reset(p, [](void* q) { (void) hipFree(q); });  // (a deleter must not throw)
        view_  = p;
        bytes_ = n;
    }
    void release() {
        block_.reset();
        view_  = nullptr;
        bytes_ = 0;
    }
    void copyTo(DeviceMemory& other) const {
        if (empty()) { other.release(); return; }
        other.create(bytes_);
        sobfuSafeCall(hipMemcpy(other.view_, view_, bytes_, hipMemcpyDeviceToDevice));
    }
    void upload(const void* host, size_t n) { create(n); sobfuSafeCall(hipMemcpy(view_, host, n, hipMemcpyHostToDevice)); }
    void download(void* host) const { sobfuSafeCall(hipMemcpy(host, view_, bytes_, hipMemcpyDeviceToHost)); }
    void swap(DeviceMemory& o) { block_.swap(o.block_); std::swap(view_, o.view_); std::swap(bytes_, o.bytes_); }
    template <class T> T* ptr() { return (T*) view_; }
    template <class T> const T* ptr() const { return (const T*) view_; }
    bool empty() const { return view_ == nullptr; }
    size_t sizeBytes() const { return bytes_; }

private:
    std::shared_ptr<void> block_;  // the owner (empty for a caller's buffer)
    void* view_   = nullptr;
    size_t bytes_ = 0;
};
typedef DeviceMemory CudaData;

// ---- DeviceArray2D<T>: the subset the path uses (rows x cols image with a byte step) ----------------------------
template <class T>
class DeviceArray2D {
public:
    DeviceArray2D() : rows_(0), cols_(0), step_(0) {}
    DeviceArray2D(int rows, int cols) : rows_(0), cols_(0), step_(0) { create(rows, cols); }
    void create(int rows, int cols) {
        if (rows == rows_ && cols == cols_) return;
        rows_ = rows; cols_ = cols;
        step_ = ((size_t) cols * sizeof(T) + 255) / 256 * 256;  // 256-B aligned rows (cudaMallocPitch-like)
        mem_.create(step_ * rows);
    }
    void upload(const void* host, size_t host_step, int rows, int cols) {
        create(rows, cols);
        sobfuSafeCall(hipMemcpy2D(mem_.ptr<void>(), step_, host, host_step, (size_t) cols * sizeof(T), rows, hipMemcpyHostToDevice));
    }
    void download(void* host, size_t host_step) const {
        sobfuSafeCall(hipMemcpy2D(host, host_step, mem_.ptr<void>(), step_, (size_t) cols_ * sizeof(T), rows_, hipMemcpyDeviceToHost));
    }
    T* ptr() { return mem_.ptr<T>(); }
    const T* ptr() const { return mem_.ptr<T>(); }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    size_t step() const { return step_; }
    bool empty() const { return mem_.empty(); }

private:
    DeviceMemory mem_;
    int rows_, cols_;
    size_t step_;
};
typedef DeviceArray2D<unsigned short> Depth;
typedef DeviceArray2D<float> Dists;

// ---- DeviceArray<T>: typed 1-D view of a DeviceMemory blob (include/kfusion/cuda/device_array.hpp:20-90) --------
template <class T>
class DeviceArray : public DeviceMemory {
public:
    DeviceArray() {}
    explicit DeviceArray(size_t n) : DeviceMemory(n * sizeof(T)) {}
    DeviceArray(T* p, size_t n) : DeviceMemory(p, n * sizeof(T)) {}  // user buffer
    void create(size_t n) { DeviceMemory::create(n * sizeof(T)); }
    void upload(const T* host, size_t n) { DeviceMemory::upload(host, n * sizeof(T)); }
    void upload(const std::vector<T>& v) { upload(v.data(), v.size()); }
    void download(T* host) const { DeviceMemory::download(host); }
    void download(std::vector<T>& v) const { v.resize(size()); if (!v.empty()) download(v.data()); }
    T* ptr() { return DeviceMemory::ptr<T>(); }
    const T* ptr() const { return DeviceMemory::ptr<T>(); }
    size_t size() const { return sizeBytes() / sizeof(T); }
};

// point / normal element of the mesh buffers: the reference aliases pcl::PointXYZ / pcl::Normal onto float4
// (marching_cubes.cpp:63-65, kfusion::device::PointType)
typedef float4 Point;
typedef float4 Normal;
struct Surface {  // include/kfusion/types.hpp:85-88
    DeviceArray<Point> vertices;
    DeviceArray<Normal> normals;
};

// ---- image pre-steps (include/kfusion/cuda/imgproc.hpp:11-28, src/kfusion/imgproc.cpp:3-41) ---------------------
inline void depthBilateralFilter(const Depth& in, Depth& out, int kernel_size, float sigma_spatial, float sigma_depth) {
    out.create(in.rows(), in.cols());
    sobfuSafeCall(sobfu_hip_bilateral_filter(in.ptr(), (int) in.step(), out.ptr(), (int) out.step(), in.rows(), in.cols(),
                                             kernel_size, sigma_spatial, sigma_depth, nullptr));
}
inline void depthTruncation(Depth& depth, float threshold) {
    sobfuSafeCall(sobfu_hip_truncate_depth(depth.ptr(), (int) depth.step(), depth.rows(), depth.cols(), threshold, nullptr));
}
inline void computeDists(const Depth& depth, Dists& dists, const Intr& intr) {
    dists.create(depth.rows(), depth.cols());
    sobfuSafeCall(sobfu_hip_compute_dists(depth.ptr(), (int) depth.step(), dists.ptr(), (int) dists.step(), depth.rows(),
                                          depth.cols(), intr.fx, intr.fy, intr.cx, intr.cy, nullptr));
}
}  // namespace cuda

// ---- device PODs + launchers (include/kfusion/internal.hpp:59-78,189-198) ----------------------------------------
namespace device {
typedef int3 Vec3i;
typedef float3 Vec3f;
struct Mat3f { float3 data[3]; };
struct Aff3f { Mat3f R; Vec3f t; };
struct Projector {
    float2 f, c;
    Projector() {}
    Projector(float fx, float fy, float cx, float cy) { f.x = fx; f.y = fy; c.x = cx; c.y = cy; }
};
struct TsdfVolume {
    float2* const data;
    const int3 dims;
    const float3 voxel_size;
    const float trunc_dist, eta, max_weight;
    TsdfVolume(float2* d, int3 dm, float3 vs, float trunc, float eta_, float maxw)
        : data(d), dims(dm), voxel_size(vs), trunc_dist(trunc), eta(eta_), max_weight(maxw) {}
};
inline void clear_volume(TsdfVolume& v) { sobfuSafeCall(sobfu_hip_clear_volume((float*) v.data, v.dims.x, v.dims.y, v.dims.z, nullptr)); }
inline void integrate(TsdfVolume& g, TsdfVolume& n) {
    sobfuSafeCall(sobfu_hip_integrate_fuse((float*) g.data, (const float*) n.data, g.dims.x, g.dims.y, g.dims.z, g.max_weight, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());  // tsdf_volume.cu:172
}
inline void integrate(const cuda::Dists& dists, TsdfVolume& v, const Aff3f& a, const Projector& p) {
    float R[9] = {a.R.data[0].x, a.R.data[0].y, a.R.data[0].z, a.R.data[1].x, a.R.data[1].y, a.R.data[1].z,
                  a.R.data[2].x, a.R.data[2].y, a.R.data[2].z};
    float t[3] = {a.t.x, a.t.y, a.t.z}, vs[3] = {v.voxel_size.x, v.voxel_size.y, v.voxel_size.z};
    sobfuSafeCall(sobfu_hip_integrate_depth(dists.ptr(), (int) dists.step(), dists.rows(), dists.cols(), (float*) v.data, v.dims.x,
                                            v.dims.y, v.dims.z, vs, v.trunc_dist, v.eta, R, t, p.f.x, p.f.y, p.c.x, p.c.y, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());  // tsdf_volume.cu:161
}
#define SOBFU_AMD_VS(v) float vs[3] = {(v).voxel_size.x, (v).voxel_size.y, (v).voxel_size.z}
inline void init_sphere(TsdfVolume& v, const float3& c, const float& r) {
    SOBFU_AMD_VS(v); float cc[3] = {c.x, c.y, c.z};
    sobfuSafeCall(sobfu_hip_init_sphere((float*) v.data, v.dims.x, v.dims.y, v.dims.z, vs, v.trunc_dist, v.eta, cc, r, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());
}
inline void init_box(TsdfVolume& v, const float3& b) {
    SOBFU_AMD_VS(v); float bb[3] = {b.x, b.y, b.z};
    sobfuSafeCall(sobfu_hip_init_box((float*) v.data, v.dims.x, v.dims.y, v.dims.z, vs, v.trunc_dist, bb, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());
}
inline void init_ellipsoid(TsdfVolume& v, const float3& r) {
    SOBFU_AMD_VS(v); float rr[3] = {r.x, r.y, r.z};
    sobfuSafeCall(sobfu_hip_init_ellipsoid((float*) v.data, v.dims.x, v.dims.y, v.dims.z, vs, v.trunc_dist, rr, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());
}
inline void init_plane(TsdfVolume& v, const float& z) {
    SOBFU_AMD_VS(v);
    sobfuSafeCall(sobfu_hip_init_plane((float*) v.data, v.dims.x, v.dims.y, v.dims.z, vs, v.trunc_dist, z, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());
}
inline void init_torus(TsdfVolume& v, const float2& t) {
    SOBFU_AMD_VS(v); float tt[2] = {t.x, t.y};
    sobfuSafeCall(sobfu_hip_init_torus((float*) v.data, v.dims.x, v.dims.y, v.dims.z, vs, v.trunc_dist, tt, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());
}
#undef SOBFU_AMD_VS
}  // namespace device

template <class D, class S> inline D device_cast(const S& s);
template <> inline int3 device_cast<int3, cv::Vec3i>(const cv::Vec3i& v) { int3 r; r.x = v[0]; r.y = v[1]; r.z = v[2]; return r; }
template <> inline float3 device_cast<float3, cv::Vec3f>(const cv::Vec3f& v) { float3 r; r.x = v[0]; r.y = v[1]; r.z = v[2]; return r; }
template <> inline device::Aff3f device_cast<device::Aff3f, cv::Affine3f>(const cv::Affine3f& a) {
    device::Aff3f r;
    for (int i = 0; i < 3; ++i) { r.R.data[i].x = a.R[3 * i]; r.R.data[i].y = a.R[3 * i + 1]; r.R.data[i].z = a.R[3 * i + 2]; }
    r.t.x = a.t[0]; r.t.y = a.t[1]; r.t.z = a.t[2];
    return r;
}

namespace cuda {
// ---- TsdfVolume host owner (include/kfusion/cuda/tsdf_volume.hpp:17-92, src/kfusion/tsdf_volume.cpp:18-146) -----
class TsdfVolume {
public:
    explicit TsdfVolume(const Params& p)
        : trunc_dist_(p.tsdf_trunc_dist), eta_(p.eta), max_weight_(p.tsdf_max_weight), dims_(p.volume_dims), size_(p.volume_size),
          pose_(p.volume_pose), gradient_delta_factor_(p.gradient_delta_factor), raycast_step_factor_(0.f) { create(dims_); }
    virtual ~TsdfVolume() {}
    void create(const Vec3i& dims) {
        dims_ = dims;
        data_.create((size_t) dims_[0] * dims_[1] * dims_[2] * 2 * sizeof(float));
        clear();
    }
    Vec3i getDims() const { return dims_; }
    Vec3f getVoxelSize() const { return Vec3f(size_[0] / dims_[0], size_[1] / dims_[1], size_[2] / dims_[2]); }
    const CudaData data() const { return data_; }
    CudaData data() { return data_; }
    Vec3f getSize() const { return size_; }
    void setSize(const Vec3f& s) { size_ = s; }
    float getTruncDist() const { return trunc_dist_; }
    void setTruncDist(float& d) { trunc_dist_ = d; }
    float getEta() const { return eta_; }
    void setEta(float& e) { eta_ = e; }
    float getMaxWeight() const { return max_weight_; }
    void setMaxWeight(float& w) { max_weight_ = w; }
    Affine3f getPose() const { return pose_; }
    void setPose(const Affine3f& p) { pose_ = p; }
    float getGradientDeltaFactor() const { return gradient_delta_factor_; }
    void setGradientDeltaFactor(float& f) { gradient_delta_factor_ = f; }
    float getRaycastStepFactor() const { return raycast_step_factor_; }
    void setRaycastStepFactor(float& f) { raycast_step_factor_ = f; }
    virtual void clear() { device::TsdfVolume v = pod(); device::clear_volume(v); }
    void swap(CudaData& d) { data_.swap(d); }
    virtual void applyAffine(const Affine3f& a) { pose_ = a * pose_; }
    virtual void integrate(const TsdfVolume& phi_n_psi) {
        device::TsdfVolume g = pod();
        device::TsdfVolume n((float2*) phi_n_psi.data_.ptr<float2>(), g.dims, g.voxel_size, trunc_dist_, eta_, max_weight_);
        device::integrate(g, n);
    }
    virtual void integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr) {
        Affine3f vol2cam = camera_pose.inv() * pose_;  // src/kfusion/tsdf_volume.cpp:96
        device::TsdfVolume v = pod();
        device::integrate(dists, v, device_cast<device::Aff3f>(vol2cam), device::Projector(intr.fx, intr.fy, intr.cx, intr.cy));
    }
    virtual void initBox(const float3& b) { device::TsdfVolume v = pod(); device::init_box(v, b); }
    virtual void initEllipsoid(const float3& r) { device::TsdfVolume v = pod(); device::init_ellipsoid(v, r); }
    virtual void initPlane(const float& z) { device::TsdfVolume v = pod(); device::init_plane(v, z); }
    virtual void initSphere(const float3& c, const float& r) { device::TsdfVolume v = pod(); device::init_sphere(v, c, r); }
    virtual void initTorus(const float2& t) { device::TsdfVolume v = pod(); device::init_torus(v, t); }
    device::TsdfVolume pod() {
        return device::TsdfVolume(data_.ptr<float2>(), device_cast<int3>(dims_), device_cast<float3>(getVoxelSize()), trunc_dist_, eta_, max_weight_);
    }

private:
    CudaData data_;
    float trunc_dist_, eta_, max_weight_;
    Vec3i dims_;
    Vec3f size_;
    Affine3f pose_;
    float gradient_delta_factor_, raycast_step_factor_;
};
}  // namespace cuda

namespace device {
// ---- marching cubes launchers (include/kfusion/internal.hpp:213-225) ---------------------------------------------
typedef float4 PointType;
inline void bindTextures(const int*, const int*, const int*) {}  // the case table lives in the library's constant data
inline void unbindTextures() {}
// occupied_voxels: 3 x cols ints (voxel index / vertex count / vertex offset); row stride in ints = step() / 4
// `scratch` (optional, beyond the reference's signature): a buffer the caller keeps between frames for the scan steps
// (cuda::MarchingCubes owns one) -- without it every call allocates and frees its scratch
inline int getOccupiedVoxels(const TsdfVolume& v, cuda::DeviceArray2D<int>& occupied_voxels, cuda::DeviceArray<unsigned char>* scratch = nullptr) {
    int n = 0;
    if (scratch && scratch->size() < sobfu_hip_mc_workspace_bytes(v.dims.x, v.dims.y, v.dims.z))
        scratch->create(sobfu_hip_mc_workspace_bytes(v.dims.x, v.dims.y, v.dims.z));
    sobfuSafeCall(sobfu_hip_mc_occupied_voxels(nullptr, (const float*) v.data, v.dims.x, v.dims.y, v.dims.z, occupied_voxels.ptr(),
                                               (int) (occupied_voxels.step() / sizeof(int)), occupied_voxels.cols(), &n,
                                               scratch ? scratch->ptr() : nullptr, scratch ? scratch->size() : 0));
    return n;
}
inline int computeOffsetsAndTotalVertices(cuda::DeviceArray2D<int>& occupied_voxels, int active_voxels, cuda::DeviceArray<unsigned char>* scratch = nullptr) {
    int total = 0;
    sobfuSafeCall(sobfu_hip_mc_offsets(nullptr, occupied_voxels.ptr(), (int) (occupied_voxels.step() / sizeof(int)), active_voxels, &total,
                                       scratch ? scratch->ptr() : nullptr, scratch ? scratch->size() : 0));
    return total;
}
inline void generateTriangles(const TsdfVolume& v, const cuda::DeviceArray2D<int>& occupied_voxels, int active_voxels, const float3& volume_size,
                              const Aff3f& pose, cuda::DeviceArray<PointType>& out_vertices, cuda::DeviceArray<PointType>& out_normals) {
    const float R[9] = {pose.R.data[0].x, pose.R.data[0].y, pose.R.data[0].z, pose.R.data[1].x, pose.R.data[1].y, pose.R.data[1].z,
                        pose.R.data[2].x, pose.R.data[2].y, pose.R.data[2].z};
    const float t[3] = {pose.t.x, pose.t.y, pose.t.z};
    sobfuSafeCall(sobfu_hip_mc_generate_triangles(nullptr, (const float*) v.data, v.dims.x, v.dims.y, v.dims.z, occupied_voxels.ptr(),
                                                  (int) (occupied_voxels.step() / sizeof(int)), active_voxels, volume_size.x, volume_size.y,
                                                  volume_size.z, R, t, (float*) out_vertices.ptr(), (float*) out_normals.ptr(),
                                                  (int) std::min(out_vertices.size(), out_normals.size())));
    sobfuSafeCall(hipDeviceSynchronize());  // marching_cubes.cu:312
}
}  // namespace device

namespace cuda {
// ---- MarchingCubes (include/kfusion/cuda/marching_cubes.hpp:17-61, src/kfusion/marching_cubes.cpp:14-79) ------------
class MarchingCubes {
public:
    enum { POINTS_PER_TRIANGLE = 3, DEFAULT_TRIANGLES_BUFFER_SIZE = 2 * 1000 * 1000 * POINTS_PER_TRIANGLE };
    typedef std::shared_ptr<MarchingCubes> Ptr;
    MarchingCubes() : pose(Affine3f::Identity()) {}
    void setPose(const Affine3f& pose_) { pose = pose_; }
    Surface run(const TsdfVolume& volume, DeviceArray<Point>& vertices_buffer, DeviceArray<Normal>& normals_buffer) {
        if (vertices_buffer.empty()) vertices_buffer.create(DEFAULT_TRIANGLES_BUFFER_SIZE);
        if (normals_buffer.empty()) normals_buffer.create(DEFAULT_TRIANGLES_BUFFER_SIZE);
        occupied_voxels_buffer_.create(3, (int) (vertices_buffer.size() / 3));
        device::TsdfVolume vol = const_cast<TsdfVolume&>(volume).pod();
        const int active_voxels = device::getOccupiedVoxels(vol, occupied_voxels_buffer_, &scan_scratch_);
        std::cout << "no. of active voxels: " << active_voxels << std::endl;  // marching_cubes.cpp:49
        if (!active_voxels) return Surface();
        int total_vertices = device::computeOffsetsAndTotalVertices(occupied_voxels_buffer_, active_voxels, &scan_scratch_);
        const int cap = (int) (std::min(vertices_buffer.size(), normals_buffer.size()) / 3 * 3);  // whole triangles that fit
        if (total_vertices > cap) total_vertices = cap;
        device::generateTriangles(vol, occupied_voxels_buffer_, active_voxels, device_cast<float3>(volume.getSize()),
                                  device_cast<device::Aff3f>(pose), vertices_buffer, normals_buffer);
        Surface s;
        s.vertices = DeviceArray<Point>(vertices_buffer.ptr(), (size_t) total_vertices);
        s.normals  = DeviceArray<Normal>(normals_buffer.ptr(), (size_t) total_vertices);
        return s;
    }

private:
    DeviceArray2D<int> occupied_voxels_buffer_;
    DeviceArray<unsigned char> scan_scratch_;  // kept between frames: the scan steps allocate nothing per call
    Affine3f pose;
};
}  // namespace cuda
}  // namespace kfusion

namespace sobfu_amd {
// Host triangle soup standing in for pcl::PolygonMesh (three consecutive vertices per polygon, sob_fusion.cpp:160-183)
struct TriangleMesh {
    std::vector<float4> vertices;
    size_t triangles() const { return vertices.size() / 3; }
    bool empty() const { return vertices.empty(); }
};
// Legacy-ASCII VTK polydata, the sections pcl::io::saveVTKFile writes for a PolygonMesh (demo.cpp:244): POINTS, VERTICES, POLYGONS
inline bool write_vtk(const std::string& path, const TriangleMesh& m) {
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) return false;
    const size_t n = m.vertices.size(), nt = m.triangles();
    std::fprintf(f, "# vtk DataFile Version 3.0\nvtk output\nASCII\nDATASET POLYDATA\nPOINTS %zu float\n", n);
    for (const float4& v : m.vertices) std::fprintf(f, "%.9g %.9g %.9g\n", v.x, v.y, v.z);
    std::fprintf(f, "\nVERTICES %zu %zu\n", n, 2 * n);
    for (size_t i = 0; i < n; ++i) std::fprintf(f, "1 %zu\n", i);
    std::fprintf(f, "\nPOLYGONS %zu %zu\n", nt, 4 * nt);
    for (size_t i = 0; i < nt; ++i) std::fprintf(f, "3 %zu %zu %zu\n", 3 * i, 3 * i + 1, 3 * i + 2);
    return std::fclose(f) == 0;
}
}  // namespace sobfu_amd

// ================================================================================================================
// sobfu
// ================================================================================================================
namespace sobfu {
namespace device {

// VectorField POD + typedefs (include/sobfu/vector_fields.hpp:124-136,146,164,191,216)
struct VectorField {
    VectorField(float4* const d, const int3 dm) : data(d), dims(dm) {}
    float4* const data;
    const int3 dims;
};
typedef VectorField DeformationField;
typedef VectorField TsdfGradient;
typedef VectorField Laplacian;
typedef VectorField PotentialGradient;
struct Jacobian {
    Jacobian(Mat4f* const d, int3 dm) : data(d), dims(dm) {}
    Mat4f* const data;
    const int3 dims;
};

inline void clear(VectorField& f) { sobfuSafeCall(sobfu_hip_clear_field((float*) f.data, f.dims.x, f.dims.y, f.dims.z, nullptr)); }
inline void clear(Jacobian& J) {
    sobfuSafeCall(sobfu_hip_clear_jacobian((float*) J.data, J.dims.x, J.dims.y, J.dims.z, nullptr));
    sobfuSafeCall(hipDeviceSynchronize());
}
inline void init_identity(DeformationField& psi) { sobfuSafeCall(sobfu_hip_init_identity((float*) psi.data, psi.dims.x, psi.dims.y, psi.dims.z, nullptr)); }
inline void apply(const kfusion::device::TsdfVolume& phi, kfusion::device::TsdfVolume& warped, const DeformationField& psi) {
    sobfuSafeCall(sobfu_hip_apply((const float*) phi.data, (float*) warped.data, (const float*) psi.data, phi.dims.x, phi.dims.y, phi.dims.z, nullptr));
}
inline void estimate_inverse(DeformationField& psi, DeformationField& psi_inv) {
    sobfuSafeCall(sobfu_hip_estimate_inverse((const float*) psi.data, (float*) psi_inv.data, psi.dims.x, psi.dims.y, psi.dims.z, 48, nullptr));
}

struct TsdfDifferentiator {
    explicit TsdfDifferentiator(kfusion::device::TsdfVolume& v) : vol(v) {}
    void calculate(TsdfGradient& g) {
        sobfuSafeCall(sobfu_hip_tsdf_gradient((const float*) vol.data, (float*) g.data, g.dims.x, g.dims.y, g.dims.z, nullptr));
    }
    kfusion::device::TsdfVolume vol;
};
struct SecondOrderDifferentiator {
    explicit SecondOrderDifferentiator(DeformationField& p) : psi(p) {}
    void calculate(Laplacian& L) { sobfuSafeCall(sobfu_hip_laplacian((const float*) psi.data, (float*) L.data, L.dims.x, L.dims.y, L.dims.z, nullptr)); }
    DeformationField psi;
};
struct Differentiator {
    explicit Differentiator(DeformationField& p) : psi(p) {}
    void calculate(Jacobian& J) { sobfuSafeCall(sobfu_hip_jacobian((const float*) psi.data, (float*) J.data, J.dims.x, J.dims.y, J.dims.z, 0, nullptr)); }
    void calculate_deformation_jacobian(Jacobian& J) {
        sobfuSafeCall(sobfu_hip_jacobian((const float*) psi.data, (float*) J.data, J.dims.x, J.dims.y, J.dims.z, 1, nullptr));
    }
    DeformationField psi;
};

// solver launchers (include/sobfu/solver.hpp:109-136).  The reference keeps the taps in a __constant__ symbol set by
// set_convolution_kernel (solver.cu:229-234); the C ABI takes them by value, so the shell keeps the last-set taps.
inline float* conv_taps() { static float taps[7] = {0, 0, 0, 1, 0, 0, 0}; return taps; }
inline void set_convolution_kernel(float* d_kernel) {
    sobfuSafeCall(hipMemcpy(conv_taps(), d_kernel, 7 * sizeof(float), hipMemcpyDeviceToHost));
}
inline void convolution_rows(float4* dst, float4* src, int w, int h, int d) { sobfuSafeCall(sobfu_hip_convolution_rows((float*) dst, (const float*) src, conv_taps(), w, h, d, nullptr)); }
inline void convolution_columns(float4* dst, float4* src, int w, int h, int d) { sobfuSafeCall(sobfu_hip_convolution_columns((float*) dst, (const float*) src, conv_taps(), w, h, d, nullptr)); }
inline void convolution_depth(float4* dst, float4* src, int w, int h, int d) { sobfuSafeCall(sobfu_hip_convolution_depth((float*) dst, (const float*) src, conv_taps(), w, h, d, nullptr)); }
inline void calculate_potential_gradient(kfusion::device::TsdfVolume& pnp, kfusion::device::TsdfVolume& pg, TsdfGradient& g, Laplacian& L,
                                         PotentialGradient& nU, float w_reg) {
    sobfuSafeCall(sobfu_hip_potential_gradient((const float*) pnp.data, (const float*) pg.data, (const float*) g.data, (const float*) L.data,
                                               (float*) nU.data, w_reg, pnp.dims.x, pnp.dims.y, pnp.dims.z, nullptr));
}
inline void update_psi(DeformationField& psi, PotentialGradient& nUS, float4* updates, float alpha) {
    sobfuSafeCall(sobfu_hip_update_psi((float*) psi.data, (const float*) nUS.data, (float*) updates, alpha, psi.dims.x, psi.dims.y, psi.dims.z, nullptr));
}

// Reductor (include/sobfu/reductor.hpp:24-50, src/sobfu/reductor.cpp)
struct Reductor {
    Reductor(int3 dims_, float vsz_, float trunc_dist_) : dims(dims_), vsz(vsz_), trunc_dist(trunc_dist_) {
        no_voxels = dims.x * dims.y * dims.z;
        sobfuSafeCall(sobfu_hip_reduce_config(no_voxels, &blocks, &threads));
        sobfuSafeCall(hipMalloc(&scratch, (size_t) blocks * 8));
        sobfuSafeCall(hipMalloc((void**) &updates, (size_t) no_voxels * sizeof(float4)));
    }
    ~Reductor() { (void) hipFree(scratch); (void) hipFree(updates); }
    Reductor(const Reductor&) = delete;
    Reductor& operator=(const Reductor&) = delete;
    float data_energy(float2* phi_global, float2* phi_n) {
        float e;
        sobfuSafeCall(sobfu_hip_data_energy((const float*) phi_global, (const float*) phi_n, no_voxels, scratch, &e, nullptr));
        return e;
    }
    float reg_energy_sobolev(Mat4f* J) {
        float e;
        sobfuSafeCall(sobfu_hip_reg_energy_sobolev((const float*) J, no_voxels, scratch, &e, nullptr));
        return e;
    }
    float2 max_update_norm() {
        float o[2];
        sobfuSafeCall(sobfu_hip_max_update_norm((const float*) updates, no_voxels, scratch, o, nullptr));
        float2 r; r.x = o[0]; r.y = o[1];
        return r;
    }
    int3 dims;
    float vsz, trunc_dist;
    int no_voxels, blocks, threads;
    float4* updates;
    void* scratch;
};
}  // namespace device

namespace cuda {
// host owners (include/sobfu/vector_fields.hpp:20-112, src/sobfu/vector_fields.cpp)
class VectorField {
public:
    explicit VectorField(cv::Vec3i d) : dims(d) {
        data.create((size_t) dims[0] * dims[1] * dims[2] * sizeof(float4));
        clear();
    }
    virtual ~VectorField() {}
    cv::Vec3i get_dims() const { return dims; }
    kfusion::cuda::CudaData get_data() { return data; }
    const kfusion::cuda::CudaData get_data() const { return data; }
    void set_data(kfusion::cuda::CudaData& d) { data = d; }
    void clear() { device::VectorField f(data.ptr<float4>(), kfusion::device_cast<int3>(dims)); device::clear(f); }
    void print() {  // debug listing of the non-zero vectors (src/sobfu/vector_fields.cpp:31-54), x outermost like the reference
        const size_t X = (size_t) dims[0], Y = (size_t) dims[1], Z = (size_t) dims[2];
        std::vector<float4> h(X * Y * Z);
        data.download(h.data());
        std::cout << "--- FIELD ---" << std::endl;
        for (size_t i = 0; i < X; ++i)
            for (size_t j = 0; j < Y; ++j)
                for (size_t k = 0; k < Z; ++k) {
                    const float4& v = h[i + X * (j + Y * k)];
                    if (std::fabs(v.x) > 1e-5f || std::fabs(v.y) > 1e-5f || std::fabs(v.z) > 1e-5f)
                        std::cout << "(x,y,z)=(" << i << ", " << j << ", " << k << "), (u,v,w)=(" << v.x << ", " << v.y << "," << v.z << ")"
                                  << std::endl;
                }
    }
    int get_no_nans() {  // debug helper of the reference (src/sobfu/vector_fields.cpp:56-79)
        size_t n = (size_t) dims[0] * dims[1] * dims[2];
        std::unique_ptr<float4[]> h(new float4[n]);
        data.download(h.get());
        int c = 0;
        for (size_t i = 0; i < n; ++i) c += (h[i].x != h[i].x || h[i].y != h[i].y || h[i].z != h[i].z);
        return c;
    }

protected:
    kfusion::cuda::CudaData data;
    cv::Vec3i dims;
};
typedef VectorField TsdfGradient;
typedef VectorField Laplacian;
typedef VectorField PotentialGradient;

class DeformationField : public VectorField {
public:
    explicit DeformationField(cv::Vec3i d) : VectorField(d) { clear(); }
    void clear() { device::DeformationField p(data.ptr<float4>(), kfusion::device_cast<int3>(dims)); device::init_identity(p); }
    void apply(const cv::Ptr<kfusion::cuda::TsdfVolume> phi, cv::Ptr<kfusion::cuda::TsdfVolume> phi_psi) {
        kfusion::device::TsdfVolume a = phi->pod(), b = phi_psi->pod();
        device::DeformationField p(data.ptr<float4>(), a.dims);
        device::apply(a, b, p);
        kfusion::cuda::waitAllDefaultStream();  // src/sobfu/vector_fields.cpp:122
    }
    void get_inverse(DeformationField& psi_inv) {
        int3 d = kfusion::device_cast<int3>(dims);
        device::DeformationField a(data.ptr<float4>(), d), b(psi_inv.data.ptr<float4>(), d);
        device::estimate_inverse(a, b);
    }
};

class Jacobian {
public:
    explicit Jacobian(cv::Vec3i d) : dims(d) {
        data.create((size_t) dims[0] * dims[1] * dims[2] * sizeof(Mat4f));
        clear();
    }
    kfusion::cuda::CudaData get_data() { return data; }
    void clear() { device::Jacobian J(data.ptr<Mat4f>(), kfusion::device_cast<int3>(dims)); device::clear(J); }

private:
    kfusion::cuda::CudaData data;
    cv::Vec3i dims;
};

// SpatialGradients (include/sobfu/vector_fields.hpp:102-112, src/sobfu/vector_fields.cpp:148-165): the reference's solver
// workspace -- six vector fields and two Jacobians (5.1 GB at 256^3, half of it never touched).  Kept for source
// compatibility only: Solver below does NOT allocate it (its workspace is one nabla_U field + the compact state).
struct SpatialGradients {
    explicit SpatialGradients(cv::Vec3i d)
        : nabla_phi_n(new TsdfGradient(d)), nabla_phi_n_o_psi(new TsdfGradient(d)), J(new Jacobian(d)), J_inv(new Jacobian(d)),
          L(new Laplacian(d)), L_o_psi_inv(new Laplacian(d)), nabla_U(new PotentialGradient(d)), nabla_U_S(new PotentialGradient(d)) {}
    ~SpatialGradients() {
        delete nabla_phi_n; delete nabla_phi_n_o_psi; delete J; delete J_inv; delete L; delete L_o_psi_inv; delete nabla_U; delete nabla_U_S;
    }
    SpatialGradients(const SpatialGradients&) = delete;
    SpatialGradients& operator=(const SpatialGradients&) = delete;
    TsdfGradient *nabla_phi_n, *nabla_phi_n_o_psi;
    Jacobian *J, *J_inv;
    Laplacian *L, *L_o_psi_inv;
    PotentialGradient *nabla_U, *nabla_U_S;
};

// Solver (include/sobfu/solver.hpp:52-101, src/sobfu/solver.cpp:7-101): the workspace + hot loop live behind the
// opaque C handle; estimate_psi has the reference's signature and side effects (mutates psi, psi_inv, phi_n_psi,
// phi_global_psi_inv) and prints the reference's progress lines to stdout.
class Solver {
public:
    explicit Solver(Params& p) : h_(nullptr) {
        sobfu_hip_solver_params sp;
        sp.verbosity = p.verbosity; sp.max_iter = p.max_iter; sp.s = p.s; sp.max_update_norm = p.max_update_norm;
        sp.lambda = p.lambda; sp.alpha = p.alpha; sp.w_reg = p.w_reg;
        sobfuSafeCall(sobfu_hip_solver_create(&h_, p.volume_dims[0], p.volume_dims[1], p.volume_dims[2], &sp));
    }
    ~Solver() { sobfu_hip_solver_destroy(h_); }  // (the reference's defaulted dtor leaks everything, solver.cpp:67)
    Solver(const Solver&) = delete;
    Solver& operator=(const Solver&) = delete;
    void estimate_psi(const cv::Ptr<kfusion::cuda::TsdfVolume> phi_global, cv::Ptr<kfusion::cuda::TsdfVolume> phi_global_psi_inv,
                      const cv::Ptr<kfusion::cuda::TsdfVolume> phi_n, cv::Ptr<kfusion::cuda::TsdfVolume> phi_n_psi,
                      std::shared_ptr<DeformationField> psi, std::shared_ptr<DeformationField> psi_inv) {
        sobfuSafeCall(sobfu_hip_solver_estimate_psi(h_, phi_global->data().ptr<float>(), phi_global_psi_inv->data().ptr<float>(),
                                                    phi_n->data().ptr<float>(), phi_n_psi->data().ptr<float>(),
                                                    psi->get_data().ptr<float>(), psi_inv->get_data().ptr<float>(), &last_report,
                                                    nullptr, nullptr));
    }
    sobfu_hip_solver* handle() { return h_; }
    sobfu_hip_solver_report last_report{};

private:
    sobfu_hip_solver* h_;
};
}  // namespace cuda
}  // namespace sobfu

// ================================================================================================================
// Headless frame driver: SobFusion::operator() without PCL / viz / marching cubes, and the
// .ini parameter schema of the reference app.
// ================================================================================================================
#include <fstream>
#include <map>
#include <sstream>

namespace sobfu_amd {

// Reads the reference's parameter file (params/*.ini; schema src/apps/demo.cpp:87-160, derived values :71-74).
// Unknown keys (e.g. RHO_0 in params_boxing.ini:40, which makes the reference's own parser throw) are ignored; '#' starts a
// comment anywhere on a line, as in boost::program_options' config-file parser.  Returns false (reason in *why) if the file
// cannot be opened, if one of the keys the reference reads unconditionally (TSDF_TRUNC_DIST, ETA, VOL_POSE_T_Z --
// vm[...].as<float>() throws there, demo.cpp:71-74) is missing, or if the volume dims / size / truncation distance are not
// positive (a zero truncation distance would fill the volume with inf / NaN).
inline bool read_params_ini(const std::string& path, Params& p, std::map<std::string, std::string>* raw = nullptr, std::string* why = nullptr) {
    std::ifstream f(path);
    if (!f) {
        if (why) *why = "cannot open " + path;
        return false;
    }
    std::map<std::string, std::string> kv;
    std::string line;
    while (std::getline(f, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line.erase(h);
        size_t e = line.find('=');
        if (e == std::string::npos) continue;
        auto trim = [](std::string s) {
            size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
            return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
        };
        kv[trim(line.substr(0, e))] = trim(line.substr(e + 1));
    }
    auto F = [&](const char* k, float& v) { auto it = kv.find(k); if (it != kv.end()) v = std::strtof(it->second.c_str(), nullptr); };
    auto I = [&](const char* k, int& v) { auto it = kv.find(k); if (it != kv.end()) v = (int) std::strtol(it->second.c_str(), nullptr, 10); };
    I("VOL_DIMS_X", p.volume_dims[0]); I("VOL_DIMS_Y", p.volume_dims[1]); I("VOL_DIMS_Z", p.volume_dims[2]);
    F("VOL_SIZE_X", p.volume_size[0]); F("VOL_SIZE_Y", p.volume_size[1]); F("VOL_SIZE_Z", p.volume_size[2]);
    F("TSDF_MAX_WEIGHT", p.tsdf_max_weight);
    F("GRADIENT_DELTA_FACTOR", p.gradient_delta_factor);
    F("INTR_FX", p.intr.fx); F("INTR_FY", p.intr.fy); F("INTR_CX", p.intr.cx); F("INTR_CY", p.intr.cy);
    F("TRUNC_DEPTH", p.icp_truncate_depth_dist);
    F("BILATERAL_SIGMA_DEPTH", p.bilateral_sigma_depth); F("BILATERAL_SIGMA_SPATIAL", p.bilateral_sigma_spatial);
    I("BILATERAL_KERNEL_SIZE", p.bilateral_kernel_size);
    I("START_FRAME", p.start_frame); I("MAX_ITER", p.max_iter);
    F("MAX_UPDATE_NORM", p.max_update_norm);
    I("S", p.s); F("LAMBDA", p.lambda); F("ALPHA", p.alpha); F("W_REG", p.w_reg);
    float trunc_vox = 0.f, eta_vox = 0.f, tz = 0.f;
    for (const char* k : {"TSDF_TRUNC_DIST", "ETA", "VOL_POSE_T_Z"})
        if (!kv.count(k)) {
            if (why) *why = std::string("required key ") + k + " is missing from " + path;
            return false;
        }
    F("TSDF_TRUNC_DIST", trunc_vox); F("ETA", eta_vox); F("VOL_POSE_T_Z", tz);
    for (int i = 0; i < 3; ++i)
        if (p.volume_dims[i] <= 0 || !(p.volume_size[i] > 0.f)) {
            if (why) *why = "VOL_DIMS_* / VOL_SIZE_* must be positive in " + path;
            return false;
        }
    if (!(trunc_vox > 0.f)) {
        if (why) *why = "TSDF_TRUNC_DIST must be positive in " + path;
        return false;
    }
    p.tsdf_trunc_dist = trunc_vox * p.voxel_sizes()[0];  // demo.cpp:71
    p.eta             = eta_vox * p.voxel_sizes()[0];    // demo.cpp:72
    p.volume_pose     = cv::Affine3f().translate(cv::Vec3f(-p.volume_size[0] / 2.f, -p.volume_size[1] / 2.f, tz));  // :73-74
    if (raw) *raw = kv;
    return true;
}
}  // namespace sobfu_amd

// SobFusion (include/sobfu/sob_fusion.hpp, src/sobfu/sob_fusion.cpp:71-145) -- per-frame driver: bilateral filter ->
// depth truncation -> dists; frame 0 builds phi_global and allocates everything; frame n builds phi_n, fuses it
// directly while n < START_FRAME, otherwise estimates psi (warm-started) and fuses phi_n o psi.
// A maintainer who keeps a SobFusion of their own (the reference's src/sobfu/sob_fusion.cpp with its PCL mesh getters) over the shells
// below defines SOBFU_AMD_NO_SOBFUSION before including this header: the class then stays theirs.
#ifndef SOBFU_AMD_NO_SOBFUSION
class SobFusion {
public:
    explicit SobFusion(const Params& p) : frame_counter_(0), params(p), camera_pose_(cv::Affine3f::Identity()) {
        dists_.create(params.rows, params.cols);
        mc = std::make_shared<kfusion::cuda::MarchingCubes>();  // sob_fusion.cpp:35-36
        mc->setPose(params.volume_pose);
    }
    Params& getParams() { return params; }
    bool operator()(const kfusion::cuda::Depth& depth) {
        std::printf("--- FRAME NO. %d ---\n", frame_counter_);
        kfusion::cuda::depthBilateralFilter(depth, filtered_, params.bilateral_kernel_size, params.bilateral_sigma_spatial,
                                            params.bilateral_sigma_depth);                              // sob_fusion.cpp:78
        kfusion::cuda::depthTruncation(filtered_, params.icp_truncate_depth_dist);                     // :85
        kfusion::cuda::computeDists(filtered_, dists_, params.intr);                                   // :91
        if (frame_counter_ == 0) {                                                                     // :93-123
            phi_global = cv::Ptr<kfusion::cuda::TsdfVolume>(new kfusion::cuda::TsdfVolume(params));
            phi_global->integrate(dists_, camera_pose_, params.intr);
            phi_global_psi_inv = cv::Ptr<kfusion::cuda::TsdfVolume>(new kfusion::cuda::TsdfVolume(params));
            phi_n              = cv::Ptr<kfusion::cuda::TsdfVolume>(new kfusion::cuda::TsdfVolume(params));
            phi_n_psi          = cv::Ptr<kfusion::cuda::TsdfVolume>(new kfusion::cuda::TsdfVolume(params));
            psi     = std::make_shared<sobfu::cuda::DeformationField>(params.volume_dims);
            psi_inv = std::make_shared<sobfu::cuda::DeformationField>(params.volume_dims);
            solver  = std::make_shared<sobfu::cuda::Solver>(params);
            return ++frame_counter_, true;
        }
        phi_n->clear();                                                                                // :129
        phi_n->integrate(dists_, camera_pose_, params.intr);                                           // :130
        if (frame_counter_ < params.start_frame) {                                                     // :136-139
            phi_global->integrate(*phi_n);
            return ++frame_counter_, true;
        }
        solver->estimate_psi(phi_global, phi_global_psi_inv, phi_n, phi_n_psi, psi, psi_inv);          // :141
        phi_global->integrate(*phi_n_psi);                                                             // :142
        return ++frame_counter_, true;
    }
    std::shared_ptr<sobfu::cuda::DeformationField> getDeformationField() { return psi; }
    // meshes of the four volumes (sob_fusion.cpp:147-183); a host triangle soup replaces pcl::PolygonMesh
    sobfu_amd::TriangleMesh get_phi_global_mesh() { return get_mesh(phi_global); }
    sobfu_amd::TriangleMesh get_phi_global_psi_inv_mesh() { return get_mesh(phi_global_psi_inv); }
    sobfu_amd::TriangleMesh get_phi_n_mesh() { return get_mesh(phi_n); }
    sobfu_amd::TriangleMesh get_phi_n_psi_mesh() { return get_mesh(phi_n_psi); }
    sobfu_amd::TriangleMesh get_mesh(cv::Ptr<kfusion::cuda::TsdfVolume> vol) {
        kfusion::cuda::DeviceArray<kfusion::cuda::Point> vertices_buffer;
        kfusion::cuda::DeviceArray<kfusion::cuda::Normal> normals_buffer;
        kfusion::cuda::Surface model = mc->run(*vol, vertices_buffer, normals_buffer);
        kfusion::cuda::waitAllDefaultStream();
        return convert_to_mesh(model.vertices);
    }
    static sobfu_amd::TriangleMesh convert_to_mesh(const kfusion::cuda::DeviceArray<kfusion::cuda::Point>& triangles) {
        sobfu_amd::TriangleMesh m;
        if (!triangles.empty()) triangles.download(m.vertices);
        return m;
    }
    std::shared_ptr<kfusion::cuda::MarchingCubes> mc;

    cv::Ptr<kfusion::cuda::TsdfVolume> phi_global, phi_global_psi_inv, phi_n, phi_n_psi;
    std::shared_ptr<sobfu::cuda::DeformationField> psi, psi_inv;
    std::shared_ptr<sobfu::cuda::Solver> solver;

private:
    int frame_counter_;
    Params params;
    cv::Affine3f camera_pose_;  // fixed to identity, sob_fusion.cpp:33
    kfusion::cuda::Depth filtered_;
    kfusion::cuda::Dists dists_;
};
#endif  // SOBFU_AMD_NO_SOBFUSION
