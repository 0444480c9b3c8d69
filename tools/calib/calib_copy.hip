// Counter calibration for gfx950: streaming copies of a KNOWN byte count with the access patterns the fused passes use --
// 12-byte (global_load/store_dwordx3) and 4-byte (dword) elements, plain and nontemporal -- next to the 16-byte pattern the
// MI355X guide calibrated (FETCH_SIZE reports 1/2 of a 16 B/lane streaming read).  Run under `rocprofv3 --pmc FETCH_SIZE` and
// `--pmc WRITE_SIZE` (separate passes, tools/calibrate_counters.sh); every pattern is its own kernel name, every launch reads
// N elements once and writes N elements once, so  factor = known bytes / (counter KiB * 1024).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v3f __attribute__((ext_vector_type(3)));
typedef v3f __attribute__((aligned(4))) v3f_u;
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x)                                                              \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) {                                            \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));     \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

__global__ void __launch_bounds__(256) copy_f4(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
__global__ void __launch_bounds__(256) copy_f4_nt(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
__global__ void __launch_bounds__(256) copy_x3(const float* __restrict__ s, float* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) *(v3f_u*) (d + 3 * i) = *(const v3f_u*) (s + 3 * i);
}
__global__ void __launch_bounds__(256) copy_x3_nt(const float* __restrict__ s, float* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load((const v3f_u*) (s + 3 * i)), (v3f_u*) (d + 3 * i));
}
__global__ void __launch_bounds__(256) copy_f1(const float* __restrict__ s, float* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
__global__ void __launch_bounds__(256) copy_f1_nt(const float* __restrict__ s, float* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
// read-only / write-only variants separate the two directions of the x3 pattern
__global__ void __launch_bounds__(256) read_x3(const float* __restrict__ s, float* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    v3f v = *(const v3f_u*) (s + 3 * i);
    if (v.x + v.y + v.z == 12345.678f) d[0] = 1.f;  // never true: keeps the load alive, no store traffic
}
__global__ void __launch_bounds__(256) write_x3(const float* __restrict__ s, float* __restrict__ d, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) *(v3f_u*) (d + 3 * i) = v3f{1.f, 2.f, 3.f};
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t) std::atoll(argv[1]) : ((size_t) 1 << 25);  // elements per array (2 x 256^3)
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    float *s = nullptr, *d = nullptr;
    CK(hipMalloc((void**) &s, n * 16));
    CK(hipMalloc((void**) &d, n * 16));
    CK(hipMemset(s, 0, n * 16));
    CK(hipMemset(d, 0, n * 16));
    const dim3 grid((unsigned) ((n + 255) / 256)), block(256);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(copy_f4, grid, block, 0, 0, (const v4f*) s, (v4f*) d, n);
        hipLaunchKernelGGL(copy_f4_nt, grid, block, 0, 0, (const v4f*) s, (v4f*) d, n);
        hipLaunchKernelGGL(copy_x3, grid, block, 0, 0, s, d, n);
        hipLaunchKernelGGL(copy_x3_nt, grid, block, 0, 0, s, d, n);
        hipLaunchKernelGGL(copy_f1, grid, block, 0, 0, s, d, n);
        hipLaunchKernelGGL(copy_f1_nt, grid, block, 0, 0, s, d, n);
        hipLaunchKernelGGL(read_x3, grid, block, 0, 0, s, d, n);
        hipLaunchKernelGGL(write_x3, grid, block, 0, 0, s, d, n);
    }
    CK(hipDeviceSynchronize());
    std::printf("calib_copy: n = %zu elements per array, %d launches per pattern; known bytes per launch and direction: "
                "f4 %zu, x3 %zu, f1 %zu\n", n, reps, n * 16, n * 12, n * 4);
    return 0;
}
