// What does a back-to-back kernel launch cost on this part?  Empty kernels of the tile loop's shapes (grid, block, kernel-argument
// size), timed as N launches on one stream.  hipcc --offload-arch=gfx950 -O3 tools/calib/launch_cost.hip -o build/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int v[340]; };  // ~1.3 KB, like TilePassAArgs
__global__ void k_empty(int x) { if (x == 12345678) asm volatile("s_nop 0"); }
__global__ void k_big(Big b) { if (b.v[0] == 12345678) asm volatile("s_nop 0"); }
__global__ void k_store(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1.f; }
template <class F> double time_us(F f, int n) {
    for (int i = 0; i < 50; ++i) f();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float* p; hipMalloc(&p, 64 << 20);
    Big b{};
    for (int grid : {1, 256, 438, 1024, 1400}) {
        printf("grid %4d x 512: empty %.2f us   1.3KB-args %.2f us   store 2M floats(8MB) %.2f us\n", grid,
               time_us([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), 0, st, 1); }, 2000),
               time_us([&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(512), 0, st, b); }, 2000),
               time_us([&] { hipLaunchKernelGGL(k_store, dim3(4096), dim3(512), 0, st, p, 2 << 20); }, 2000));
    }
    printf("two alternating kernels (empty, grid 438 / 1400): %.2f us per pair\n",
           time_us([&] { hipLaunchKernelGGL(k_big, dim3(1400), dim3(512), 0, st, b); hipLaunchKernelGGL(k_empty, dim3(438), dim3(512), 0, st, 1); }, 2000));
    return 0;
}
