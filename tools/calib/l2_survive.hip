// Do lines a kernel WROTE (or read) survive in the XCD's L2 until the next kernel on the same stream reads them?
// Workgroup b runs on XCD b % 8 (observed).  W: workgroup b writes slice b (64 KiB).  R0: workgroup b reads slice b (same XCD as the
// writer); R1: workgroup b reads slice b + 1 (another XCD).  Total 16 MiB = 2 MiB per XCD (the L2 holds 4 MiB).  Run under
// `rocprofv3 --pmc FETCH_SIZE --kernel-trace`: FETCH_SIZE of R0 ~ 0 means the written lines were still valid in the writer's L2.
//   hipcc --offload-arch=gfx950 -O3 tools/calib/l2_survive.hip -o build/l2_survive && build/l2_survive
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kSlice = 64 * 1024 / 16;  // float4 per slice
__global__ void __launch_bounds__(256) write_k(float4* a, float v) {
    float4* s = a + (size_t) blockIdx.x * kSlice;
    for (int i = threadIdx.x; i < kSlice; i += 256) s[i] = make_float4(v, v + 1.f, v + 2.f, (float) i);
}
template <int SHIFT>
__global__ void __launch_bounds__(256) read_k(const float4* a, float* out, int nb) {
    const float4* s = a + (size_t) ((blockIdx.x + SHIFT) % nb) * kSlice;
    float acc = 0.f;
    for (int i = threadIdx.x; i < kSlice; i += 256) { const float4 v = s[i]; acc += v.x + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    const int nb = 256;
    float4* a; float* out;
    hipMalloc(&a, (size_t) nb * kSlice * 16); hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(write_k, dim3(nb), dim3(256), 0, 0, a, (float) rep);
        hipLaunchKernelGGL(read_k<0>, dim3(nb), dim3(256), 0, 0, a, out, nb);
        hipLaunchKernelGGL(write_k, dim3(nb), dim3(256), 0, 0, a, (float) rep + 0.5f);
        hipLaunchKernelGGL(read_k<1>, dim3(nb), dim3(256), 0, 0, a, out, nb);
        hipLaunchKernelGGL(read_k<1>, dim3(nb), dim3(256), 0, 0, a, out, nb);  // a second time: lines READ by the same XCD a kernel ago
    }
    hipDeviceSynchronize();
    float ms[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        hipEventRecord(e0);
        for (int rep = 0; rep < 200; ++rep) {
            if (k == 0) hipLaunchKernelGGL(read_k<0>, dim3(nb), dim3(256), 0, 0, a, out, nb);
            else if (k == 1) { hipLaunchKernelGGL(read_k<0>, dim3(nb), dim3(256), 0, 0, a, out, nb); hipLaunchKernelGGL(read_k<1>, dim3(nb), dim3(256), 0, 0, a, out, nb); }
            else { hipLaunchKernelGGL(write_k, dim3(nb), dim3(256), 0, 0, a, 1.f); hipLaunchKernelGGL(read_k<0>, dim3(nb), dim3(256), 0, 0, a, out, nb); }
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[k], e0, e1);
    }
    printf("us per launch: same-slice re-read %.2f; alternating slices (pair) %.2f; write + read same slice (pair) %.2f\n", 1e3 * ms[0] / 200, 1e3 * ms[1] / 200, 1e3 * ms[2] / 200);
    return 0;
}
