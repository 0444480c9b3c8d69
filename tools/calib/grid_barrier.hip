// What would ONE LAUNCH PER ITERATION buy a multi-GPU tile?  (lever (a): a persistent kernel -- pass A, a
// device-wide barrier, pass B -- instead of two launches.)  A persistent kernel replaces a kernel boundary by a grid barrier that
// must do by hand what the boundary does for free: make every workgroup's stores visible to every other XCD (the eight L2s are not
// coherent with each other: write-back before, invalidate behind).  This program measures that barrier on the machine at hand, for
// the tile loop's launch shape (512 resident workgroups of 512 lanes, 2 per CU), against back-to-back launches of the same work:
//
//     phase work:  every lane stores one 12-byte cell and, behind the barrier / kernel boundary, reads a cell another XCD's
//                  workgroup stored (checked: a stale read is counted)
//     variants:    flat (one counter), two-level (32 counters of 16 arrivals, then one), each with and without the agent-scope
//                  release / acquire fences (buffer_wbl2 + buffer_inv) -- without them the reads are stale, the time is the floor
//
// hipcc --offload-arch=gfx950 -O3 tools/calib/grid_barrier.hip -o build/grid_barrier && build/grid_barrier
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Bar {
    uint32_t top;       // arrivals at the top level (monotone)
    uint32_t gen;       // generation: the number of barriers completed
    uint32_t err;       // a waiter gave up
    uint32_t pad[13];
    uint32_t grp[64 * 16];  // one counter per group, 64 bytes apart
};

template <bool TWO_LEVEL, bool FENCE>
__device__ __forceinline__ void grid_barrier(Bar* b, uint32_t it, uint32_t n_wg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        bool last;
        if (TWO_LEVEL) {
            const uint32_t g = blockIdx.x / 16u, gsz = min(16u, n_wg - g * 16u), ngrp = (n_wg + 15u) / 16u;
            const uint32_t k = __hip_atomic_fetch_add(&b->grp[g * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = false;
            if (k == it * gsz + gsz - 1u) {
                const uint32_t t = __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = t == it * ngrp + ngrp - 1u;
            }
        } else {
            const uint32_t t = __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == it * n_wg + n_wg - 1u;
        }
        if (last) __hip_atomic_store(&b->gen, it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            uint32_t spins = 0;
            while ((int32_t) (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (it + 1u)) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {  // never hang the GPU
                    __hip_atomic_store(&b->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// one phase of work: lane (wg, l) stores value(it) into its cell; then reads the cell of workgroup (wg + 1) % n -- which the
// dispatcher put on the next XCD -- as it was stored in the PREVIOUS phase
__device__ __forceinline__ void phase(float* cells, uint32_t it, uint32_t n_wg, uint32_t* stale) {
    const uint32_t me = blockIdx.x * blockDim.x + threadIdx.x, other = ((blockIdx.x + 1u) % n_wg) * blockDim.x + threadIdx.x;
    const size_t half = (size_t) n_wg * blockDim.x * 3;  // two halves of a ping-pong: a phase never overwrites what its neighbour still reads
    if (it > 0) {
        const float v = cells[((it - 1u) & 1u) * half + 3 * other];
        if (v != (float) it) atomicAdd(stale, 1u);
    }
    float* w = cells + (it & 1u) * half + 3 * me;
    w[0] = (float) (it + 1u);
    w[1] = 0.f;
    w[2] = 0.f;
}

template <bool TWO_LEVEL, bool FENCE>
__global__ void __launch_bounds__(512) persistent(float* cells, Bar* b, uint32_t iters, uint32_t base, uint32_t* stale) {
    for (uint32_t it = 0; it < iters; ++it) {
        phase(cells, it, gridDim.x, stale);
        grid_barrier<TWO_LEVEL, FENCE>(b, base + it, gridDim.x);
        // second barrier of an iteration (pass B -> next pass A) costs the same: one barrier per phase is what is timed
    }
}
__global__ void __launch_bounds__(512) one_phase(float* cells, uint32_t it, uint32_t* stale) { phase(cells, it, gridDim.x, stale); }
__global__ void __launch_bounds__(512) barrier_only(Bar* b, uint32_t iters, uint32_t base) {
    for (uint32_t it = 0; it < iters; ++it) grid_barrier<true, false>(b, base + it, gridDim.x);
}

int main() {
    const uint32_t n_wg = 512, iters = 400;
    float* cells;
    Bar* bar;
    uint32_t* stale;
    hipMalloc(&cells, (size_t) 2 * n_wg * 512 * 12);
    hipMalloc(&bar, sizeof(Bar));
    hipMalloc(&stale, 4);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto run = [&](const char* name, auto launch) {
        hipMemset(bar, 0, sizeof(Bar));
        hipMemset(stale, 0, 4);
        hipMemset(cells, 0, (size_t) 2 * n_wg * 512 * 12);
        hipDeviceSynchronize();
        launch(0u);  // warm-up (also leaves the counters at `iters` arrivals: `base` continues from there)
        hipStreamSynchronize(st);
        hipMemset(stale, 0, 4);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        launch(iters);
        hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        uint32_t s = 0, e = 0;
        hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&e, &bar->err, 4, hipMemcpyDeviceToHost);
        std::printf("%-64s %7.2f us per phase   stale reads %u%s\n", name, us / iters, s, e ? "   (A WAITER GAVE UP)" : "");
    };
    // NB: the warm-up call starts every cell at phase numbering 0 again, so the first read of the timed call compares against the warm-up's
    // last value: phases are numbered per call and the check skips it == 0
    run("persistent, flat barrier, no fences (floor; stale reads expected)", [&](uint32_t base) { hipLaunchKernelGGL((persistent<false, false>), dim3(n_wg), dim3(512), 0, st, cells, bar, iters, base, stale); });
    run("persistent, flat barrier, release + acquire at agent scope", [&](uint32_t base) { hipLaunchKernelGGL((persistent<false, true>), dim3(n_wg), dim3(512), 0, st, cells, bar, iters, base, stale); });
    run("persistent, two-level barrier, no fences (floor)", [&](uint32_t base) { hipLaunchKernelGGL((persistent<true, false>), dim3(n_wg), dim3(512), 0, st, cells, bar, iters, base, stale); });
    run("persistent, two-level barrier, release + acquire at agent scope", [&](uint32_t base) { hipLaunchKernelGGL((persistent<true, true>), dim3(n_wg), dim3(512), 0, st, cells, bar, iters, base, stale); });
    run("two-level barrier alone (no work, no fences)", [&](uint32_t base) { hipLaunchKernelGGL(barrier_only, dim3(n_wg), dim3(512), 0, st, bar, iters, base); });
    run("one launch per phase (the kernel boundary does the rest)", [&](uint32_t) { for (uint32_t it = 0; it < iters; ++it) hipLaunchKernelGGL(one_phase, dim3(n_wg), dim3(512), 0, st, cells, it, stale); });
    return 0;
}
