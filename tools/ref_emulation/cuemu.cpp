// Block scheduler for the host emulation (see shim/cuda_runtime.h for what this is and is not).
//
// A launch runs its blocks one after another; the threads of a block are ucontext coroutines.  __syncthreads() and the warp
// exchanges switch back to the scheduler, which resumes every live thread once per phase in thread-id order and sets
// threadIdx before each resume.  A thread that returns early simply stops taking part (the reference's kernels return
// before barriers only for threads that no later phase depends on).  Warp-synchronous code without a barrier would NOT be
// emulated correctly; none of the translation units compiled by build.py contains any (kfusion's Block::reduce, which does,
// is used by proj_icp.cu only).
#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

#include <cuda_runtime.h>  // last: it turns 'asm' into a macro

namespace cuemu {
uint3 tIdx, bIdx;
dim3 bDim, gDim;

namespace {
struct Thread {
    ucontext_t ctx;
    bool done;
};
enum { kStack = 256 << 10 };
ucontext_t sched;
std::vector<Thread> threads;
std::vector<void*> stacks;
std::vector<unsigned long long> xbuf;  // one exchange slot per thread
std::vector<char> smem(64 << 10);
body cur_body;
int cur = -1, n_threads = 0;

void trampoline() {
    cur_body.fn(cur_body.closure);
    threads[cur].done = true;
    swapcontext(&threads[cur].ctx, &sched);
}
void set_tid(int t) {
    cur    = t;
    tIdx.x = t % bDim.x;
    tIdx.y = (t / bDim.x) % bDim.y;
    tIdx.z = t / (bDim.x * bDim.y);
}
}  // namespace

void launch_impl(const cfg& c, body b) {
    if (cur >= 0) fprintf(stderr, "cuemu: nested launch\n"), abort();
    bDim = c.block, gDim = c.grid, cur_body = b;
    n_threads = (int) (c.block.x * c.block.y * c.block.z);
    if ((int) threads.size() < n_threads) threads.resize(n_threads), xbuf.resize(n_threads);
    while ((int) stacks.size() < n_threads) {
        void* s = mmap(0, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (s == MAP_FAILED) perror("cuemu: mmap"), abort();
        stacks.push_back(s);
    }
    if (smem.size() < c.smem) smem.resize(c.smem);
    for (unsigned gz = 0; gz < c.grid.z; ++gz)
        for (unsigned gy = 0; gy < c.grid.y; ++gy)
            for (unsigned gx = 0; gx < c.grid.x; ++gx) {
                bIdx = uint3{gx, gy, gz};
                for (int t = 0; t < n_threads; ++t) {
                    getcontext(&threads[t].ctx);
                    threads[t].ctx.uc_stack.ss_sp   = stacks[t];
                    threads[t].ctx.uc_stack.ss_size = kStack;
                    threads[t].ctx.uc_link          = 0;
                    threads[t].done                 = false;
                    makecontext(&threads[t].ctx, trampoline, 0);
                }
                for (int live = n_threads; live;)
                    for (int t = 0; t < n_threads; ++t)
                        if (!threads[t].done) {
                            set_tid(t);
                            swapcontext(&sched, &threads[t].ctx);
                            if (threads[t].done) --live;
                        }
            }
    cur = -1;
}

void syncthreads() { swapcontext(&threads[cur].ctx, &sched); }
void* dynamic_smem() { return smem.data(); }

// lane src of the caller's warp, as it was when that lane made the same call; the caller's own value when src is outside
// the warp or that lane is not running (CUDA leaves that case undefined; the reference's reductions never use the result)
unsigned long long exchange(unsigned long long v, int src_lane) {
    int me = cur;
    xbuf[me] = v;
    syncthreads();
    int src = (me & ~31) + src_lane;
    unsigned long long r = (src_lane >= 0 && src_lane < 32 && src < n_threads && !threads[src].done) ? xbuf[src] : v;
    syncthreads();
    return r;
}
unsigned int ballot(int predicate) {
    int me = cur;
    xbuf[me] = predicate ? 1u : 0u;
    syncthreads();
    unsigned int mask = 0;
    for (int l = 0, base = me & ~31; l < 32 && base + l < n_threads; ++l)
        if (!threads[base + l].done && xbuf[base + l]) mask |= 1u << l;
    syncthreads();
    return mask;
}
unsigned int ptx_special(const char* text) {
    unsigned int lane = (unsigned int) cur & 31u;
    if (strstr(text, "%laneid")) return lane;
    if (strstr(text, "%lanemask_lt")) return (1u << lane) - 1u;
    fprintf(stderr, "cuemu: unknown PTX special register in '%s'\n", text), abort();
}

void* dev_alloc(size_t bytes) {
    char* p = (char*) calloc(1, bytes + 2 * (size_t) kGuard);
    if (!p) perror("cuemu: calloc"), abort();
    return p + kGuard;
}
void dev_free(void* p) {
    if (p) free((char*) p - kGuard);
}
}  // namespace cuemu
