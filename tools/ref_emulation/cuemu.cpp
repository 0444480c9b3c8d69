// Block scheduler for the host emulation (see shim/cuda_runtime.h for what this is and is not).
//
// A launch spreads its blocks over host threads; on each of them blocks run one after another and the threads of a block are coroutines (a small hand-written switch on x86-64, ucontext elsewhere).  __syncthreads() and the warp
// exchanges switch back to the scheduler, which resumes every live thread once per phase in thread-id order and sets
// threadIdx before each resume.  A thread that returns early simply stops taking part (the reference's kernels return
// before barriers only for threads that no later phase depends on).  Warp-synchronous code without a barrier would NOT be
// emulated correctly; none of the translation units compiled by build.py contains any (kfusion's Block::reduce, which does,
// is used by proj_icp.cu only).
#include <omp.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <vector>

#include <cuda_runtime.h>  // last: it turns 'asm' into a macro

namespace cuemu {
thread_local uint3 tIdx, bIdx;
dim3 bDim, gDim;
int max_threads = [] {
    const char* e = getenv("CUEMU_THREADS");
    const int n   = e ? atoi(e) : omp_get_max_threads();
    return n > 0 ? n : 1;
}();

namespace {
// Context switch.  x86-64: a dozen instructions (callee-saved registers, MXCSR / x87 control word, the stack pointer) instead of
// swapcontext's two signal-mask system calls -- the convolution kernels alone make millions of switches per launch at 128^3 and up.
// Elsewhere: ucontext.
#if defined(__x86_64__)
extern "C" void cuemu_switch(void** save_sp, void* load_sp);
__asm__(R"(
    .text
    .globl cuemu_switch
    .type cuemu_switch, @function
cuemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size cuemu_switch, .-cuemu_switch
)");
struct Thread {
    void* sp;
    bool done;
};
thread_local void* sched_sp;
#else
struct Thread {
    ucontext_t ctx;
    bool done;
};
thread_local ucontext_t sched;
#endif
enum { kStack = 256 << 10 };
// per HOST thread: the coroutines of the block it is running
thread_local std::vector<Thread> threads;
thread_local std::vector<void*> stacks;
thread_local std::vector<unsigned long long> xbuf;  // one exchange slot per thread of the block
thread_local std::vector<char> smem;
thread_local int cur = -1;
body cur_body;      // per launch
int n_threads = 0;  // threads of a block

void to_scheduler() {
#if defined(__x86_64__)
    cuemu_switch(&threads[cur].sp, sched_sp);
#else
    swapcontext(&threads[cur].ctx, &sched);
#endif
}
void trampoline() {
    cur_body.fn(cur_body.closure);
    threads[cur].done = true;
    to_scheduler();
    abort();  // a finished thread is never resumed
}
void prepare(int t) {  // thread t will start in trampoline() on its own stack when first resumed
    threads[t].done = false;
#if defined(__x86_64__)
    // what cuemu_switch pops: [mxcsr | x87 cw] r15 r14 r13 r12 rbx rbp <return address>; after its `ret` the stack pointer must
    // be 8 modulo 16, as at any function entry
    uintptr_t top = ((uintptr_t) stacks[t] + kStack) & ~(uintptr_t) 15;
    void** sp     = (void**) (top - 8);  // where rsp points after the `ret`
    *--sp         = (void*) &trampoline;
    for (int k = 0; k < 6; ++k) *--sp = nullptr;
    --sp;
    unsigned int csr = 0x1F80;  // default MXCSR; default x87 control word
    unsigned short cw = 0x037F;
    memcpy((char*) sp, &csr, 4);
    memcpy((char*) sp + 4, &cw, 2);
    threads[t].sp = sp;
#else
    getcontext(&threads[t].ctx);
    threads[t].ctx.uc_stack.ss_sp   = stacks[t];
    threads[t].ctx.uc_stack.ss_size = kStack;
    threads[t].ctx.uc_link          = 0;
    makecontext(&threads[t].ctx, trampoline, 0);
#endif
}
void resume(int t) {
#if defined(__x86_64__)
    cuemu_switch(&sched_sp, threads[t].sp);
#else
    swapcontext(&sched, &threads[t].ctx);
#endif
}
void set_tid(int t) {
    cur    = t;
    tIdx.x = t % bDim.x;
    tIdx.y = (t / bDim.x) % bDim.y;
    tIdx.z = t / (bDim.x * bDim.y);
}
}  // namespace

// Blocks of a launch are independent by the programming model, so they are spread over host threads (OpenMP); the threads of a
// block stay coroutines of ONE host thread.  Every array the fixtures record is independent of the block order; the one kernel whose
// OUTPUT ORDER depends on it (marching cubes' atomic append) is run with max_threads = 1 by the driver.
static void run_block(const cfg& c, unsigned gx, unsigned gy, unsigned gz) {
    if ((int) threads.size() < n_threads) threads.resize(n_threads), xbuf.resize(n_threads);
    while ((int) stacks.size() < n_threads) {
        void* s = mmap(0, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (s == MAP_FAILED) perror("cuemu: mmap"), abort();
        stacks.push_back(s);
    }
    if (smem.size() < std::max<size_t>(c.smem, 64 << 10)) smem.resize(std::max<size_t>(c.smem, 64 << 10));
    bIdx = uint3{gx, gy, gz};
    for (int t = 0; t < n_threads; ++t) prepare(t);
    for (int live = n_threads; live;)
        for (int t = 0; t < n_threads; ++t)
            if (!threads[t].done) {
                set_tid(t);
                resume(t);
                if (threads[t].done) --live;
            }
    cur = -1;
}

void launch_impl(const cfg& c, body b) {
    if (cur >= 0) fprintf(stderr, "cuemu: nested launch\n"), abort();
    bDim = c.block, gDim = c.grid, cur_body = b;
    n_threads = (int) (c.block.x * c.block.y * c.block.z);
    const long nb = (long) c.grid.x * c.grid.y * c.grid.z;
    const int nt  = (int) std::min<long>(max_threads, nb);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) if (nt > 1)
    for (long i = 0; i < nb; ++i) run_block(c, (unsigned) (i % c.grid.x), (unsigned) ((i / c.grid.x) % c.grid.y), (unsigned) (i / ((long) c.grid.x * c.grid.y)));
}

void syncthreads() { to_scheduler(); }
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
