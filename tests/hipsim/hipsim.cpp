// hipsim runtime — see hipsim.h.  TEST INFRASTRUCTURE ONLY.
#include "hipsim.h"

#include <ucontext.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace hipsim {

thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {

constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;

// ---- fiber schedule ------------------------------------------------------------------------
// On the GPU the waves of a workgroup run in no particular order between two barriers; the simulator's default (waves 0, 1, ...
// one after the other, lanes 0..63 inside each) is ONE of those orders, and a kernel with a missing barrier (wave 1 reading what
// wave 0 writes) passes under it by luck.  The schedule can therefore be perturbed: REVERSE walks waves and lanes backwards, RANDOM
// draws a new permutation of the waves and of each wave's lanes at every scheduling pass (seeded by the block index, reproducible).
// A kernel whose result is a function of its inputs alone gives the same bits under every schedule (tests/test_schedule_sim.py).
enum Sched { SCHED_FORWARD = 0, SCHED_REVERSE = 1, SCHED_RANDOM = 2 };
std::atomic<int> g_sched_mode{-1};
std::atomic<uint64_t> g_sched_seed{1};

int sched_mode() {
    int m = g_sched_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    const char* e = getenv("HIPSIM_SCHED");           // forward | reverse | random[:seed]
    m = SCHED_FORWARD;
    if (e && !strncmp(e, "reverse", 7)) m = SCHED_REVERSE;
    if (e && !strncmp(e, "random", 6)) {
        m = SCHED_RANDOM;
        if (e[6] == ':') g_sched_seed.store(strtoull(e + 7, nullptr, 10));
    }
    g_sched_mode.store(m);
    return m;
}

struct Rng {
    uint64_t s;
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1DULL; }
};

// fills order[0..n) with the visiting order of the indices 0..n-1 under the current schedule
void fill_order(int* order, int n, int mode, Rng& rng) {
    for (int i = 0; i < n; ++i) order[i] = mode == SCHED_REVERSE ? n - 1 - i : i;
    if (mode == SCHED_RANDOM)
        for (int i = n - 1; i > 0; --i) std::swap(order[i], order[(int)(rng.next() % (uint64_t)(i + 1))]);
}

enum State { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };

// ---- context switch --------------------------------------------------------------------------
// glibc's swapcontext saves and restores the signal mask with a system call on every switch - a third of the simulator's run time
// (a kernel thread yields at every shuffle, MFMA and barrier).  On x86-64 the switch is therefore a dozen instructions of our own:
// callee-saved registers, the SSE / x87 control words and the stack pointer.  HIPSIM_UCONTEXT (set by the sanitizer builds, whose
// runtime follows swapcontext) or any other architecture keeps the portable form.
#if defined(__x86_64__) && !defined(HIPSIM_UCONTEXT)
#define HIPSIM_ASM_SWITCH 1
struct Ctx { void* sp; };
extern "C" void hipsim_switch(Ctx* from, Ctx* to);
asm(R"(
    .text
    .globl hipsim_switch
    .type hipsim_switch,@function
hipsim_switch:
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
    movq (%rsi), %rsp
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
    .size hipsim_switch,.-hipsim_switch
)");
// a fresh context whose first switch-in "returns" into entry() on its own stack
inline void ctx_make(Ctx* c, char* stack, size_t bytes, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
    uint64_t* sp = (uint64_t*)top;
    *--sp = 0;                        // the return address entry() would return to (it never does): rsp = 16 n + 8 at its first instruction
    *--sp = (uint64_t)entry;          // popped by hipsim_switch's ret
    for (int i = 0; i < 6; ++i) *--sp = 0;      // rbp rbx r12 r13 r14 r15
    --sp;
    ((uint32_t*)sp)[0] = 0x1F80;      // MXCSR: all exceptions masked, round to nearest
    ((uint32_t*)sp)[1] = 0x037F;      // x87 control word: the same
    c->sp = sp;
}
#else
#define HIPSIM_ASM_SWITCH 0
struct Ctx { ucontext_t uc; };
inline void hipsim_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
inline void ctx_make(Ctx* c, char* stack, size_t bytes, void (*entry)()) {
    getcontext(&c->uc);
    c->uc.uc_stack.ss_sp = stack;
    c->uc.uc_stack.ss_size = bytes;
    c->uc.uc_link = nullptr;
    makecontext(&c->uc, entry, 0);
}
#endif

struct Fiber {
    Ctx ctx;
    int state;
    dim3 tid;
};

struct Worker {
    std::vector<Fiber> fibers;
    char* stacks = nullptr;          // kMaxThreads * kStackBytes, lazily committed
    Ctx sched;
    int cur = -1;
    int nthreads = 0;
    const std::function<void()>* body = nullptr;
    std::vector<Slot> slots;         // nwaves * 64
    std::vector<unsigned char> dyn;  // dynamic shared memory
    Worker() {
        fibers.resize(kMaxThreads);
        stacks = (char*)mmap(nullptr, (size_t)kMaxThreads * kStackBytes, PROT_READ | PROT_WRITE,
                             MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == MAP_FAILED) { perror("hipsim mmap"); abort(); }
        slots.resize((kMaxThreads / 64) * 64);
    }
    ~Worker() { munmap(stacks, (size_t)kMaxThreads * kStackBytes); }
};

thread_local Worker* t_worker = nullptr;

void trampoline() {
    Worker* w = t_worker;
    (*w->body)();
    Fiber& f = w->fibers[w->cur];
    f.state = DONE;
    hipsim_switch(&f.ctx, &w->sched);
    abort();                          // a finished fiber is never resumed
}

void yield_with(int st) {
    Worker* w = t_worker;
    Fiber& f = w->fibers[w->cur];
    f.state = st;
    hipsim_switch(&f.ctx, &w->sched);
}

void run_block(Worker* w, dim3 grid, dim3 block, dim3 bid, size_t dyn_bytes,
               const std::function<void()>& body) {
    const int n = (int)(block.x * block.y * block.z);
    if (n > kMaxThreads) { fprintf(stderr, "hipsim: block too large (%d)\n", n); abort(); }
    w->nthreads = n;
    w->body = &body;
    if (w->dyn.size() < dyn_bytes + 64) w->dyn.resize(dyn_bytes + 64);
    // A workgroup finds in its dynamic LDS whatever the previous workgroup on that CU left there.  The simulator hands every
    // block 0xFF bytes (NaN as fp32 / bf16, -1 as an integer), so a kernel that reads LDS it has not written fails loudly here
    // instead of differing from run to run on the GPU.  HIPSIM_POISON_LDS=0 switches it off.
    static const bool poison_lds = !(getenv("HIPSIM_POISON_LDS") && atoi(getenv("HIPSIM_POISON_LDS")) == 0);
    if (poison_lds && dyn_bytes) memset(w->dyn.data(), 0xFF, dyn_bytes + 64);
    t_blockIdx = bid; t_blockDim = block; t_gridDim = grid;
    for (int i = 0; i < n; ++i) {
        Fiber& f = w->fibers[i];
        f.state = RUNNABLE;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        ctx_make(&f.ctx, w->stacks + (size_t)i * kStackBytes, kStackBytes, trampoline);
    }
    const int nwaves = (n + 63) / 64;
    int live = n;
    const int mode = sched_mode();
    Rng rng{(g_sched_seed.load() * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)bid.x * 0x100000001B3ULL + bid.y * 7919u + bid.z * 104729u + 1)};
    rng.next();
    int wave_order[kMaxThreads / 64], lane_order[64];
    while (live > 0) {
        bool progressed = false;
        fill_order(wave_order, nwaves, mode, rng);
        for (int wi = 0; wi < nwaves; ++wi) {
            const int wv = wave_order[wi];
            const int lo = wv * 64, hi = std::min(n, lo + 64);
            for (;;) {
                bool ran = false;
                fill_order(lane_order, hi - lo, mode, rng);
                for (int li = 0; li < hi - lo; ++li) {
                    const int i = lo + lane_order[li];
                    Fiber& f = w->fibers[i];
                    if (f.state != RUNNABLE) continue;
                    w->cur = i;
                    t_threadIdx = f.tid;
                    hipsim_switch(&w->sched, &f.ctx);
                    ran = true;
                    if (f.state == DONE) --live;
                }
                progressed |= ran;
                int nl = 0, nw = 0;
                for (int i = lo; i < hi; ++i) {
                    if (w->fibers[i].state != DONE) ++nl;
                    if (w->fibers[i].state == WAIT_WAVE) ++nw;
                }
                if (nl > 0 && nw == nl) {
                    for (int i = lo; i < hi; ++i)
                        if (w->fibers[i].state == WAIT_WAVE) w->fibers[i].state = RUNNABLE;
                    progressed = true;
                    continue;
                }
                break;
            }
        }
        if (live > 0) {
            int nb = 0;
            for (int i = 0; i < n; ++i) if (w->fibers[i].state == WAIT_BLOCK) ++nb;
            if (nb == live) {
                for (int i = 0; i < n; ++i)
                    if (w->fibers[i].state == WAIT_BLOCK) w->fibers[i].state = RUNNABLE;
                progressed = true;
            }
        }
        if (!progressed) {
            fprintf(stderr, "hipsim: DEADLOCK in block (%u,%u,%u): divergent barrier? live=%d\n", bid.x, bid.y,
                    bid.z, live);
            for (int i = 0; i < n; ++i)
                if (w->fibers[i].state != DONE) fprintf(stderr, "  thread %d state %d\n", i, w->fibers[i].state);
            abort();
        }
    }
}

// ---- worker pool ---------------------------------------------------------------------------
struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    int nworkers = 0;
    // current job
    uint64_t job_id = 0;
    dim3 grid, block;
    size_t dyn = 0;
    const std::function<void()>* body = nullptr;
    std::atomic<long> next{0};
    long total = 0;
    int active = 0;
    bool stop = false;

    void worker_main() {
        Worker* w = new Worker();
        t_worker = w;
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || job_id != seen; });
                if (stop) break;
                seen = job_id;
            }
            for (;;) {
                long b = next.fetch_add(1);
                if (b >= total) break;
                dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y),
                         (unsigned)(b / ((long)grid.x * grid.y)));
                run_block(w, grid, block, bid, dyn, *body);
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
        delete w;
    }

    void ensure(int n) {
        if (nworkers == n) return;
        shutdown();
        stop = false;
        nworkers = n;
        for (int i = 0; i < n; ++i) threads.emplace_back([this] { worker_main(); });
    }
    void shutdown() {
        {
            std::unique_lock<std::mutex> lk(mu);
            stop = true;
            cv_work.notify_all();
        }
        for (auto& t : threads) t.join();
        threads.clear();
        nworkers = 0;
    }
    ~Pool() { shutdown(); }
};

Pool& pool() {
    static Pool* p = new Pool();  // leaked on purpose: threads may outlive static destruction order
    return *p;
}
int g_num_workers = 0;
std::mutex g_launch_mu;

}  // namespace

int num_workers() {
    if (g_num_workers <= 0) {
        const char* e = getenv("HIPSIM_WORKERS");
        int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        if (n <= 0) n = 1;
        if (n > 16) n = 16;
        g_num_workers = n;
    }
    return g_num_workers;
}
void set_num_workers(int n) { g_num_workers = n; }

void launch(dim3 grid, dim3 block, size_t dyn_bytes, const std::function<void()>& body) {
    std::lock_guard<std::mutex> g(g_launch_mu);
    Pool& p = pool();
    p.ensure(num_workers());
    {
        std::unique_lock<std::mutex> lk(p.mu);
        p.grid = grid; p.block = block; p.dyn = dyn_bytes; p.body = &body;
        p.total = (long)grid.x * grid.y * grid.z;
        p.next.store(0);
        p.active = p.nworkers;
        ++p.job_id;
        p.cv_work.notify_all();
        p.cv_done.wait(lk, [&] { return p.active == 0; });
    }
}

void set_schedule(int mode, uint64_t seed) {
    g_sched_seed.store(seed ? seed : 1);
    g_sched_mode.store(mode < 0 || mode > SCHED_RANDOM ? SCHED_FORWARD : mode);
}
int schedule() { return sched_mode(); }

void sync_block() { yield_with(WAIT_BLOCK); }
void sync_wave() { yield_with(WAIT_WAVE); }
int lane_id() { return t_worker->cur & 63; }
int wave_live_lanes() {
    Worker* w = t_worker;
    int lo = (w->cur / 64) * 64, hi = std::min(w->nthreads, lo + 64), n = 0;
    for (int i = lo; i < hi; ++i) if (w->fibers[i].state != DONE) ++n;
    return n;
}
Slot* wave_slots() { return t_worker->slots.data() + (t_worker->cur / 64) * 64; }
void* dyn_smem() {
    uintptr_t p = (uintptr_t)t_worker->dyn.data();
    return (void*)((p + 63) & ~(uintptr_t)63);
}

}  // namespace hipsim

// ---- HIP host API stubs --------------------------------------------------------------------
struct hipsimStream { int dummy; };
struct hipsimEvent { std::chrono::steady_clock::time_point t; };
struct hipsimGraph { int dummy; };
struct hipsimGraphExec { int dummy; };

hipError_t hipMalloc(void** p, size_t n) {
    void* q = nullptr;
    if (posix_memalign(&q, 256, n ? ((n + 255) / 256) * 256 : 256) != 0) return hipErrorOutOfMemory;
    memset(q, 0xCD, n);  // poison: catches reads of uninitialised device memory
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipsimStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipsim error"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipsimEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorStreamCaptureUnsupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorStreamCaptureUnsupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }

// C entry points for the tests (ctypes): 0 forward, 1 reverse, 2 random (seed)
extern "C" void hipsim_set_schedule(int mode, unsigned long long seed) { hipsim::set_schedule(mode, seed); }
extern "C" int hipsim_get_schedule(void) { return hipsim::schedule(); }
