// CPU SIMT emulator runtime -- TEST INFRASTRUCTURE ONLY (see include/hip/hip_runtime.h).
// Fibers (one per GPU thread) with a hand-rolled x86-64 context switch; blocks of a grid are
// distributed over a small pool of OS threads.
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>

namespace pf_emu {

thread_local Block* tl_block = nullptr;

asm(R"(
.text
.globl pf_emu_switch
.type pf_emu_switch,@function
pf_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size pf_emu_switch,.-pf_emu_switch
)");

void yield() {
    Block* b = tl_block;
    Fiber* f = b->cur;
    pf_emu_switch(&f->sp, b->sched_sp);
}

static void fiber_entry() {
    Block* b = tl_block;
    Fiber* f = b->cur;
    (*b->body)();
    f->done = true;
    b->live--;
    b->waves[f->linear / kWave].live--;
    for (;;) pf_emu_switch(&f->sp, b->sched_sp);  // never resumed
}

static void prepare_fiber(Fiber& f) {
    uintptr_t top = (uintptr_t)(f.stack + kStackBytes);
    top &= ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // fake return address of fiber_entry
    *--sp = (void*)&fiber_entry;     // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12..r15
    f.sp = sp;
    f.done = false;
}

struct StackPool {
    std::vector<char*> stacks;
    ~StackPool() { for (char* s : stacks) std::free(s); }
    char* get(size_t i) {
        while (stacks.size() <= i) stacks.push_back((char*)std::aligned_alloc(64, kStackBytes));
        return stacks[i];
    }
};

static void run_block(Block& b, StackPool& pool) {
    const unsigned n = b.bdim.x * b.bdim.y * b.bdim.z;
    b.fibers.assign(n, Fiber());
    b.waves.assign((n + kWave - 1) / kWave, WaveState());
    b.live = (int)n;
    b.arrived = 0;
    b.gen = 0;
    for (unsigned i = 0; i < n; ++i) {
        Fiber& f = b.fibers[i];
        f.block = &b;
        f.linear = (int)i;
        f.tid.x = i % b.bdim.x;
        f.tid.y = (i / b.bdim.x) % b.bdim.y;
        f.tid.z = i / (b.bdim.x * b.bdim.y);
        f.stack = pool.get(i);
        prepare_fiber(f);
        b.waves[i / kWave].live++;
    }
    tl_block = &b;
    long guard = 0;
    while (b.live > 0) {
        for (unsigned i = 0; i < n; ++i) {
            Fiber& f = b.fibers[i];
            if (f.done) continue;
            b.cur = &f;
            pf_emu_switch(&b.sched_sp, f.sp);
        }
        if (++guard > 200000000L) { std::fprintf(stderr, "pf_emu: deadlock suspected\n"); std::abort(); }
    }
    tl_block = nullptr;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("PF_EMU_THREADS")) hw = (unsigned)std::atoi(e);
    if (hw < 1) hw = 1;
    size_t nworkers = std::min<size_t>(hw, nblocks);
    std::atomic<size_t> next{0};
    static std::mutex pool_mu;
    static std::vector<StackPool*> free_pools;
    auto worker = [&]() {
        StackPool* pool_p = nullptr;
        {
            std::lock_guard<std::mutex> g(pool_mu);
            if (!free_pools.empty()) { pool_p = free_pools.back(); free_pools.pop_back(); }
        }
        if (!pool_p) pool_p = new StackPool();
        StackPool& pool = *pool_p;
        Block b;
        b.bdim = block;
        b.gdim = grid;
        b.body = &body;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bid.x = (unsigned)(i % grid.x);
            b.bid.y = (unsigned)((i / grid.x) % grid.y);
            b.bid.z = (unsigned)(i / ((size_t)grid.x * grid.y));
            run_block(b, pool);
        }
        std::lock_guard<std::mutex> g(pool_mu);
        free_pools.push_back(pool_p);
    };
    if (nworkers == 1) { worker(); return; }
    std::vector<std::thread> ts;
    for (size_t t = 0; t < nworkers; ++t) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
}

}  // namespace pf_emu

double pf_emu_now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
