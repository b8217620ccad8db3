// TEST INFRASTRUCTURE -- run time of the CPU device model declared in tests/emu/hip/hip_runtime.h: fibres, the wavefront
// collectives with the gfx950 lane layouts, the workgroup barrier, the OS-thread pool that runs the workgroups of a launch.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>
#include <x86intrin.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

namespace emu {

thread_local Fiber* g_cur = nullptr;
thread_local unsigned g_poll = 0;

namespace {

enum { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
constexpr size_t STACK_BYTES = 512 * 1024;
constexpr int MAX_THREADS_PER_BLOCK = 1024;

struct Job {
    dim3 grid, block;
    size_t dyn;
    void (*thunk)(void*);
    void* ctx;
    const char* name;
    std::atomic<long long> next{0};
    long long nblocks = 0;
    std::atomic<int> failed{0};
};

struct Worker {
    void* sched_sp = nullptr;
    char* stacks = nullptr;
    Fiber fibers[MAX_THREADS_PER_BLOCK];
    Block blk;
    int runnable[MAX_WAVES];        // runnable lanes per wavefront
    int cursor[MAX_WAVES];
    std::vector<char> dyn;
    Job* job = nullptr;
    uint64_t rng = 0x9e3779b97f4a7c15ULL;
    bool spun = false;              // the fibre that just came back was polling (os_yield), not parked
};
thread_local Worker* t_worker = nullptr;
std::vector<Worker*> g_workers;     // for the SIGUSR1 state dump

int g_sched_mode = 0;               // S2AG_EMU_SCHED: 0 round robin at collective granularity, 1 / 2 a wavefront runs until it
unsigned g_sched_seed = 1;          // blocks (ascending / descending order), 3 seeded random wavefront choice
std::atomic<unsigned long long> g_launches{0}, g_collectives{0};

inline void set_state(Worker* w, Fiber* f, int s) {
    if (f->state == RUNNABLE) --w->runnable[f->wave_id];
    f->state = s;
    if (s == RUNNABLE) ++w->runnable[f->wave_id];
}

void complete_wave(Worker* w, Wave* wv) {
    wv->fn(wv);
    for (int i = 0; i < wv->nlanes; ++i) {
        Fiber* f = wv->lanes[i];
        if (f->state == WAIT_WAVE) set_state(w, f, RUNNABLE);
        f->rec = nullptr;
    }
    wv->arrived = 0;
    ++wv->ncoll;
}

void release_block(Worker* w) {
    Block& b = w->blk;
    for (int i = 0; i < b.nthreads; ++i)
        if (b.fibers[i].state == WAIT_BLOCK) set_state(w, &b.fibers[i], RUNNABLE);
    b.arrived = 0;
}

void fiber_entry() {
    Worker* w = t_worker;
    Fiber* f = g_cur;
    w->job->thunk(w->job->ctx);
    // exit: an exited thread takes part in nothing any more
    Wave* wv = f->wave;
    set_state(w, f, DONE);
    --wv->alive;
    --w->blk.alive;
    if (wv->arrived > 0 && wv->arrived == wv->alive) complete_wave(w, wv);
    if (w->blk.arrived > 0 && w->blk.arrived == w->blk.alive) release_block(w);
    emu_switch(&f->sp, w->sched_sp);
    abort();
}

void describe_deadlock(Worker* w) {
    Block& b = w->blk;
    fprintf(stderr, "[s2ag emu] DEADLOCK in kernel %s, workgroup (%u,%u,%u): no thread can run\n", w->job->name, b.bid.x, b.bid.y,
            b.bid.z);
    for (int v = 0; v < b.nwaves; ++v) {
        int cnt[4] = {0, 0, 0, 0};
        for (int i = 0; i < b.waves[v].nlanes; ++i) ++cnt[b.waves[v].lanes[i]->state];
        fprintf(stderr, "  wavefront %d: %d at a collective (op %d), %d at the workgroup barrier, %d exited\n", v, cnt[WAIT_WAVE],
                b.waves[v].op, cnt[WAIT_BLOCK], cnt[DONE]);
    }
}

void run_block(Worker* w, long long lin) {
    Job* job = w->job;
    Block& b = w->blk;
    const dim3 g = job->grid, bd = job->block;
    b.gdim = {g.x, g.y, g.z};
    b.bdim = {bd.x, bd.y, bd.z};
    b.bid.x = (unsigned)(lin % g.x);
    b.bid.y = (unsigned)((lin / g.x) % g.y);
    b.bid.z = (unsigned)(lin / ((long long)g.x * g.y));
    b.nthreads = (int)(bd.x * bd.y * bd.z);
    b.nwaves = (b.nthreads + WAVE - 1) / WAVE;
    b.alive = b.nthreads;
    b.arrived = 0;
    b.fibers = w->fibers;
    if (w->dyn.size() < job->dyn + 64) w->dyn.resize(job->dyn + 64);
    b.dyn_lds = (char*)(((uintptr_t)w->dyn.data() + 63) & ~(uintptr_t)63);
    b.dyn_bytes = job->dyn;
    if (job->dyn) memset(b.dyn_lds, 0xff, job->dyn);       // LDS is not zero on entry: 0xff.. is a NaN in every float format
    for (int v = 0; v < b.nwaves; ++v) {
        Wave& wv = b.waves[v];
        wv.nlanes = std::min(WAVE, b.nthreads - v * WAVE);
        wv.alive = wv.nlanes;
        wv.arrived = 0;
        wv.op = 0;
        wv.fn = nullptr;
        wv.ncoll = 0;
        w->runnable[v] = 0;
        w->cursor[v] = 0;
    }
    for (int t = 0; t < b.nthreads; ++t) {
        Fiber& f = w->fibers[t];
        f.stack = w->stacks + (size_t)t * STACK_BYTES;
        f.tid.x = (unsigned)(t % bd.x);
        f.tid.y = (unsigned)((t / bd.x) % bd.y);
        f.tid.z = (unsigned)(t / (bd.x * bd.y));
        f.lin = t;
        f.lane = t % WAVE;
        f.wave_id = t / WAVE;
        f.wave = &b.waves[f.wave_id];
        f.blk = &b;
        f.rec = nullptr;
        f.state = DONE;                      // so that set_state counts it in
        f.wave->lanes[f.lane] = &f;
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void** frame = (void**)(top - 16 - 6 * sizeof(void*));
        for (int i = 0; i < 6; ++i) frame[i] = nullptr;
        frame[6] = (void*)&fiber_entry;
        f.sp = frame;
        set_state(w, &f, RUNNABLE);
    }
    int cur_wave = (g_sched_mode == 2) ? b.nwaves - 1 : 0;
    int spin_streak = 0;
    g_poll = 0;
    while (b.alive > 0) {
        // choose a wavefront with a runnable lane
        int wv = -1;
        if (g_sched_mode == 0) {
            for (int k = 0; k < b.nwaves; ++k) {                     // round robin, starting behind the one that just ran
                const int c = (cur_wave + k) % b.nwaves;
                if (w->runnable[c] > 0) { wv = c; break; }
            }
        } else if (w->runnable[cur_wave] > 0) {
            wv = cur_wave;
        } else if (g_sched_mode == 3) {
            int cand[MAX_WAVES], n = 0;
            for (int c = 0; c < b.nwaves; ++c) if (w->runnable[c] > 0) cand[n++] = c;
            if (n) {
                w->rng ^= w->rng << 13; w->rng ^= w->rng >> 7; w->rng ^= w->rng << 17;
                wv = cand[w->rng % (unsigned)n];
            }
        } else {
            for (int k = 1; k <= b.nwaves; ++k) {
                const int c = g_sched_mode == 2 ? (cur_wave - k + 2 * b.nwaves) % b.nwaves : (cur_wave + k) % b.nwaves;
                if (w->runnable[c] > 0) { wv = c; break; }
            }
        }
        if (wv < 0) {
            describe_deadlock(w);
            job->failed.store(1);
            return;                                                   // abandon the workgroup (its fibres are simply dropped)
        }
        Wave& W = b.waves[wv];
        int l = w->cursor[wv];
        bool wrapped = false;                                         // this wavefront's lanes have all had their turn
        while (W.lanes[l]->state != RUNNABLE) {
            if (++l == W.nlanes) { l = 0; wrapped = true; }
        }
        Fiber* f = W.lanes[l];
        if (l + 1 == W.nlanes) { w->cursor[wv] = 0; wrapped = true; } else w->cursor[wv] = l + 1;
        g_cur = f;
        w->spun = false;
        emu_switch(&w->sched_sp, f->sp);
        g_cur = nullptr;
        if (w->spun) {                                                // a polling thread: everybody else first
            int total = 0;
            for (int c = 0; c < b.nwaves; ++c) total += w->runnable[c];
            if (++spin_streak >= total) {
                sched_yield();
                spin_streak = 0;
            }
            if (wrapped) {
                // every lane of this wavefront has polled once: hand over to the NEXT wavefront that can run, in the mode's own
                // direction (mode 2 searches downwards: stepping up by one and searching down came straight back here and
                // starved every other wavefront -- a cooperative-GRU gather under the reversed schedule never ended)
                const int dir = g_sched_mode == 2 ? -1 : 1;
                int nxt = wv;
                for (int k = 1; k < b.nwaves; ++k) {
                    const int c = ((wv + dir * k) % b.nwaves + b.nwaves) % b.nwaves;
                    if (w->runnable[c] > 0) { nxt = c; break; }
                }
                cur_wave = nxt;
            } else {
                cur_wave = wv;
            }
            continue;
        }
        spin_streak = 0;
        // mode 0: after a lane parks move on to the next lane of the same wavefront while it has one, else the next wavefront
        if (g_sched_mode == 0) cur_wave = (w->runnable[wv] > 0 && !wrapped) ? wv : (wv + 1) % b.nwaves;
        else cur_wave = wv;
    }
    unsigned long long n = 0;
    for (int v = 0; v < b.nwaves; ++v) n += b.waves[v].ncoll;
    g_collectives.fetch_add(n, std::memory_order_relaxed);
}

// ---- pool -------------------------------------------------------------------------------------------------------------
struct Pool {
    std::mutex mu, mu2;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    Job* job = nullptr;
    unsigned long long epoch = 0;
    int want = 0;                 // workers that should take part in the current job
    int running = 0;
    bool stop = false;
};
Pool* g_pool = nullptr;
std::once_flag g_once;

void worker_main(int index) {
    Worker* w = new Worker();
    w->stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS_PER_BLOCK, PROT_READ | PROT_WRITE,
                            MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (w->stacks == MAP_FAILED) { perror("[s2ag emu] mmap"); abort(); }
    w->rng += (uint64_t)index * 0x632be5abULL + g_sched_seed;
    t_worker = w;
    {
        std::lock_guard<std::mutex> lk(g_pool->mu2);
        g_workers.push_back(w);
    }
    unsigned long long seen = 0;
    Pool* p = g_pool;
    for (;;) {
        Job* job;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_work.wait(lk, [&] { return p->stop || (p->epoch != seen && index < p->want); });
            if (p->stop) break;
            seen = p->epoch;
            job = p->job;
        }
        w->job = job;
        for (;;) {
            const long long lin = job->next.fetch_add(1);
            if (lin >= job->nblocks) break;
            run_block(w, lin);
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (--p->running == 0) p->cv_done.notify_all();
        }
    }
}

void dump_state(int) {
    for (Worker* w : g_workers) {
        if (!w->job || w->blk.alive <= 0) continue;
        Block& b = w->blk;
        fprintf(stderr, "[s2ag emu] %s workgroup (%u,%u,%u): %d threads alive, barrier arrivals %d\n", w->job->name, b.bid.x, b.bid.y,
                b.bid.z, b.alive, b.arrived);
        for (int v = 0; v < b.nwaves; ++v) {
            int cnt[4] = {0, 0, 0, 0};
            for (int i = 0; i < b.waves[v].nlanes; ++i) ++cnt[b.waves[v].lanes[i]->state];
            fprintf(stderr, "    wavefront %d: runnable %d, at collective %d (op %d), at barrier %d, exited %d; %llu collectives done\n", v,
                    cnt[RUNNABLE], cnt[WAIT_WAVE], b.waves[v].op, cnt[WAIT_BLOCK], cnt[DONE], b.waves[v].ncoll);
        }
    }
}

void init_once() {
    g_pool = new Pool();
    if (getenv("S2AG_EMU_TRACE")) signal(SIGUSR1, dump_state);
    if (const char* s = getenv("S2AG_EMU_SCHED")) g_sched_mode = atoi(s);
    if (const char* s = getenv("S2AG_EMU_SEED")) g_sched_seed = (unsigned)atoi(s);
}

}  // namespace

void yield_to_scheduler() {
    Fiber* f = g_cur;
    emu_switch(&f->sp, t_worker->sched_sp);
}

void collective(int op, CollectiveFn fn, void* rec) {
    Fiber* f = g_cur;
    Wave* wv = f->wave;
    if (wv->arrived == 0) {
        wv->op = op;
        wv->fn = fn;
    } else if (wv->op != op || wv->fn != fn) {
        fprintf(stderr, "[s2ag emu] kernel %s: wavefront %d of workgroup (%u,%u,%u) diverged: lane %d is at collective %d while others "
                        "wait at %d\n", t_worker->job->name, f->wave_id, f->blk->bid.x, f->blk->bid.y, f->blk->bid.z, f->lane, op, wv->op);
        abort();
    }
    f->rec = rec;
    ++wv->arrived;
    if (wv->arrived == wv->alive) {
        complete_wave(t_worker, wv);
        // every segment between two collectives runs the lanes of a wavefront in ASCENDING order (whatever the scheduling
        // mode): same-address LDS atomics issued by one instruction are resolved in a fixed order on the hardware too
        t_worker->cursor[f->wave_id] = 0;
        if (f->lane != 0) yield_to_scheduler();          // (still runnable: taken up again in its turn)
        return;
    }
    set_state(t_worker, f, WAIT_WAVE);
    yield_to_scheduler();
}

void block_barrier() {
    Worker* w = t_worker;
    Fiber* f = g_cur;
    Block& b = w->blk;
    ++b.arrived;
    if (b.arrived == b.alive) {
        release_block(w);
        for (int v = 0; v < b.nwaves; ++v) w->cursor[v] = 0;     // lanes in ascending order behind a barrier as well
        if (f->lane != 0) yield_to_scheduler();
        return;
    }
    set_state(w, f, WAIT_BLOCK);
    yield_to_scheduler();
}

// A thread that polls memory another workgroup -- or another thread of its OWN workgroup -- is going to write must let
// both run: hand the OS thread to the next fibre (still runnable); the scheduler yields the OS thread once every runnable
// fibre of the workgroup has come back from a poll without anything else happening.
void os_yield() {
    Fiber* f = g_cur;
    if (!f) { sched_yield(); return; }
    t_worker->spun = true;
    yield_to_scheduler();
}

void fail(const char* what) {
    fprintf(stderr, "[s2ag emu] %s\n", what);
    abort();
}

unsigned long long ticks() { return __rdtsc(); }

// Dynamic LDS beyond 64 KB is only granted to a kernel whose MaxDynamicSharedMemorySize attribute was raised (per
// instantiation): a launch that forgets it fails on the hardware with hipErrorInvalidValue -- and nowhere else.
namespace {
std::mutex g_attr_mu;
std::vector<std::pair<const void*, int>> g_attr;
}
void set_max_dynamic_lds(const void* kernel, int bytes) {
    std::lock_guard<std::mutex> lk(g_attr_mu);
    for (auto& e : g_attr)
        if (e.first == kernel) { e.second = bytes; return; }
    g_attr.emplace_back(kernel, bytes);
}
void check_dynamic_lds(const void* kernel, size_t dyn, const char* name) {
    if (dyn <= 64 * 1024) return;
    int granted = 64 * 1024;
    {
        std::lock_guard<std::mutex> lk(g_attr_mu);
        for (auto& e : g_attr)
            if (e.first == kernel) granted = e.second;
    }
    if ((size_t)granted < dyn) {
        fprintf(stderr, "[s2ag emu] kernel %s launched with %zu bytes of dynamic LDS but its MaxDynamicSharedMemorySize is %d\n", name,
                dyn, granted);
        abort();
    }
}

void launch_impl(dim3 grid, dim3 block, size_t dyn, void (*thunk)(void*), void* ctx, const char* name) {
    std::call_once(g_once, init_once);
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nblocks <= 0 || nthreads <= 0) return;
    if (nthreads > MAX_THREADS_PER_BLOCK) fail("workgroup larger than 1024 threads");
    if (dyn > 160 * 1024) fail("more than 160 KB of dynamic LDS");
    if (grid.y > 65535u || grid.z > 65535u) fail("gridDim.y / gridDim.z beyond 65535");
    Job job;
    job.grid = grid;
    job.block = block;
    job.dyn = dyn;
    job.thunk = thunk;
    job.ctx = ctx;
    job.name = name;
    job.nblocks = nblocks;
    // Grids a chip would hold at once are fully co-resident here too (one OS thread per workgroup), so that kernels whose
    // workgroups wait for each other make progress; bigger grids run on one thread per core.
    static const int ncores = std::max(1, (int)sysconf(_SC_NPROCESSORS_ONLN));
    static const int max_res = getenv("S2AG_EMU_RESIDENT") ? atoi(getenv("S2AG_EMU_RESIDENT")) : 320;
    int want = nblocks <= max_res ? (int)nblocks : ncores;
    if (getenv("S2AG_EMU_THREADS")) want = std::min<long long>(nblocks, atoi(getenv("S2AG_EMU_THREADS")));
    static const bool trace = getenv("S2AG_EMU_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[s2ag emu] launch %s grid (%u,%u,%u) block (%u,%u,%u) lds %zu on %d threads\n", name, grid.x, grid.y, grid.z,
                block.x, block.y, block.z, dyn, want);
    Pool* p = g_pool;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        while ((int)p->threads.size() < want) {
            const int idx = (int)p->threads.size();
            p->threads.emplace_back(worker_main, idx);
        }
        p->job = &job;
        p->want = want;
        p->running = want;
        ++p->epoch;
        p->cv_work.notify_all();
        p->cv_done.wait(lk, [&] { return p->running == 0; });
        p->job = nullptr;
        p->want = 0;
    }
    g_launches.fetch_add(1, std::memory_order_relaxed);
    if (job.failed.load()) {
        fprintf(stderr, "[s2ag emu] kernel %s failed\n", name);
        abort();
    }
}

// ---- the collectives ------------------------------------------------------------------------------------------------------
void shfl_fn(Wave* w) {
    uint64_t v[WAVE];
    bool have[WAVE];
    for (int l = 0; l < WAVE; ++l) {
        ShflRec* r = l < w->nlanes ? (ShflRec*)w->lanes[l]->rec : nullptr;
        have[l] = r != nullptr;
        v[l] = r ? r->v : 0;
    }
    for (int l = 0; l < w->nlanes; ++l) {
        ShflRec* r = (ShflRec*)w->lanes[l]->rec;
        if (!r) continue;
        const int src = (l & ~(r->width - 1)) | r->src;
        r->out = (src < WAVE && have[src]) ? v[src] : r->v;
    }
}
void ballot_fn(Wave* w) {
    uint64_t m = 0;
    for (int l = 0; l < w->nlanes; ++l) {
        BallotRec* r = (BallotRec*)w->lanes[l]->rec;
        if (r && r->pred) m |= 1ULL << l;
    }
    for (int l = 0; l < w->nlanes; ++l)
        if (BallotRec* r = (BallotRec*)w->lanes[l]->rec) r->out = m;
}
void rfl_fn(Wave* w) {
    uint32_t v = 0;
    for (int l = 0; l < w->nlanes; ++l)
        if (RflRec* r = (RflRec*)w->lanes[l]->rec) { v = r->v; break; }
    for (int l = 0; l < w->nlanes; ++l)
        if (RflRec* r = (RflRec*)w->lanes[l]->rec) r->out = v;
}
// A[i][k], B[k][j], C/D[4 * (lane >> 4) + reg][lane & 15]; KL = operand elements per lane, lane l holds k = KL * (l >> 4) + t
template <int KL, bool FMA_CHAIN>
static void mfma_generic(Wave* w) {
    constexpr int K = 4 * KL;
    float A[16][K], B[K][16];
    for (int l = 0; l < WAVE; ++l) {
        MfmaRec* r = l < w->nlanes ? (MfmaRec*)w->lanes[l]->rec : nullptr;
        for (int t = 0; t < KL; ++t) {
            A[l & 15][KL * (l >> 4) + t] = r ? r->a[t] : 0.f;
            B[KL * (l >> 4) + t][l & 15] = r ? r->b[t] : 0.f;
        }
    }
    for (int l = 0; l < w->nlanes; ++l) {
        MfmaRec* r = (MfmaRec*)w->lanes[l]->rec;
        if (!r) continue;
        const int col = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int row = 4 * (l >> 4) + j;
            if (FMA_CHAIN) {                         // v_mfma_f32_16x16x4_f32: bit for bit an fmaf chain over k
                float acc = r->c[j];
                for (int k = 0; k < K; ++k) acc = fmaf(A[row][k], B[k][col], acc);
                r->c[j] = acc;
            } else {                                 // bf16 operands: every product is exact; one rounding of the sum
                double acc = (double)r->c[j];
                for (int k = 0; k < K; ++k) acc += (double)A[row][k] * (double)B[k][col];
                r->c[j] = (float)acc;
            }
        }
    }
}
void mfma_k32_fn(Wave* w) { mfma_generic<8, false>(w); }
void mfma_k16_fn(Wave* w) { mfma_generic<4, false>(w); }
void mfma_k4_fn(Wave* w) { mfma_generic<1, true>(w); }
// ds_read_b64_tr_b16: within a group of 16 lanes, lane t supplies the address of 4 consecutive 16-bit elements; lane t
// receives element (t % 4) of the lanes 4 j + t / 4, j = 0..3  (== column t of the 4 x 16 matrix whose row j the lanes
// 4 j .. 4 j + 3 point at; probed on the hardware by tools/probe/tr16_probe.hip)
void tr16_fn(Wave* w) {
    for (int l = 0; l < w->nlanes; ++l) {
        Tr16Rec* r = (Tr16Rec*)w->lanes[l]->rec;
        if (!r) continue;
        if (((uintptr_t)r->addr & 7) != 0) fail("ds_read_b64_tr_b16 at an address that is not 8-byte aligned");
        const int g = l & ~15, t = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int src = g + 4 * j + t / 4;
            Tr16Rec* s = src < w->nlanes ? (Tr16Rec*)w->lanes[src]->rec : nullptr;
            r->out[j] = s ? s->addr[t % 4] : (short)0;
        }
    }
}
void wbar_fn(Wave*) {}

}  // namespace emu

extern "C" {
void s2ag_emu_set_sched(int mode, unsigned seed) {
    std::call_once(emu::g_once, emu::init_once);
    emu::g_sched_mode = mode;
    emu::g_sched_seed = seed;
}
void s2ag_emu_counters(unsigned long long* out2) {
    out2[0] = emu::g_launches.load();
    out2[1] = emu::g_collectives.load();
}
}
