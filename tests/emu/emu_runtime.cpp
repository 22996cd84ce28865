// TEST INFRASTRUCTURE ONLY — CPU emulator of the HIP execution model (see hip/hip_runtime.h).
// One workgroup runs at a time; each thread is a fiber with its own stack.  A fiber yields
// to the scheduler at __syncthreads() (block rendezvous) and at every wave-level collective
// (shuffle / ballot / MFMA: wave rendezvous + snapshot of every lane's deposit).
#include <hip/hip_runtime.h>
#include <vector>
#include <chrono>
#include <mutex>
#include <sys/mman.h>

uint3_ threadIdx, blockIdx;
dim3 blockDim, gridDim;

extern "C" void emu_swap(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_swap
.type emu_swap,@function
emu_swap:
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
)");

namespace {
enum { ST_READY = 0, ST_WAIT_BLOCK = 1, ST_WAIT_WAVE = 2, ST_DONE = 3 };
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = ST_DONE;
    uint3_ tid{0, 0, 0};
    alignas(16) char slot[emu::SLOT];
};

Fiber g_fibers[MAX_THREADS];
alignas(16) char g_snapshot[MAX_THREADS / 64][64][emu::SLOT];
unsigned long long g_active_mask[MAX_THREADS / 64];
void* g_sched_sp = nullptr;
int g_cur = -1;
const std::function<void()>* g_body = nullptr;
std::recursive_mutex g_mutex;   // the emulator is single-GPU: serialise concurrent host threads

void fiber_entry() {
    (*g_body)();
    Fiber& f = g_fibers[g_cur];
    f.state = ST_DONE;
    emu_swap(&f.sp, g_sched_sp);
    fprintf(stderr, "emu: resumed a finished fiber\n");
    abort();
}

void init_fiber(Fiber& f) {
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char*)MAP_FAILED) { perror("mmap"); abort(); }
    }
    uintptr_t top = ((uintptr_t)(f.stack + STACK_BYTES)) & ~(uintptr_t)15;
    uint64_t* sp = (uint64_t*)(top - 64);
    for (int i = 0; i < 6; ++i) sp[i] = 0;
    sp[6] = (uint64_t)(uintptr_t)&fiber_entry;
    sp[7] = 0;
    f.sp = sp;
    f.state = ST_READY;
}

void run_fiber(int i) {
    g_cur = i;
    threadIdx = g_fibers[i].tid;
    emu_swap(&g_sched_sp, g_fibers[i].sp);
}

void yield_to_sched() {
    Fiber& f = g_fibers[g_cur];
    int me = g_cur;
    emu_swap(&f.sp, g_sched_sp);
    g_cur = me;
    threadIdx = g_fibers[me].tid;
}

void run_block(int nthreads) {
    int nwaves = (nthreads + 63) / 64;
    for (int i = 0; i < nthreads; ++i) init_fiber(g_fibers[i]);
    for (;;) {
        bool any_alive = false;
        for (int w = 0; w < nwaves; ++w) {
            int lo = w * 64, hi = std::min(nthreads, lo + 64);
            for (;;) {
                for (int i = lo; i < hi; ++i)
                    if (g_fibers[i].state == ST_READY) run_fiber(i);
                int n_wave = 0, n_block = 0, n_done = 0;
                for (int i = lo; i < hi; ++i) {
                    int s = g_fibers[i].state;
                    n_wave += s == ST_WAIT_WAVE; n_block += s == ST_WAIT_BLOCK; n_done += s == ST_DONE;
                }
                if (n_wave == 0) break;                      // wave parked at block barrier or finished
                if (n_block != 0) {
                    fprintf(stderr, "emu: divergent wave: %d lanes at a wave collective, %d at __syncthreads\n", n_wave, n_block);
                    abort();
                }
                unsigned long long mask = 0;
                for (int i = lo; i < hi; ++i)
                    if (g_fibers[i].state == ST_WAIT_WAVE) {
                        memcpy(g_snapshot[w][i - lo], g_fibers[i].slot, emu::SLOT);
                        mask |= 1ull << (i - lo);
                        g_fibers[i].state = ST_READY;
                    }
                g_active_mask[w] = mask;
            }
        }
        int n_block = 0;
        for (int i = 0; i < nthreads; ++i) {
            n_block += g_fibers[i].state == ST_WAIT_BLOCK;
            any_alive |= g_fibers[i].state != ST_DONE;
        }
        if (!any_alive) break;
        if (n_block == 0) { fprintf(stderr, "emu: scheduler deadlock\n"); abort(); }
        // HIP semantics: all non-exited threads must reach the barrier
        for (int i = 0; i < nthreads; ++i)
            if (g_fibers[i].state == ST_WAIT_BLOCK) g_fibers[i].state = ST_READY;
    }
}

struct Node { dim3 grid, block; std::function<void()> body; int kind; void* dst; const void* src; size_t n; int val; };
}  // namespace

struct emuStream_ { bool capturing = false; std::vector<Node>* cap = nullptr; emuStream_* origin = nullptr; std::vector<emuStream_*> joined; };
struct emuEvent_ { std::chrono::steady_clock::time_point t; emuStream_* cap_src = nullptr; };
struct emuGraph_ { std::vector<Node> nodes; };
struct emuGraphExec_ { std::vector<Node> nodes; };
static emuStream_ g_null_stream;
static emuStream_* S(hipStream_t s) { return s ? s : &g_null_stream; }

static void exec_kernel(dim3 grid, dim3 block, const std::function<void()>& body) {
    std::lock_guard<std::recursive_mutex> lk(g_mutex);
    int nthreads = block.x * block.y * block.z;
    if (nthreads <= 0 || nthreads > MAX_THREADS) { fprintf(stderr, "emu: bad block size %d\n", nthreads); abort(); }
    gridDim = grid; blockDim = block;
    g_body = &body;
    for (int i = 0; i < nthreads; ++i) {
        g_fibers[i].tid.x = i % block.x;
        g_fibers[i].tid.y = (i / block.x) % block.y;
        g_fibers[i].tid.z = i / (block.x * block.y);
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                run_block(nthreads);
            }
    g_body = nullptr;
}

static void exec_node(const Node& n) {
    if (n.kind == 0) exec_kernel(n.grid, n.block, n.body);
    else if (n.kind == 1) memcpy(n.dst, n.src, n.n);
    else if (n.kind == 2) memset(n.dst, n.val, n.n);
}

namespace emu {
void launch(dim3 grid, dim3 block, hipStream_t s, std::function<void()> body) {
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
    Node n{grid, block, std::move(body), 0, nullptr, nullptr, 0, 0};
    if (S(s)->capturing) { S(s)->cap->push_back(std::move(n)); return; }
    exec_node(n);
}
void* wave_exchange(const void* mine, size_t bytes) {
    if (bytes > (size_t)SLOT) { fprintf(stderr, "emu: wave_exchange payload too large\n"); abort(); }
    Fiber& f = g_fibers[g_cur];
    memcpy(f.slot, mine, bytes);
    f.state = ST_WAIT_WAVE;
    int w = g_cur / 64;
    yield_to_sched();
    return g_snapshot[w];
}
void block_barrier() {
    g_fibers[g_cur].state = ST_WAIT_BLOCK;
    yield_to_sched();
}
int lane_id() { return g_cur & 63; }
}  // namespace emu

unsigned long long emu_ballot(int pred) {
    int p = pred != 0;
    char* base = (char*)emu::wave_exchange(&p, sizeof(int));
    unsigned long long active = g_active_mask[g_cur / 64], m = 0;
    for (int i = 0; i < 64; ++i)
        if ((active >> i) & 1) { int v; memcpy(&v, base + (size_t)i * emu::SLOT, 4); if (v) m |= 1ull << i; }
    return m;
}

float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
int atomicMax(int* p, int v) { int o = *p; *p = std::max(o, v); return o; }

// ---- MFMA: every lane deposits (a, b, c); results follow the gfx950 fragment maps ----------
namespace {
struct Dep4 { float a, b; float c[4]; };
struct Dep16 { float a, b; float c[16]; };
struct DepB4 { short a[8], b[8]; float c[4]; };
struct DepB16 { short a[8], b[8]; float c[16]; };
inline float bf2f(short h) { unsigned u = ((unsigned)(unsigned short)h) << 16; float f; memcpy(&f, &u, 4); return f; }
template <typename D> const D& dep(char* base, int lane) { return *(const D*)(base + (size_t)lane * emu::SLOT); }
}

emu_v4f emu_mfma_f32_16x16x4f32(float a, float b, emu_v4f c) {
    Dep4 d{a, b, {c[0], c[1], c[2], c[3]}};
    char* base = (char*)emu::wave_exchange(&d, sizeof d);
    int l = emu::lane_id();
    emu_v4f out;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(dep<Dep4>(base, row + 16 * k).a, dep<Dep4>(base, col + 16 * k).b, acc);
        out[r] = acc;
    }
    return out;
}

emu_v16f emu_mfma_f32_32x32x2f32(float a, float b, emu_v16f c) {
    Dep16 d; d.a = a; d.b = b; for (int i = 0; i < 16; ++i) d.c[i] = c[i];
    char* base = (char*)emu::wave_exchange(&d, sizeof d);
    int l = emu::lane_id();
    emu_v16f out;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(dep<Dep16>(base, row + 32 * k).a, dep<Dep16>(base, col + 32 * k).b, acc);
        out[r] = acc;
    }
    return out;
}

emu_v4f emu_mfma_f32_16x16x32_bf16(emu_v8s a, emu_v8s b, emu_v4f c) {
    DepB4 d; for (int i = 0; i < 8; ++i) { d.a[i] = a[i]; d.b[i] = b[i]; } for (int i = 0; i < 4; ++i) d.c[i] = c[i];
    char* base = (char*)emu::wave_exchange(&d, sizeof d);
    int l = emu::lane_id();
    emu_v4f out;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc += bf2f(dep<DepB4>(base, row + 16 * (k >> 3)).a[k & 7]) * bf2f(dep<DepB4>(base, col + 16 * (k >> 3)).b[k & 7]);
        out[r] = acc;
    }
    return out;
}

// ---- OCP fp8 e4m3 (the gfx950 format of v_cvt_pk_fp8_f32 / v_mfma_f32_16x16x32_fp8_fp8): 1 sign, 4 exponent (bias 7), 3 mantissa bits, no infinities,
// 0x7f / 0xff = NaN, largest finite value 448
float emu_fp8_to_f32(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = ldexpf((float)m / 8.0f, -6);
    else r = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -r : r;
}
unsigned char emu_f32_to_fp8(float x) {                       // round to nearest even, saturating to +-448
    if (x != x) return 0x7f;
    const unsigned char sign = std::signbit(x) ? 0x80 : 0;
    const float a = fabsf(x);
    if (a >= 448.0f) return sign | 0x7e;
    int best = 0; float bd = a;                               // code 0 = +0
    for (int code = 1; code <= 0x7e; ++code) {
        const float d = fabsf(emu_fp8_to_f32((unsigned char)code) - a);
        if (d < bd || (d == bd && (code & 1) == 0)) { bd = d; best = code; }
    }
    return sign | (unsigned char)best;
}
int emu_cvt_pk_fp8_f32(float a, float b, int old, bool word_sel) {
    const unsigned pair = (unsigned)emu_f32_to_fp8(a) | ((unsigned)emu_f32_to_fp8(b) << 8);
    const unsigned o = (unsigned)old;
    return (int)(word_sel ? ((o & 0x0000ffffu) | (pair << 16)) : ((o & 0xffff0000u) | pair));
}
struct DepF8 { unsigned char a[8], b[8]; float c[4]; };
emu_v4f emu_mfma_f32_16x16x32_fp8_fp8(long a, long b, emu_v4f c) {
    DepF8 d; memcpy(d.a, &a, 8); memcpy(d.b, &b, 8); for (int i = 0; i < 4; ++i) d.c[i] = c[i];
    char* base = (char*)emu::wave_exchange(&d, sizeof d);
    int l = emu::lane_id();
    emu_v4f out;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc += emu_fp8_to_f32(dep<DepF8>(base, row + 16 * (k >> 3)).a[k & 7]) * emu_fp8_to_f32(dep<DepF8>(base, col + 16 * (k >> 3)).b[k & 7]);
        out[r] = acc;
    }
    return out;
}

emu_v16f emu_mfma_f32_32x32x16_bf16(emu_v8s a, emu_v8s b, emu_v16f c) {
    DepB16 d; for (int i = 0; i < 8; ++i) { d.a[i] = a[i]; d.b[i] = b[i]; } for (int i = 0; i < 16; ++i) d.c[i] = c[i];
    char* base = (char*)emu::wave_exchange(&d, sizeof d);
    int l = emu::lane_id();
    emu_v16f out;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc += bf2f(dep<DepB16>(base, row + 32 * (k >> 3)).a[k & 7]) * bf2f(dep<DepB16>(base, col + 32 * (k >> 3)).b[k & 7]);
        out[r] = acc;
    }
    return out;
}

// ---- host runtime -----------------------------------------------------------------------
// Guard-page allocator: the buffer END is flush (to 16 B) against a PROT_NONE page and a PROT_NONE page precedes the
// mapping, so an out-of-bounds read past the end of any device buffer faults immediately under the emulator.
#include <map>
static std::map<void*, std::pair<void*, size_t>> g_allocs;
static std::mutex g_alloc_mutex;
extern "C" void* emu_guard_alloc(size_t n) {
    const size_t page = 4096, n16 = (n + 15) / 16 * 16, body = (n16 + page - 1) / page * page;
    char* base = (char*)mmap(nullptr, body + 2 * page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == (char*)MAP_FAILED) return nullptr;
    mprotect(base, page, PROT_NONE);
    mprotect(base + page + body, page, PROT_NONE);
    char* p = base + page + body - n16;
    memset(p, 0xFF, n16);                               // poison: uninitialised reads show up as NaN
    std::lock_guard<std::mutex> lk(g_alloc_mutex);
    g_allocs[p] = {base, body + 2 * page};
    return p;
}
extern "C" void emu_guard_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_alloc_mutex);
    auto it = g_allocs.find(p);
    if (it == g_allocs.end()) { fprintf(stderr, "emu: free of unknown pointer\n"); abort(); }
    munmap(it->second.first, it->second.second);
    g_allocs.erase(it);
}
hipError_t hipMalloc(void** p, size_t n) { *p = emu_guard_alloc(n); return *p ? hipSuccess : hipErrorUnknown; }
hipError_t hipFree(void* p) { emu_guard_free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
    if (S(st)->capturing) { S(st)->cap->push_back(Node{dim3(), dim3(), nullptr, 1, d, s, n, 0}); return hipSuccess; }
    memcpy(d, s, n); return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
    if (S(st)->capturing) { S(st)->cap->push_back(Node{dim3(), dim3(), nullptr, 2, d, nullptr, n, v}); return hipSuccess; }
    memset(d, v, n); return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st) {
    for (size_t r = 0; r < h; ++r) hipMemcpyAsync((char*)d + r * dp, (const char*)s + r * sp, w, k, st);
    return hipSuccess;
}
hipError_t hipMemset2DAsync(void* d, size_t dp, int v, size_t w, size_t h, hipStream_t st) {
    for (size_t r = 0; r < h; ++r) hipMemsetAsync((char*)d + r * dp, v, w, st);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emuStream_(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent_(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    e->t = std::chrono::steady_clock::now();
    e->cap_src = S(s)->capturing ? (S(s)->origin ? S(s)->origin : S(s)) : nullptr;      // recorded inside a capture: a dependency marker
    return hipSuccess;
}
// cross-stream capture (fork / join through events): a stream that waits on an event recorded in a capturing stream joins that capture -
// its launches are appended to the same node list (the emulator executes a graph in capture order, which respects every dependency)
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    emuStream_* st = S(s);
    if (e->cap_src && e->cap_src->capturing && !st->capturing) {
        st->capturing = true; st->cap = e->cap_src->cap; st->origin = e->cap_src; e->cap_src->joined.push_back(st);
    }
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
    S(s)->capturing = true; S(s)->cap = new std::vector<Node>(); return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
    for (emuStream_* j : S(s)->joined) { j->capturing = false; j->cap = nullptr; j->origin = nullptr; }
    S(s)->joined.clear();
    *g = new emuGraph_(); (*g)->nodes = std::move(*S(s)->cap); delete S(s)->cap; S(s)->cap = nullptr; S(s)->capturing = false; return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { *e = new emuGraphExec_(); (*e)->nodes = g->nodes; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { for (auto& n : e->nodes) exec_node(n); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
