// Dev tool (round 6): ONE Qwen2-0.5B decode layer at batch 1 as a persistent launch - the SKELETON of the guide's engine (MI355X_MICROARCH.md
// "engine-vs-launches": 1 LDS-DMA loader wave per CU running ahead through the layer's weight slice, consumer waves, 8-byte tagged granules for every vector that
// crosses CUs), with the layer's real data movement and its real dependency edges, and the arithmetic at its real size but without claim to correct values
// (random data; a product build would need the exact fp32 orders of llm_kernels.h on top).  What it measures is the FLOOR of that design on this layer: if the
// skeleton does not beat the five-launch chain (21.9 us per layer inside the decode hipGraph), no engine will.
//
//   256 workgroups (one per CU: 154 KB of LDS each), 4 waves: wave 0 = loader, waves 1-3 = consumers (wave 1 also gathers).
//   Per layer and CU: 120 KB of bf16 weights through a 32 x 4 KB LDS ring (qkv 2 slots, o 2, gate/up 17, down 9 = the CU's rows of each matrix), 30.7 MB per layer
//   over the chip = the layer's 29.8 MB.
//   Edges (all-gathers through L2, granule = {fp32 value, sequence tag} written by ONE 8-byte sc1 store, swept by wave 1 of every consuming CU with sc1 loads):
//     E1 qkv (1152 values) -> the 14 attention CUs     E2 attention (896) -> all     E3 o_proj + residual (896) -> all
//     E4 silu(gate) * up (4864) -> all                 E5 down + residual (896) -> all (the next layer's input)
//   Attention: 14 CUs stream their head's K / V rows (context 256: 131 KB) from memory; the other CUs wait at E2 with their loader running ahead.
//   Every wait is bounded by a wall-clock deadline: a protocol error ends the launch with an error flag instead of hanging the box.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cosyvoice_amd/csrc -I include tools/ubench/layer_engine_probe.hip -o tools/ubench/layer_engine_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"
using namespace cv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NCU = 256, HID = 896, QKV = 1152, INTER = 4864, NHEAD = 14, CTX = 256;
constexpr int SLOT = 4096, NSLOT = 32;                       // LDS ring
constexpr int S_QKV = 2, S_O = 2, S_GU = 17, S_DOWN = 9, S_LAYER = S_QKV + S_O + S_GU + S_DOWN;      // 4 KB slots per operator and CU
constexpr int E_N[5] = {QKV, HID, HID, INTER, HID};
constexpr int E_OFF[5] = {0, QKV, QKV + HID, QKV + 2 * HID, QKV + 2 * HID + INTER};
constexpr int E_TOTAL = QKV + 3 * HID + INTER;

struct EngineArgs {
    const unsigned char* weights;        // [NCU][S_LAYER * SLOT] bytes, re-read every layer (a product would index by layer: same traffic, the data comes from HBM / MALL either way)
    unsigned long long* edges;           // [2 parities][E_TOTAL] granules
    const float* kv;                     // [NHEAD][2][CTX][64] fp32
    float* out;                          // [NCU]: one number per workgroup (keeps the arithmetic alive)
    int layers;
    int* err;
    long long* stamps;                   // [layers + 2] wall_clock64 of workgroup 0
    long long deadline_ticks;            // wall-clock budget of the launch (100 MHz ticks)
    int mode;                            // bit 0: no weight stream (LDS ring never refilled); bit 1: no arithmetic; bit 2: no attention K / V stream
};

__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_granule(unsigned long long* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void engine_kernel(EngineArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[NSLOT * SLOT];           // 128 KB
    __shared__ __attribute__((aligned(16))) float vec[INTER];                            // the gathered vector of the current operator (19 KB)
    __shared__ volatile unsigned filled;                                                // slots landed so far (monotonic, written by the loader)
    __shared__ volatile unsigned done[3];                                               // operators finished per consumer wave (monotonic)
    __shared__ volatile unsigned gathered;                                              // edges gathered so far (monotonic, written by wave 1)
    __shared__ volatile int abort_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cu = blockIdx.x;
    if (tid == 0) { filled = 0; done[0] = done[1] = done[2] = 0; gathered = 0; abort_flag = 0; }
    __syncthreads();
    const long long t_end = wall_clock64() + p.deadline_ticks;
    auto bail = [&]() { if (wall_clock64() > t_end) { abort_flag = 1; if (lane == 0) atomicExch(p.err, 1); } return abort_flag != 0; };
    const int total_slots = p.layers * S_LAYER;

    if (wave == 0) {
        // ---- loader: the CU's weight slice, slot after slot, as far ahead as the ring allows (a slot is free once every consumer wave finished the operator that read it)
        if (p.mode & 1) { if (lane == 0) filled = (unsigned)total_slots; return; }
        const unsigned char* src = p.weights + (size_t)cu * S_LAYER * SLOT;
        // operator index -> first slot after it (stream order): consumers finish operators, the loader converts to slots
        int issued = 0;
        while (issued < total_slots) {
            // slots consumed = slots of the operators all three waves have finished
            unsigned d = min(done[0], min(done[1], done[2]));
            const int lay = d / 4, op = d % 4;
            const int consumed = lay * S_LAYER + (op > 0 ? S_QKV : 0) + (op > 1 ? S_O : 0) + (op > 2 ? S_GU : 0);
            if (issued - consumed >= NSLOT) { if (bail()) return; __builtin_amdgcn_s_sleep(2); continue; }
            const int n = min(min(NSLOT - (issued - consumed), total_slots - issued), 8);          // up to eight slots (32 DMA pieces) per round
            for (int k = 0; k < n; ++k) {
                const int s = issued + k, in_layer = s % S_LAYER;
#pragma unroll
                for (int q = 0; q < 4; ++q) CV_GLDS16(src + (size_t)in_layer * SLOT + q * 1024 + lane * 16, ring + (s % NSLOT) * SLOT + q * 1024);
            }
            CV_VMCNT0();
            issued += n;
            if (lane == 0) filled = (unsigned)issued;
        }
        return;
    }
    // ---- consumers
    const int cw = wave - 1;                                  // 0 .. 2
    float accum = 0.f;
    unsigned seq = 0;                                         // edges gathered so far by this CU's protocol position
    unsigned nops = 0;                                        // operators this wave has finished
    int slot0 = 0;                                            // first ring slot of the current operator (stream order)
    // wait until the vector of edge e (layer parity par) is in `vec`: wave 1 sweeps the granules, the others wait for its LDS flag
    auto gather = [&](int e, int lay) -> bool {
        ++seq;
        if (cw == 0) {
            const unsigned long long* g = p.edges + (size_t)(lay & 1) * E_TOTAL + E_OFF[e];
            const unsigned tag = (unsigned)(lay * 8 + e + 1);
            const int n = E_N[e];
            // `vec` is still being read by the other consumer waves until they finished the previous operator
            while (done[1] < nops || done[2] < nops) { if (bail()) return false; __builtin_amdgcn_s_sleep(1); }
            for (int base = 0; base < n; base += 64 * 16) {                   // one pass = 16 loads in flight per lane, then the tag checks; repeated until the chunk is complete
                unsigned long long v[16];
                bool ok;
                do {
                    ok = true;
#pragma unroll
                    for (int j = 0; j < 16; ++j) { const int idx = base + j * 64 + lane; v[j] = idx < n ? ld_granule(g + idx) : ((unsigned long long)tag << 32); }
#pragma unroll
                    for (int j = 0; j < 16; ++j) ok = ok && (unsigned)(v[j] >> 32) == tag;
                    ok = __all(ok);
                    if (!ok && bail()) return false;
                } while (!ok);
#pragma unroll
                for (int j = 0; j < 16; ++j) { const int idx = base + j * 64 + lane; if (idx < n) vec[idx] = __uint_as_float((unsigned)v[j]); }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) gathered = seq;
        } else {
            while (gathered < seq) { if (bail()) return false; __builtin_amdgcn_s_sleep(1); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        return true;
    };
    // rows [r0, r0 + nrows) of this CU for an operator with K inputs out of `vec`, weights from ring slots starting at slot0: row r of the CU goes to wave r % 3; output j
    // of the chip = cu + 256 * r, published as edge e_out (nullptr: kept).  Returns false on abort.
    auto gemv = [&](int nrows, int K, int nslots, int e_out, int lay, bool pair) -> bool {
        while ((int)filled < slot0 + nslots) { if (bail()) return false; __builtin_amdgcn_s_sleep(1); }
        const unsigned short* w = reinterpret_cast<const unsigned short*>(ring + (slot0 % NSLOT) * SLOT);           // (the skeleton lets a row run over the ring's end: wrap by index)
        const int ring_elems = NSLOT * SLOT / 2, base = (slot0 % NSLOT) * (SLOT / 2);
        const unsigned short* r0 = reinterpret_cast<const unsigned short*>(ring);
        unsigned long long* g = p.edges + (size_t)(lay & 1) * E_TOTAL + E_OFF[e_out];
        const unsigned tag = (unsigned)(lay * 8 + e_out + 1);
        for (int r = cw; r < nrows; r += 3) {
            float s = 0.f;
            if (!(p.mode & 2)) {
                const int stride = pair ? 2 * K : K;
                float s2 = 0.f;
                for (int k = 2 * lane; k < K; k += 128) {                                  // two bf16 per LDS read
                    const int idx = (base + r * stride + k) % ring_elems;
                    const unsigned w2 = *reinterpret_cast<const unsigned*>(r0 + idx);
                    s += __uint_as_float(w2 << 16) * vec[k] + __uint_as_float(w2 & 0xffff0000u) * vec[k + 1];
                    if (pair) { const unsigned u2 = *reinterpret_cast<const unsigned*>(r0 + (idx + K) % ring_elems); s2 += __uint_as_float(u2 << 16) * vec[k] + __uint_as_float(u2 & 0xffff0000u) * vec[k + 1]; }
                }
                s = wave_sum(s);
                if (pair) { s2 = wave_sum(s2); s = s * fast_rcp(1.f + fast_exp(-s)) * s2; }
            }
            accum += s;
            const int j = cu + NCU * r;
            if (lane == 0 && j < E_N[e_out]) st_granule(g + j, s * 1e-3f + 0.01f * (float)(j & 7), tag);
        }
        (void)w;
        slot0 += nslots;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        ++nops;
        if (lane == 0) done[cw] = nops;
        return true;
    };

    if (cu == 0 && tid == 64) p.stamps[0] = wall_clock64();
    for (int lay = 0; lay < p.layers; ++lay) {
        // input of the layer: E5 of the previous one (layer 0: pre-published by the host as tag 0 * 8 + 4 + 1 of parity 1 ... see main)
        if (!gather(4, lay - 1 + 2)) break;                  // (lay - 1) with the parity kept non-negative: layer -1 = host-written, tag (lay + 1) * 8 ... handled by the host
        if (!gemv((QKV + NCU - 1) / NCU, HID, S_QKV, 0, lay + 2, false)) break;              // qkv -> E1
        if (cu < NHEAD) {                                    // attention on 14 CUs
            if (!gather(0, lay + 2)) break;
            float s = 0.f;
            if (!(p.mode & 4)) {
                const float4* kv = reinterpret_cast<const float4*>(p.kv + (size_t)cu * 2 * CTX * 64);
                for (int i = cw * 64 + lane; i < 2 * CTX * 64 / 4; i += 192) { const float4 v = kv[i]; s += v.x * vec[i & 63] + v.y + v.z + v.w; }
                s = wave_sum(s);
            }
            accum += s;
            unsigned long long* g = p.edges + (size_t)((lay + 2) & 1) * E_TOTAL + E_OFF[1];
            const unsigned tag = (unsigned)((lay + 2) * 8 + 1 + 1);
            if (lane < 22 && cw * 22 + lane < 64) st_granule(g + cu * 64 + cw * 22 + lane, s * 1e-3f + 0.001f * lane, tag);       // every wave publishes its third after ITS share of the stream
        } else ++seq;                                        // (keeps the edge count in step: this CU does not gather E1)
        if (!gather(1, lay + 2)) break;
        if (!gemv((HID + NCU - 1) / NCU, HID, S_O, 2, lay + 2, false)) break;                // o_proj (+ residual) -> E3
        if (!gather(2, lay + 2)) break;
        if (!gemv(INTER / NCU, HID, S_GU, 3, lay + 2, true)) break;                          // gate / up, SiLU(g) * u -> E4
        if (!gather(3, lay + 2)) break;
        if (!gemv((HID + NCU - 1) / NCU, INTER, S_DOWN, 4, lay + 2, false)) break;           // down (+ residual) -> E5
        if (cu == 0 && tid == 64) p.stamps[lay + 1] = wall_clock64();
    }
    if (lane == 0) p.out[cu * 4 + wave] = accum;
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 48;
    unsigned char* dw; unsigned long long* de; float *dkv, *dout; int* derr; long long* dst;
    CK(hipMalloc(&dw, (size_t)NCU * S_LAYER * SLOT)); CK(hipMalloc(&de, (size_t)2 * E_TOTAL * 8)); CK(hipMalloc(&dkv, (size_t)NHEAD * 2 * CTX * 64 * 4));
    CK(hipMalloc(&dout, NCU * 4 * 4)); CK(hipMalloc(&derr, 4)); CK(hipMalloc(&dst, (layers + 2) * 8));
    {
        std::vector<unsigned short> hw((size_t)NCU * S_LAYER * SLOT / 2); for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0x1ff));        // small positive bf16
        CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> hk((size_t)NHEAD * 2 * CTX * 64); for (auto& v : hk) v = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(dkv, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    }
    for (int mode : {0, 1, 2, 4, 7}) {
        // the input of layer 0: E5 of "layer 1" (= lay - 1 + 2 for lay = 0), parity 1, tag 1 * 8 + 4 + 1
        std::vector<unsigned long long> he((size_t)2 * E_TOTAL, 0ull);
        for (int i = 0; i < HID; ++i) { float v = 0.01f * (i & 15); unsigned u; memcpy(&u, &v, 4); he[(size_t)1 * E_TOTAL + E_OFF[4] + i] = ((unsigned long long)(1 * 8 + 4 + 1) << 32) | u; }
        float best = 1e30f; int err = 0; std::vector<long long> hs(layers + 2);
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemcpy(de, he.data(), he.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(derr, 0, 4)); CK(hipMemset(dst, 0, (layers + 2) * 8));
            EngineArgs a{dw, de, dkv, dout, layers, derr, dst, 100000000LL / 20, mode};          // 50 ms budget
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(engine_kernel, dim3(NCU), dim3(256), 0, 0, a);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&err, derr, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
            if (err) break;
            best = fminf(best, ms);
        }
        if (err) { printf("mode %d: PROTOCOL TIMEOUT (error flag set; the launch ended on its deadline)\n", mode); continue; }
        // steady state: layers 8 .. layers - 1 of workgroup 0's stamps
        const double per = (hs[layers] - hs[8]) * 0.01 / (layers - 8);
        printf("mode %d (%s): launch %.1f us for %d layers; steady state %.2f us per layer (workgroup 0, layers 8..%d)  [five-launch chain: 21.9 us per layer; go <= 18.5]\n", mode,
               mode == 0 ? "full skeleton" : mode == 1 ? "no weight stream" : mode == 2 ? "no arithmetic" : mode == 4 ? "no attention K/V stream" : "edges only", best * 1e3, layers, per, layers - 1);
    }
    return 0;
}
