// Lock-step batched decode (BASELINE.json configs[2]/[3]/[4], SURVEY.md §8e "LLM continuous batching"): up to 16 sequences advance by
// one token per step and every weight matrix is streamed from HBM ONCE per step for all of them.
//
// With nb sequences a decode "GEMV" is a skinny GEMM  Y[b][n] = sum_k W[n][k] X[b][k]  (b < 16): it goes on the matrix pipe at fp32 accuracy -
// v_mfma_f32_16x16x32_bf16 on the exact three-term bf16 split of the fp32 activations (X3, see below; the fp32 chain v_mfma_f32_16x16x4_f32 is
// kept as the A/B variant) - with the weight
// tile as the MFMA "A" operand (16 weight rows) and the sequences as the 16 "B" columns - one MFMA instruction serves all 16 sequences,
// so a step costs the weight stream once plus the per-sequence activations (the first version of this file looped the batch-1 VALU GEMV
// over the sequences: 256 VGPRs, one dependent L2 round trip per sequence, 32 us per gate/up launch at nb = 8 against 6 us for nb = 1).
//
//   * Every column (sequence) of an MFMA is an independent fp32 dot product in a fixed k order: what a sequence computes does not depend
//     on its slot nor on what the other slots hold (SURVEY.md §8e determinism requirement).  The summation ORDER differs from the
//     batch-1 kernels (llm_kernels.h), so logits agree with them to rounding (1e-6 relative), not bit for bit; greedy tokens are
//     compared with the single path and with the oracle in tests/test_zz_llm_batch.py.
//   * Latency structure as in llm_kernels.h: a wave first requests ALL of its weight and activation fragments (weights non-temporal, activations
//     straight from L2).  The 4 waves of a workgroup split K; their accumulators are combined through LDS in a fixed order.
//   * The binding resource of these kernels is what one CU can ingest (a few tens of bytes per cycle): every workgroup needs the X
//     columns of its K range for all sequences, so the shapes are cut so that X bytes per workgroup stay close to its weight bytes -
//     down (K = 4864) is split 8 ways over K across workgroups, its partial sums are combined (fixed order, + residual) by
//     sum_partials_kernel.  (Letting the LAST workgroup of a row group to arrive do that combine inside the same launch - partials, device-scope
//     fence, arrival counter - was built and measured: the two fences per workgroup write back / invalidate an XCD's L2 each and the step went
//     1064 -> 1642 us at 8 sequences; profiles/r2_batch_decode_ab.txt.  A second launch is the cheap way to make partials visible across XCDs.)
#pragma once
#include "llm_kernels.h"

namespace cv {

constexpr int MAX_NB = 32;        // lock-step slots: 16 per MFMA column tile, two column tiles per weight fragment (skinny_pk2_kernel)

struct SkinnyArgs {
    const bf16_t* W; const float* bias;
    const float* x; long long ldx;            // X [nb][K] fp32
    float* y; long long ldy;                  // mode 0: [nb][N]   mode 1: [nb][N/2]   mode 2: [ksplit][nb][N] (ldy = N)
    int N, K;
    const float* gamma; float eps;            // fused Qwen2RMSNorm over the whole row (ksplit must be 1)
    const float* res; long long ldres;
    int mode;                                 // 0: acc + bias + res   1: rows interleaved (gate_j, up_j): silu(gate) * up   2: raw split-K partial
    int nb, ksplit;
    long long* dbg;                           // dev tool (tools/ubench/skinny_probe.hip): clock64() of thread 0 at the phase boundaries, 8 slots per workgroup; null in production
};

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// Workgroup = NW waves sharing RT row tiles (16 weight rows each) and one K range (K / ksplit); wave w takes the k-tiles (32 columns)
// [w * T / NW, (w + 1) * T / NW) of the range, at most KTW of them.  Lane l: weight row l % 16 (A operand), sequence l % 16 (B operand),
// k slot group g = l / 16: 8 consecutive k per 32-wide tile -> one 16-byte weight load (8 bf16) and two float4 X loads per tile feed 8
// MFMAs.  D: lane l holds y[seq l % 16][row 4 * (l / 16) + i], i = 0..3: one float4 store per sequence.  No barrier before the MFMA chain:
// the RMSNorm scale is applied to the accumulators (see below), the only barrier is the cross-wave combine at the end.
// (Staging X through LDS - one coalesced fetch per workgroup, conflict-free fragment reads - was built and measured SLOWER on the
// MI355X: 15.9 vs 12.8 us for gate/up at nb = 8, 8.3 vs 5.6 us for qkv / o_proj: two more barriers in front of the MFMA chain cost more
// than the 28 extra L2 load instructions per lane; profiles/r2_batch_decode_ab.txt.)
//
// X3 (default): the products run on v_mfma_f32_16x16x32_bf16 instead of eight v_mfma_f32_16x16x4_f32 per tile.  The weights ARE bf16 - the 16-byte
// load is the A operand as it stands, no unpacking - and an fp32 activation is the exact sum of three bf16 numbers (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = x - x1 - x2: 3 x 8 mantissa bits, the residuals are exact in fp32), so  W x = W x1 + W x2 + W x3  with every product exact and the
// accumulation in fp32: the same accuracy as the fp32 pipe for 3 MFMA issues of 32 cycles per row tile and k-tile instead of 8.  The fp32 MFMA
// chain was the longest thing a workgroup does once its loads arrive (gate/up: 112 x 32 cycles per wave, twice that on a CU that holds two
// workgroups).  Still a fixed order per sequence, independent of the slot and of the other slots.
__device__ __forceinline__ void split3_bf16(const float4 a, const float4 b, u32x4& h1, u32x4& h2, u32x4& h3) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float r[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u = pack_bf16x2(x[2 * i], x[2 * i + 1]);
        h1[i] = u; r[2 * i] = x[2 * i] - bf_lo(u); r[2 * i + 1] = x[2 * i + 1] - bf_hi(u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u = pack_bf16x2(r[2 * i], r[2 * i + 1]);
        h2[i] = u; r[2 * i] -= bf_lo(u); r[2 * i + 1] -= bf_hi(u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) h3[i] = pack_bf16x2(r[2 * i], r[2 * i + 1]);
}

// NW = waves per workgroup (they split the K range).  4 everywhere: 8-wave workgroups for the narrow GEMMs (qkv, o_proj: only 72 / 56 workgroups of
// one row tile exist; 4 k-tiles per wave instead of 7) measured no better - LM step 1013 -> 1032 us at 8 sequences, profiles/r2_batch_decode_ab.txt.
template <int RT, int KTW, bool X3 = true, int NW = 4>
__global__ __launch_bounds__(NW * 64) void skinny_mfma_kernel(SkinnyArgs p) {
    static_assert(RT >= 1 && RT <= 4 && RT <= NW, "the final combine hands one row tile to each wave");
    __shared__ __attribute__((aligned(16))) float red[NW * RT * 256];
    __shared__ float ssq[NW][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ks = blockIdx.x % p.ksplit, rg = blockIdx.x / p.ksplit;
    const int krange = p.K / p.ksplit, tiles = krange / 32;
    const int t0 = wave * tiles / NW, t1 = (wave + 1) * tiles / NW;
    const int kbase = ks * krange + t0 * 32 + g * 8;
    const int n_base = rg * RT * 16;
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();

    // every load of the wave is requested up front, tile by tile in the order the MFMA loop consumes them (W, X, gamma of tile 0, then tile 1 ...):
    // the counted vmcnt waits the compiler places in front of each tile's MFMAs then let tile t compute while tiles t+1 .. are still in flight
    u32x4 w[RT][KTW];
    float4 xa[KTW], xb[KTW], ga[KTW], gb[KTW];
    const bf16_t* wr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wr[rt] = p.W + (long long)min(n_base + rt * 16 + c, p.N - 1) * p.K + kbase;   // clamped: ragged last tile (stores are masked)
    const float* xp = p.x + (long long)min(c, p.nb - 1) * p.ldx + kbase;       // columns >= nb recompute the last sequence (never stored)
#pragma unroll
    for (int t = 0; t < KTW; ++t) {
        const bool ok = t0 + t < t1;
        const int tt = ok ? t : 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr[rt] + tt * 32));
            if (!ok) v = (u32x4){0u, 0u, 0u, 0u};
            w[rt][t] = v;
        }
        xa[t] = *reinterpret_cast<const float4*>(xp + tt * 32);
        xb[t] = *reinterpret_cast<const float4*>(xp + tt * 32 + 4);
        if (!ok) { xa[t] = make_float4(0.f, 0.f, 0.f, 0.f); xb[t] = xa[t]; }
        if (p.gamma) {
            ga[t] = *reinterpret_cast<const float4*>(p.gamma + kbase + tt * 32);
            gb[t] = *reinterpret_cast<const float4*>(p.gamma + kbase + tt * 32 + 4);
        }
    }
    // Fused Qwen2RMSNorm: y = W (x * rstd * gamma) = rstd * (W (x * gamma)) per sequence - the MFMAs run on x * gamma while the sum of squares is
    // still being gathered, and rstd (one scalar per sequence = per MFMA column) is applied to the accumulators at the end.
    float ss = 0.f;
    v4f acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KTW; ++t) {
        if (t0 + t < t1) {                                              // wave-uniform
            float4 a4 = xa[t], b4 = xb[t];
            if (p.gamma) {
                ss += a4.x * a4.x + a4.y * a4.y + a4.z * a4.z + a4.w * a4.w + b4.x * b4.x + b4.y * b4.y + b4.z * b4.z + b4.w * b4.w;
                a4.x *= ga[t].x; a4.y *= ga[t].y; a4.z *= ga[t].z; a4.w *= ga[t].w;
                b4.x *= gb[t].x; b4.y *= gb[t].y; b4.z *= gb[t].z; b4.w *= gb[t].w;
            }
            if constexpr (X3) {
                u32x4 h1, h2, h3;
                split3_bf16(a4, b4, h1, h2, h3);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {                       // smallest term first: the order in which a compensated sum would add them
                    const v8bf wf = __builtin_bit_cast(v8bf, w[rt][t]);
                    v4f a = acc[rt];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h3), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h2), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h1), a, 0, 0, 0);
                    acc[rt] = a;
                }
                if (t == 0) stamp();
                continue;
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const u32x4 u = w[rt][t];
                v4f a = acc[rt];
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_lo(u[0]), a4.x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_hi(u[0]), a4.y, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_lo(u[1]), a4.z, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_hi(u[1]), a4.w, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_lo(u[2]), b4.x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_hi(u[2]), b4.y, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_lo(u[3]), b4.z, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_hi(u[3]), b4.w, a, 0, 0, 0);
                acc[rt] = a;
            }
        }
    }
    if (p.gamma) {
        ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);            // over the 4 k-slot groups of the wave
        if (g == 0) ssq[wave][c] = ss;
    }
    // split-K over the 4 waves, combined in fixed order; wave rt finishes row tile rt  [4 waves][RT][64 lanes][4]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float4*>(&red[((wave * RT + rt) * 64 + lane) * 4]) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
    stamp();
    __syncthreads();
    stamp();
    if (wave >= RT) return;
    const int rt = wave;
    float4 v = *reinterpret_cast<const float4*>(&red[(rt * 64 + lane) * 4]);
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
        const float4 o = *reinterpret_cast<const float4*>(&red[((ww * RT + rt) * 64 + lane) * 4]);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    if (p.gamma) {                                                      // statistics over the whole row: the waves' shares, fixed order
        float tot = (ssq[0][c] + ssq[1][c]) + (ssq[2][c] + ssq[3][c]);
        if constexpr (NW == 8) tot += (ssq[4][c] + ssq[5][c]) + (ssq[6][c] + ssq[7][c]);
        const float rstd = rsqrtf(tot / (float)p.K + p.eps);
        v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
    }
    const int n = n_base + rt * 16 + g * 4;                             // 4 consecutive weight rows of sequence c
    if (c >= p.nb || n >= p.N) return;
    if (p.mode == 1) {                                                  // (gate_j, up_j) row pairs -> two activations
        const float2 o = make_float2((v.x / (1.f + expf(-v.x))) * v.y, (v.z / (1.f + expf(-v.z))) * v.w);
        *reinterpret_cast<float2*>(p.y + (long long)c * p.ldy + (n >> 1)) = o;
    } else if (p.mode == 2) {
        *reinterpret_cast<float4*>(p.y + ((long long)ks * p.nb + c) * p.ldy + n) = v;
    } else if ((p.N & 3) == 0) {
        if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
        if (p.res) { const float4 r4 = *reinterpret_cast<const float4*>(p.res + (long long)c * p.ldres + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
        *reinterpret_cast<float4*>(p.y + (long long)c * p.ldy + n) = v;
    } else {                                                            // ragged N (the 6761-wide CosyVoice3 head): rows are not 16-byte aligned
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < p.N) {
                float o = e[i];
                if (p.bias) o += p.bias[n + i];
                if (p.res) o += p.res[(long long)c * p.ldres + n + i];
                p.y[(long long)c * p.ldy + n + i] = o;
            }
    }
    stamp();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 3: the same skinny GEMM with both operands fetched as WHOLE cache lines (skinny_pk_kernel).
//
// What bound skinny_mfma_kernel (tools/ubench/skinny_probe.hip, profiles/r3_skinny_probe.txt): not HBM - the launch takes as long with the weights
// L2-warm - but the number of cache-line requests a CU's vector memory pipe has to walk.  With row-major operands a lane's fragment is 16 bytes of ITS
// row, so one wave load touches 16 rows = 16 half-used lines for the weights and 16 more for each half of the activations: ~700 line requests per
// wave for 28 KB of payload, and the first k-tile of a workgroup was multiplied 8 900 clocks after entry.  Here
//   * the weights are read from a FRAGMENT-ORDERED copy (pack_frag_kernel, made once per matrix when the batch path is first used):
//     [row tile][k tile][lane][8 bf16] - one wave load = 1 KB contiguous, a wave's k range = one contiguous run;
//   * each wave fetches the activations of ITS k range row by row (one coalesced 16-byte-per-lane load per sequence: 896 B contiguous) into a wave-
//     private LDS slab and reads its B fragments from there; gamma likewise.  No workgroup barrier is involved (a wave's DS operations execute in
//     order; wave_lds_sync only pins the compiler), which is what made the round-2 LDS staging lose.
// Same products in the same order as skinny_mfma_kernel<.., X3 = true>: bit-identical output (tests/test_zz_llm_batch.py runs both).
// ---------------------------------------------------------------------------------------------------------------------------------
// W [N][K] row-major bf16 -> Wp [ceil(N / 16)][K / 32][64 lanes][8]: lane l = 16 g + c holds W[16 R + c][32 t + 8 g .. + 7] (rows >= N: zeros)
static __global__ __launch_bounds__(256) void pack_frag_kernel(const bf16_t* W, bf16_t* Wp, int N, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;           // one 16-byte piece each
    const int tilesK = K / 32;
    const long long pieces = (long long)((N + 15) / 16) * tilesK * 64;
    if (i >= pieces) return;
    const int lane = (int)(i & 63), g = lane >> 4, c = lane & 15;
    const long long frag = i >> 6;
    const int t = (int)(frag % tilesK), row = (int)(frag / tilesK) * 16 + c;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (row < N) v = *reinterpret_cast<const u32x4*>(W + (long long)row * K + t * 32 + g * 8);
    *reinterpret_cast<u32x4*>(Wp + i * 8) = v;
}

template <int RT, int KTW, int NW = 4>
__global__ __launch_bounds__(NW * 64) CV_WAVES_PER_EU(1, 2) void skinny_pk_kernel(SkinnyArgs p) {          // p.W = the fragment-ordered copy
    static_assert(RT >= 1 && RT <= 4 && RT <= NW && KTW <= 8, "one row tile per combining wave; a wave's k range is at most 256 columns (one 16-byte piece per lane)");
    constexpr int XP = KTW * 32 + 4;                                                  // floats per staged activation row (+4: rows start 16 bytes apart in the banks)
    __shared__ __attribute__((aligned(16))) float red[NW * RT * 256];
    __shared__ float ssq[NW][16];
    __shared__ __attribute__((aligned(16))) float xs[NW][16 * XP];
    __shared__ __attribute__((aligned(16))) float gs[NW][KTW * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ks = blockIdx.x % p.ksplit, rg = blockIdx.x / p.ksplit;
    const int tilesK = p.K / 32, tiles = tilesK / p.ksplit;
    const int t0 = wave * tiles / NW, t1 = (wave + 1) * tiles / NW, nt = t1 - t0;       // nt >= 1 (host check)
    const int kt0 = ks * tiles + t0;                                                  // first k tile of this wave in the whole K
    const int n_base = rg * RT * 16, row_tiles = (p.N + 15) / 16;
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();

    // ---- activations and gamma of the wave's k range FIRST (vmcnt retires in order: what is requested behind the weights waits for all of them)
    const int kx = kt0 * 32 + min(4 * lane, nt * 32 - 4);                             // lanes beyond the range re-read its last piece (not staged)
    // (16 named registers, not an array: with the scheduling fence below an array stays in scratch memory on this compiler)
#define CV_XROW(b) const float4 xr##b = *reinterpret_cast<const float4*>(p.x + (long long)min(b, p.nb - 1) * p.ldx + kx);   /* rows >= nb repeat the last sequence (never stored) */
    CV_XROW(0) CV_XROW(1) CV_XROW(2) CV_XROW(3) CV_XROW(4) CV_XROW(5) CV_XROW(6) CV_XROW(7) CV_XROW(8) CV_XROW(9) CV_XROW(10) CV_XROW(11) CV_XROW(12) CV_XROW(13) CV_XROW(14) CV_XROW(15)
#undef CV_XROW
    const float4 gr = *reinterpret_cast<const float4*>((p.gamma ? p.gamma : p.x) + kx);           // unconditional: a load in a branch splits the block the scheduler orders
    order_memory();                                                                   // keep these requests AHEAD of the weight stream (the scheduler would sink them behind it)
    // ---- the weight fragments: one contiguous run per row tile
    u32x4 w[RT][KTW];
    const bf16_t* wr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wr[rt] = p.W + (((long long)min(rg * RT + rt, row_tiles - 1) * tilesK + kt0) * 64 + lane) * 8;
#pragma unroll
    for (int t = 0; t < KTW; ++t) {                                                   // tile by tile, in the order the MFMA loop consumes them
        const bool ok = t < nt;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr[rt] + (ok ? t : 0) * 512));
            if (!ok) v = (u32x4){0u, 0u, 0u, 0u};
            w[rt][t] = v;
        }
    }
    // ---- stage the activations (the weights stay in flight)
    if (4 * lane < nt * 32) {
#define CV_XST(b) *reinterpret_cast<float4*>(&xs[wave][b * XP + 4 * lane]) = xr##b;
        CV_XST(0) CV_XST(1) CV_XST(2) CV_XST(3) CV_XST(4) CV_XST(5) CV_XST(6) CV_XST(7) CV_XST(8) CV_XST(9) CV_XST(10) CV_XST(11) CV_XST(12) CV_XST(13) CV_XST(14) CV_XST(15)
#undef CV_XST
        *reinterpret_cast<float4*>(&gs[wave][4 * lane]) = gr;
    }
    wave_lds_sync();

    float ss = 0.f;
    v4f acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KTW; ++t) {
        if (t < nt) {                                                   // wave-uniform
            float4 a4 = *reinterpret_cast<const float4*>(&xs[wave][c * XP + t * 32 + g * 8]);
            float4 b4 = *reinterpret_cast<const float4*>(&xs[wave][c * XP + t * 32 + g * 8 + 4]);
            if (p.gamma) {
                const float4 ga = *reinterpret_cast<const float4*>(&gs[wave][t * 32 + g * 8]), gb = *reinterpret_cast<const float4*>(&gs[wave][t * 32 + g * 8 + 4]);
                ss += a4.x * a4.x + a4.y * a4.y + a4.z * a4.z + a4.w * a4.w + b4.x * b4.x + b4.y * b4.y + b4.z * b4.z + b4.w * b4.w;
                a4.x *= ga.x; a4.y *= ga.y; a4.z *= ga.z; a4.w *= ga.w;
                b4.x *= gb.x; b4.y *= gb.y; b4.z *= gb.z; b4.w *= gb.w;
            }
            u32x4 h1, h2, h3;
            split3_bf16(a4, b4, h1, h2, h3);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {                           // smallest term first (as skinny_mfma_kernel)
                const v8bf wf = __builtin_bit_cast(v8bf, w[rt][t]);
                v4f a = acc[rt];
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h3), a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h2), a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h1), a, 0, 0, 0);
                acc[rt] = a;
            }
            if (t == 0) stamp();
        }
    }
    if (p.gamma) {
        ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);            // over the 4 k-slot groups of the wave
        if (g == 0) ssq[wave][c] = ss;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float4*>(&red[((wave * RT + rt) * 64 + lane) * 4]) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
    stamp();
    __syncthreads();
    stamp();
    if (wave >= RT) return;
    const int rt = wave;
    float4 v = *reinterpret_cast<const float4*>(&red[(rt * 64 + lane) * 4]);
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
        const float4 o = *reinterpret_cast<const float4*>(&red[((ww * RT + rt) * 64 + lane) * 4]);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    if (p.gamma) {
        float tot = (ssq[0][c] + ssq[1][c]) + (ssq[2][c] + ssq[3][c]);
        if constexpr (NW == 8) tot += (ssq[4][c] + ssq[5][c]) + (ssq[6][c] + ssq[7][c]);
        const float rstd = rsqrtf(tot / (float)p.K + p.eps);
        v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
    }
    const int n = n_base + rt * 16 + g * 4;
    if (c >= p.nb || n >= p.N) return;
    if (p.mode == 1) {
        const float2 o = make_float2((v.x / (1.f + expf(-v.x))) * v.y, (v.z / (1.f + expf(-v.z))) * v.w);
        *reinterpret_cast<float2*>(p.y + (long long)c * p.ldy + (n >> 1)) = o;
    } else if (p.mode == 2) {
        *reinterpret_cast<float4*>(p.y + ((long long)ks * p.nb + c) * p.ldy + n) = v;
    } else if ((p.N & 3) == 0) {
        if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
        if (p.res) { const float4 r4 = *reinterpret_cast<const float4*>(p.res + (long long)c * p.ldres + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
        *reinterpret_cast<float4*>(p.y + (long long)c * p.ldy + n) = v;
    } else {
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < p.N) {
                float o = e[i];
                if (p.bias) o += p.bias[n + i];
                if (p.res) o += p.res[(long long)c * p.ldres + n + i];
                p.y[(long long)c * p.ldy + n + i] = o;
            }
    }
    stamp();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// skinny_pk_kernel for 17 .. 32 sequences (round 4: 32 lock-step slots).  The MFMA takes 16 sequences as its B columns; the weight fragments a wave holds in
// registers are the expensive operand (streamed from HBM once per step), so a second COLUMN TILE reuses them: sequences 16 .. 31 are requested when the first
// sixteen have been parked in LDS (the same registers; their loads queue behind the weight stream and land about when it does), multiplied against the same
// w[][] and combined / stored by a second pass of the epilogue.  Per sequence the products, their order and the cross-wave order are those of
// skinny_pk_kernel: a sequence's result does not depend on its slot or on how many slots are in use (tests/test_zz_llm_batch.py).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int RT, int KTW, int NW = 4>
__global__ __launch_bounds__(NW * 64) CV_WAVES_PER_EU(1, 2) void skinny_pk2_kernel(SkinnyArgs p) {          // p.W = the fragment-ordered copy
    static_assert(RT >= 1 && RT <= 4 && RT <= NW && KTW <= 8, "one row tile per combining wave; a wave's k range is at most 256 columns (one 16-byte piece per lane)");
    constexpr int XP = KTW * 32 + 4;
    __shared__ __attribute__((aligned(16))) float red[NW * RT * 256];
    __shared__ float ssq[NW][16];
    __shared__ __attribute__((aligned(16))) float xs[NW][16 * XP];
    __shared__ __attribute__((aligned(16))) float gs[NW][KTW * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ks = blockIdx.x % p.ksplit, rg = blockIdx.x / p.ksplit;
    const int tilesK = p.K / 32, tiles = tilesK / p.ksplit;
    const int t0 = wave * tiles / NW, t1 = (wave + 1) * tiles / NW, nt = t1 - t0;       // nt >= 1 (host check)
    const int kt0 = ks * tiles + t0;
    const int n_base = rg * RT * 16, row_tiles = (p.N + 15) / 16;

    const int kx = kt0 * 32 + min(4 * lane, nt * 32 - 4);
    float4 xr0, xr1, xr2, xr3, xr4, xr5, xr6, xr7, xr8, xr9, xr10, xr11, xr12, xr13, xr14, xr15;
#define CV_XROW(b, base) xr##b = *reinterpret_cast<const float4*>(p.x + (long long)min((base) + b, p.nb - 1) * p.ldx + kx);   /* rows >= nb repeat the last sequence (never stored) */
#define CV_XROWS(base) CV_XROW(0, base) CV_XROW(1, base) CV_XROW(2, base) CV_XROW(3, base) CV_XROW(4, base) CV_XROW(5, base) CV_XROW(6, base) CV_XROW(7, base) \
                       CV_XROW(8, base) CV_XROW(9, base) CV_XROW(10, base) CV_XROW(11, base) CV_XROW(12, base) CV_XROW(13, base) CV_XROW(14, base) CV_XROW(15, base)
    CV_XROWS(0)
    const float4 gr = *reinterpret_cast<const float4*>((p.gamma ? p.gamma : p.x) + kx);
    order_memory();                                                                   // keep these requests AHEAD of the weight stream
    u32x4 w[RT][KTW];
    const bf16_t* wr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wr[rt] = p.W + (((long long)min(rg * RT + rt, row_tiles - 1) * tilesK + kt0) * 64 + lane) * 8;
#pragma unroll
    for (int t = 0; t < KTW; ++t) {
        const bool ok = t < nt;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr[rt] + (ok ? t : 0) * 512));
            if (!ok) v = (u32x4){0u, 0u, 0u, 0u};
            w[rt][t] = v;
        }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        if (ct * 16 >= p.nb) break;                                                   // (uniform)
        // ---- stage this column tile's activations (the weights stay in flight / in registers)
        if (4 * lane < nt * 32) {
#define CV_XST(b) *reinterpret_cast<float4*>(&xs[wave][b * XP + 4 * lane]) = xr##b;
            CV_XST(0) CV_XST(1) CV_XST(2) CV_XST(3) CV_XST(4) CV_XST(5) CV_XST(6) CV_XST(7) CV_XST(8) CV_XST(9) CV_XST(10) CV_XST(11) CV_XST(12) CV_XST(13) CV_XST(14) CV_XST(15)
#undef CV_XST
            if (ct == 0) *reinterpret_cast<float4*>(&gs[wave][4 * lane]) = gr;
        }
        wave_lds_sync();
        if (ct == 0 && p.nb > 16) { CV_XROWS(16) }                                    // the second column tile: requested behind the weight stream, lands about when it does

        float ss = 0.f;
        v4f acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KTW; ++t) {
            if (t < nt) {                                                   // wave-uniform
                float4 a4 = *reinterpret_cast<const float4*>(&xs[wave][c * XP + t * 32 + g * 8]);
                float4 b4 = *reinterpret_cast<const float4*>(&xs[wave][c * XP + t * 32 + g * 8 + 4]);
                if (p.gamma) {
                    const float4 ga = *reinterpret_cast<const float4*>(&gs[wave][t * 32 + g * 8]), gb = *reinterpret_cast<const float4*>(&gs[wave][t * 32 + g * 8 + 4]);
                    ss += a4.x * a4.x + a4.y * a4.y + a4.z * a4.z + a4.w * a4.w + b4.x * b4.x + b4.y * b4.y + b4.z * b4.z + b4.w * b4.w;
                    a4.x *= ga.x; a4.y *= ga.y; a4.z *= ga.z; a4.w *= ga.w;
                    b4.x *= gb.x; b4.y *= gb.y; b4.z *= gb.z; b4.w *= gb.w;
                }
                u32x4 h1, h2, h3;
                split3_bf16(a4, b4, h1, h2, h3);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {                           // smallest term first (as skinny_mfma_kernel)
                    const v8bf wf = __builtin_bit_cast(v8bf, w[rt][t]);
                    v4f a = acc[rt];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h3), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h2), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h1), a, 0, 0, 0);
                    acc[rt] = a;
                }
            }
        }
        if (p.gamma) {
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);            // over the 4 k-slot groups of the wave
            if (g == 0) ssq[wave][c] = ss;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float4*>(&red[((wave * RT + rt) * 64 + lane) * 4]) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
        __syncthreads();
        if (wave < RT) {
            const int rt = wave, seq = ct * 16 + c;
            float4 v = *reinterpret_cast<const float4*>(&red[(rt * 64 + lane) * 4]);
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) {
                const float4 o = *reinterpret_cast<const float4*>(&red[((ww * RT + rt) * 64 + lane) * 4]);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (p.gamma) {
                float tot = (ssq[0][c] + ssq[1][c]) + (ssq[2][c] + ssq[3][c]);
                if constexpr (NW == 8) tot += (ssq[4][c] + ssq[5][c]) + (ssq[6][c] + ssq[7][c]);
                const float rstd = rsqrtf(tot / (float)p.K + p.eps);
                v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
            }
            const int n = n_base + rt * 16 + g * 4;
            if (seq < p.nb && n < p.N) {
                if (p.mode == 1) {
                    const float2 o = make_float2((v.x / (1.f + expf(-v.x))) * v.y, (v.z / (1.f + expf(-v.z))) * v.w);
                    *reinterpret_cast<float2*>(p.y + (long long)seq * p.ldy + (n >> 1)) = o;
                } else if (p.mode == 2) {
                    *reinterpret_cast<float4*>(p.y + ((long long)ks * p.nb + seq) * p.ldy + n) = v;
                } else if ((p.N & 3) == 0) {
                    if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                    if (p.res) { const float4 r4 = *reinterpret_cast<const float4*>(p.res + (long long)seq * p.ldres + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                    *reinterpret_cast<float4*>(p.y + (long long)seq * p.ldy + n) = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (n + i < p.N) {
                            float o = e[i];
                            if (p.bias) o += p.bias[n + i];
                            if (p.res) o += p.res[(long long)seq * p.ldres + n + i];
                            p.y[(long long)seq * p.ldy + n + i] = o;
                        }
                }
            }
        }
        __syncthreads();                                                              // red / ssq / xs are reused by the second column tile
    }
#undef CV_XROWS
#undef CV_XROW
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same weight-stationary GEMM over MANY rows (round 3: the prefill of one prompt, M = 131 rows for the benchmark utterance).
// The tiled GEMM the prefill ran on (gemm_conv_kernel, AX3) restages a 32-row x 128-k tile seven times per output tile at M = 131: 22 us per
// launch for GEMMs whose weights a decode GEMV streams in 3-6 us.  Here a workgroup is what it is in skinny_pk_kernel - RT row tiles of the
// fragment-ordered weights, requested ONCE and kept in registers - and walks the rows of X in groups of 16 (the MFMA's B columns): group i + 1
// is requested while group i is multiplied, combined and stored.  What a workgroup ingests is its weight fragments plus ALL rows of X over its
// k range (M x K x 4 B, from L2); per output element the arithmetic (three-term split, k order, cross-wave order) is exactly the batched
// decode's, so a row's result does not depend on M or on its position.  Modes / epilogues as SkinnyArgs (nb = number of rows).
// MEASURED on the MI355X (profiles/r3_prefill_rows_ab.txt): 2.95 ms per 131-row prefill against 2.99 ms on the tiled GEMMs - the nine groups of a launch are a
// serial chain of LDS staging, MFMAs, cross-wave combine and two barriers (~2.4 us each), which is what the tiled kernel's seven k steps cost as well.  A second
// form (waves own different row tiles over the whole K, X shared through a double-buffered LDS tile, no cross-wave combine) measured 4.9-7.1 ms and was removed.
// Opt-in (option prefill_rows / CV_PREFILL_ROWS=1); the tiled path stays the default.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int RT, int KTW, int NW = 4>
__global__ __launch_bounds__(NW * 64) CV_WAVES_PER_EU(1, 2) void skinny_rows_kernel(SkinnyArgs p) {          // p.W = the fragment-ordered copy
    static_assert(RT >= 1 && RT <= 4 && RT <= NW && KTW <= 8, "one row tile per combining wave; a wave's k range is at most 256 columns (one 16-byte piece per lane)");
    constexpr int XP = KTW * 32 + 4;
    __shared__ __attribute__((aligned(16))) float red[NW * RT * 256];
    __shared__ float ssq[NW][16];
    __shared__ __attribute__((aligned(16))) float xs[NW][16 * XP];
    __shared__ __attribute__((aligned(16))) float gs[NW][KTW * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ks = blockIdx.x % p.ksplit, rg = blockIdx.x / p.ksplit;
    const int tilesK = p.K / 32, tiles = tilesK / p.ksplit;
    const int t0 = wave * tiles / NW, t1 = (wave + 1) * tiles / NW, nt = t1 - t0;       // nt >= 1 (host check)
    const int kt0 = ks * tiles + t0;
    const int n_base = rg * RT * 16, row_tiles = (p.N + 15) / 16;
    const int ngroups = (p.nb + 15) / 16;
    const int kx = kt0 * 32 + min(4 * lane, nt * 32 - 4);
    // (16 named registers, not an array: next to the scheduling fence an array stays in scratch memory on this compiler - as in skinny_pk_kernel)
    float4 xr0, xr1, xr2, xr3, xr4, xr5, xr6, xr7, xr8, xr9, xr10, xr11, xr12, xr13, xr14, xr15;
#define CV_XROW(b) xr##b = *reinterpret_cast<const float4*>(p.x + (long long)min(gq * 16 + b, p.nb - 1) * p.ldx + kx);   /* rows >= nb repeat the last row (never stored) */
#define CV_XLOAD_GROUP(G) { const int gq = (G); CV_XROW(0) CV_XROW(1) CV_XROW(2) CV_XROW(3) CV_XROW(4) CV_XROW(5) CV_XROW(6) CV_XROW(7) CV_XROW(8) CV_XROW(9) CV_XROW(10) CV_XROW(11) CV_XROW(12) CV_XROW(13) CV_XROW(14) CV_XROW(15) }
    CV_XLOAD_GROUP(0)
    const float4 gr = *reinterpret_cast<const float4*>((p.gamma ? p.gamma : p.x) + kx);
    order_memory();                                                                   // the first group and gamma AHEAD of the weight stream
    u32x4 w[RT][KTW];
    const bf16_t* wr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wr[rt] = p.W + (((long long)min(rg * RT + rt, row_tiles - 1) * tilesK + kt0) * 64 + lane) * 8;
#pragma unroll
    for (int t = 0; t < KTW; ++t) {
        const bool ok = t < nt;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            u32x4 v = *reinterpret_cast<const u32x4*>(wr[rt] + (ok ? t : 0) * 512);   // default cache policy: the other row-group workgroups of the launch do not share these, but a
            if (!ok) v = (u32x4){0u, 0u, 0u, 0u};                                     // second prompt's prefill right behind this one does
            w[rt][t] = v;
        }
    }
    if (4 * lane < nt * 32) *reinterpret_cast<float4*>(&gs[wave][4 * lane]) = gr;
    for (int grp = 0; grp < ngroups; ++grp) {
        if (4 * lane < nt * 32) {
#define CV_XST(b) *reinterpret_cast<float4*>(&xs[wave][b * XP + 4 * lane]) = xr##b;
            CV_XST(0) CV_XST(1) CV_XST(2) CV_XST(3) CV_XST(4) CV_XST(5) CV_XST(6) CV_XST(7) CV_XST(8) CV_XST(9) CV_XST(10) CV_XST(11) CV_XST(12) CV_XST(13) CV_XST(14) CV_XST(15)
#undef CV_XST
        }
        wave_lds_sync();
        if (grp + 1 < ngroups) CV_XLOAD_GROUP(grp + 1)                    // in flight under this group's MFMAs, combine and stores
        float ss = 0.f;
        v4f acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KTW; ++t) {
            if (t < nt) {                                                   // wave-uniform
                float4 a4 = *reinterpret_cast<const float4*>(&xs[wave][c * XP + t * 32 + g * 8]);
                float4 b4 = *reinterpret_cast<const float4*>(&xs[wave][c * XP + t * 32 + g * 8 + 4]);
                if (p.gamma) {
                    const float4 ga = *reinterpret_cast<const float4*>(&gs[wave][t * 32 + g * 8]), gb = *reinterpret_cast<const float4*>(&gs[wave][t * 32 + g * 8 + 4]);
                    ss += a4.x * a4.x + a4.y * a4.y + a4.z * a4.z + a4.w * a4.w + b4.x * b4.x + b4.y * b4.y + b4.z * b4.z + b4.w * b4.w;
                    a4.x *= ga.x; a4.y *= ga.y; a4.z *= ga.z; a4.w *= ga.w;
                    b4.x *= gb.x; b4.y *= gb.y; b4.z *= gb.z; b4.w *= gb.w;
                }
                u32x4 h1, h2, h3;
                split3_bf16(a4, b4, h1, h2, h3);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {                           // smallest term first (as skinny_mfma_kernel)
                    const v8bf wf = __builtin_bit_cast(v8bf, w[rt][t]);
                    v4f a = acc[rt];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h3), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h2), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h1), a, 0, 0, 0);
                    acc[rt] = a;
                }
            }
        }
        if (p.gamma) {
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
            if (g == 0) ssq[wave][c] = ss;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float4*>(&red[((wave * RT + rt) * 64 + lane) * 4]) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
        __syncthreads();
        if (wave < RT) {
            const int rt = wave;
            float4 v = *reinterpret_cast<const float4*>(&red[(rt * 64 + lane) * 4]);
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) {
                const float4 o = *reinterpret_cast<const float4*>(&red[((ww * RT + rt) * 64 + lane) * 4]);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (p.gamma) {
                const float tot = (ssq[0][c] + ssq[1][c]) + (ssq[2][c] + ssq[3][c]);
                const float rstd = rsqrtf(tot / (float)p.K + p.eps);
                v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
            }
            const int n = n_base + rt * 16 + g * 4;
            const int row = grp * 16 + c;
            if (row < p.nb && n < p.N) {
                if (p.mode == 1) {
                    const float2 o = make_float2((v.x / (1.f + expf(-v.x))) * v.y, (v.z / (1.f + expf(-v.z))) * v.w);
                    *reinterpret_cast<float2*>(p.y + (long long)row * p.ldy + (n >> 1)) = o;
                } else if (p.mode == 2) {
                    *reinterpret_cast<float4*>(p.y + ((long long)ks * p.nb + row) * p.ldy + n) = v;
                } else {                                                    // N % 4 == 0 (host check)
                    if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                    if (p.res) { const float4 r4 = *reinterpret_cast<const float4*>(p.res + (long long)row * p.ldres + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                    *reinterpret_cast<float4*>(p.y + (long long)row * p.ldy + n) = v;
                }
            }
        }
        __syncthreads();                                                     // red / ssq are rewritten by the next group
    }
}

#undef CV_XLOAD_GROUP
#undef CV_XROW

// ---------------------------------------------------------------------------------------------------------------------------------
// fp8 variant of the skinny GEMM (BASELINE.json configs[4]: "fp8 MFMA LLM path"; opt-in, batched decode only).  Weights: OCP e4m3 with one fp32
// scale per output row (quantised once at load, cosyvoice_amd/weights.py::quantize_fp8_rows).  Activations: quantised in the kernel, one
// scale per sequence and workgroup K range (absmax / 448), after the RMSNorm gamma.  Products on v_mfma_f32_16x16x32_fp8_fp8 (exact in fp32,
// fp32 accumulate): y[b][n] = acc * sx[b] * rstd[b] * sw[n].  Half the weight bytes of the bf16 kernel and 1/16 of its MFMA time; the price is
// the e4m3 rounding of both operands - there is no reference for this mode, it is held to an oracle that mirrors the
// same quantisation (tests/test_llm_fp8.py) and reported separately by bench.py.
// Lane l: weight row l % 16, sequence l % 16, k block g = l / 16: 16 consecutive k per 64-wide tile -> one 16-byte weight load (16 fp8) and
// four float4 X loads feed two MFMAs.
// ---------------------------------------------------------------------------------------------------------------------------------
struct SkinnyF8Args {
    const unsigned char* W; const float* wscale; const float* bias;
    const float* x; long long ldx; float* y; long long ldy; int N, K;
    const float* gamma; float eps; const float* res; long long ldres; int mode; int nb, ksplit;      // as SkinnyArgs
};

__device__ __forceinline__ long pack8_fp8(float4 a, float4 b, float inv) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a.x * inv, a.y * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a.z * inv, a.w * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(b.x * inv, b.y * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(b.z * inv, b.w * inv, hi, true);
    return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int RT, int KTW, bool GAMMA>
__global__ __launch_bounds__(256) void skinny_fp8_kernel(SkinnyF8Args p) {
    static_assert(RT >= 1 && RT <= 4, "the final combine hands one row tile to each wave");
    __shared__ __attribute__((aligned(16))) float red[4 * RT * 256];
    __shared__ float ssq[4][16];
    __shared__ float amx[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ks = blockIdx.x % p.ksplit, rg = blockIdx.x / p.ksplit;
    const int krange = p.K / p.ksplit, tiles = krange / 64;
    const int t0 = wave * tiles / 4, t1 = (wave + 1) * tiles / 4;
    const int kbase = ks * krange + t0 * 64 + g * 16;
    const int n_base = rg * RT * 16;

    u32x4 w[RT][KTW];
    float4 xv[KTW][4], gv[GAMMA ? KTW : 1][4];
    const unsigned char* wr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wr[rt] = p.W + (long long)min(n_base + rt * 16 + c, p.N - 1) * p.K + kbase;
    const float* xp = p.x + (long long)min(c, p.nb - 1) * p.ldx + kbase;
#pragma unroll
    for (int t = 0; t < KTW; ++t) {
        const bool ok = t0 + t < t1;
        const int tt = ok ? t : 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr[rt] + tt * 64));
            if (!ok) v = (u32x4){0u, 0u, 0u, 0u};
            w[rt][t] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[t][j] = *reinterpret_cast<const float4*>(xp + tt * 64 + 4 * j);
            if (!ok) xv[t][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (GAMMA) gv[t][j] = *reinterpret_cast<const float4*>(p.gamma + kbase + tt * 64 + 4 * j);
        }
    }
    // statistics of this workgroup's K range per sequence: sum of squares (RMSNorm: the range is the whole row when gamma is set) and absmax of
    // x * gamma (the fp8 scale)
    float ss = 0.f, am = 0.f;
#pragma unroll
    for (int t = 0; t < KTW; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 a = xv[t][j];
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
            if constexpr (GAMMA) { a.x *= gv[t][j].x; a.y *= gv[t][j].y; a.z *= gv[t][j].z; a.w *= gv[t][j].w; xv[t][j] = a; }
            am = fmaxf(am, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
    ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
    am = fmaxf(am, __shfl_xor(am, 16)); am = fmaxf(am, __shfl_xor(am, 32));
    if (g == 0) { ssq[wave][c] = ss; amx[wave][c] = am; }
    __syncthreads();
    const float amax = fmaxf(fmaxf(amx[0][c], amx[1][c]), fmaxf(amx[2][c], amx[3][c]));
    const float sx = amax > 0.f ? amax / 448.f : 1.f;          // e4m3: largest finite value 448
    const float inv = 1.f / sx;
    v4f acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KTW; ++t) {
        if (t0 + t < t1) {                                              // wave-uniform
            const long b0 = pack8_fp8(xv[t][0], xv[t][1], inv), b1 = pack8_fp8(xv[t][2], xv[t][3], inv);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const u32x4 u = w[rt][t];
                const long a0 = (long)(((unsigned long long)u[1] << 32) | u[0]), a1 = (long)(((unsigned long long)u[3] << 32) | u[2]);
                v4f a = acc[rt];
                a = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, a, 0, 0, 0);
                acc[rt] = a;
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float4*>(&red[((wave * RT + rt) * 64 + lane) * 4]) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
    __syncthreads();
    if (wave >= RT) return;
    const int rt = wave;
    float4 v = *reinterpret_cast<const float4*>(&red[(rt * 64 + lane) * 4]);
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) {
        const float4 o = *reinterpret_cast<const float4*>(&red[((ww * RT + rt) * 64 + lane) * 4]);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    const int n = n_base + rt * 16 + g * 4;
    if (c >= p.nb || n >= p.N) return;
    float sc = sx;
    if constexpr (GAMMA) {
        const float tot = (ssq[0][c] + ssq[1][c]) + (ssq[2][c] + ssq[3][c]);
        sc = sx * rsqrtf(tot / (float)p.K + p.eps);
    }
    const float e[4] = {v.x * sc * p.wscale[min(n, p.N - 1)], v.y * sc * p.wscale[min(n + 1, p.N - 1)], v.z * sc * p.wscale[min(n + 2, p.N - 1)],
                        v.w * sc * p.wscale[min(n + 3, p.N - 1)]};
    if (p.mode == 1) {
        const float2 o = make_float2((e[0] / (1.f + expf(-e[0]))) * e[1], (e[2] / (1.f + expf(-e[2]))) * e[3]);
        *reinterpret_cast<float2*>(p.y + (long long)c * p.ldy + (n >> 1)) = o;
    } else if (p.mode == 2) {
        *reinterpret_cast<float4*>(p.y + ((long long)ks * p.nb + c) * p.ldy + n) = make_float4(e[0], e[1], e[2], e[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < p.N) {
                float o = e[i];
                if (p.bias) o += p.bias[n + i];
                if (p.res) o += p.res[(long long)c * p.ldres + n + i];
                p.y[(long long)c * p.ldy + n + i] = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The DEEP skinny GEMM: the down projection (K = inter = 4864, N = hidden = 896) of the batched decode in ONE launch.
// Round 2 cut K 8 ways across workgroups (448 workgroups, raw partials) and combined them in a second launch (sum_partials_kernel):
// 5.7 + 4.8 us per layer, the second launch pure boundary + latency.  Here one workgroup of NW = 16 waves owns a 16-row tile over the
// WHOLE K: wave w takes k-tiles [w T / NW, (w + 1) T / NW) (9 or 10 of the 152), keeps a ring of D = 5 tiles in registers (weights
// non-temporal, activations from L2), and re-requests a slot the moment it has been multiplied - two memory round trips per wave instead of
// one plus a kernel boundary; the 16 accumulators meet in LDS in a fixed order and wave 0 adds the residual.  56 workgroups: the launch is
// bound by what 56 CUs ingest (155 KB of weights + nb x 19 KB of activations each).  MEASURED SLOWER (profiles/r3_batch_decode_ab.txt: 8-sequence
// step 975 -> 1086 us): an HBM weight stream reaches a CU at ~25 GB/s, so only the whole chip pulls 8.7 MB in a few microseconds.  Opt-in
// (CV_DOWN_DEEP=1), kept as the measured alternative.
// Arithmetic per sequence: the three-term split products of skinny_mfma_kernel, k-tiles summed in ascending order inside a wave, waves in
// ascending order: fixed, independent of the slot and of the other slots.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NW, int D>
__global__ __launch_bounds__(NW * 64) void skinny_deep_kernel(SkinnyArgs p) {
    __shared__ __attribute__((aligned(16))) float red[NW * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int tiles = p.K / 32;
    const int t0 = wave * tiles / NW, nt = (wave + 1) * tiles / NW - t0;          // <= 2 D (host check)
    const int kbase = t0 * 32 + g * 8;
    const int n_base = blockIdx.x * 16;
    const bf16_t* wr = p.W + (long long)min(n_base + c, p.N - 1) * p.K + kbase;
    const float* xp = p.x + (long long)min(c, p.nb - 1) * p.ldx + kbase;
    u32x4 w[D];
    float4 xa[D], xb[D];
    auto request = [&](int slot, int t) {                                         // tile t of this wave into ring slot `slot` (unconditional, clamped)
        const bool ok = t < nt;
        const int tt = ok ? t : 0;
        u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + tt * 32));
        float4 a = *reinterpret_cast<const float4*>(xp + tt * 32), b = *reinterpret_cast<const float4*>(xp + tt * 32 + 4);
        if (!ok) { v = (u32x4){0u, 0u, 0u, 0u}; a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
        w[slot] = v; xa[slot] = a; xb[slot] = b;
    };
#pragma unroll
    for (int t = 0; t < D; ++t) request(t, t);
    v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int t = 0; t < D; ++t) {
            if (r * D + t < nt) {                                               // wave-uniform
                u32x4 h1, h2, h3;
                split3_bf16(xa[t], xb[t], h1, h2, h3);
                const v8bf wf = __builtin_bit_cast(v8bf, w[t]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h3), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h2), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(v8bf, h1), acc, 0, 0, 0);
            }
            if (r == 0) request(t, D + t);
        }
    }
    *reinterpret_cast<float4*>(&red[(wave * 64 + lane) * 4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (wave != 0) return;
    float4 v = *reinterpret_cast<const float4*>(&red[lane * 4]);
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
        const float4 o = *reinterpret_cast<const float4*>(&red[(ww * 64 + lane) * 4]);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    const int n = n_base + g * 4;                                               // 4 consecutive weight rows of sequence c (N % 4 == 0: host check)
    if (c >= p.nb || n >= p.N) return;
    if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
    if (p.res) { const float4 r4 = *reinterpret_cast<const float4*>(p.res + (long long)c * p.ldres + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
    *reinterpret_cast<float4*>(p.y + (long long)c * p.ldy + n) = v;
}

// y[b][n] = res[b][n] + sum_ks part[ks][b][n]   (fixed order; the split-K combine of the down projection + residual)
static __global__ __launch_bounds__(256) void sum_partials_kernel(const float* part, int ksplit, int nb, int N, const float* res, long long ldres,
                                                               float* y, long long ldy) {
    const int i = blockIdx.x * 256 + threadIdx.x;                       // float4 index over [nb][N / 4]
    const int per = N >> 2;
    if (i >= nb * per) return;
    const int b = i / per, n = (i % per) * 4;
    // round 5: all the partials (and the residual) requested before the first add - with a run-time trip count every load of the loop was its own round trip
    // to the memory side (the partials were written by other XCDs): 5.2 us per launch for 1 MB.  Same order of additions: same bits.
    float4 o[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) o[s] = *reinterpret_cast<const float4*>(part + ((long long)min(s, ksplit - 1) * nb + b) * N + n);
    const float4 r = *reinterpret_cast<const float4*>((res ? res + (long long)b * ldres : part) + n);
    float4 v = o[0];
#pragma unroll
    for (int s = 1; s < 8; ++s) if (s < ksplit) { v.x += o[s].x; v.y += o[s].y; v.z += o[s].z; v.w += o[s].w; }
    for (int s = 8; s < ksplit; ++s) {                                  // (never taken by the models here: down_ksplit() <= 8)
        const float4 x = *reinterpret_cast<const float4*>(part + ((long long)s * nb + b) * N + n);
        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    if (res) { v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    *reinterpret_cast<float4*>(y + (long long)b * ldy + n) = v;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Decode attention of one new position per sequence: workgroup (head h, sequence b), NW waves over the keys, normalised output.
// With >= 8 sequences there are >= 112 (b, h) pairs, so the keys are not split across workgroups (the batch-1 kernel needs 8 slices
// per head to occupy the chip and leaves the merge to o_proj); rotate-half RoPE on q / new k in registers, KV append, online softmax
// per wave (12 slots x 4 key rows per pass and wave, the next pass prefetched), waves merged through LDS in fixed order.
// ---------------------------------------------------------------------------------------------------------------------------------
struct AttnDecodeBatchArgs {
    const float* qkv; long long ldqkv; float* kcache; float* vcache; long long cache_stride;      // per-sequence strides
    const float* rope_cos; const float* rope_sin; int heads, kv_heads, max_len;
    const DecodeState* st; float* out; long long ldo;
    int nb = 0;                                      // sequences of the launch (grid = heads * nb workgroups, 1-D)
    float* part = nullptr; int nslice = 1;           // attn_decode_batch_mfma_kernel: key slices per (sequence, kv head); > 1: un-normalised partials [pair][slice][head][ATTN_PART]
};

template <int NW>
static __global__ __launch_bounds__(NW * 64) void attn_decode_batch_kernel(AttnDecodeBatchArgs p) {
    constexpr int NS = 12, WPASS = 4 * NS, PASS = NW * WPASS;
    __shared__ __attribute__((aligned(16))) float pw[NW][ATTN_PART];
    // Workgroup -> (sequence b, head h).  The heads of one kv group read the SAME K / V rows; workgroup i runs on XCD i % 8 and the eight L2s are not coherent, so with
    // heads as the fast index the 7 heads of a group sat on 7 XCDs and every one of them pulled the group's K / V from HBM (round 4, rocprof of the 32-slot mixed
    // workload: 30 us per launch at contexts of ~600, the second largest kernel of the run).  Remapped: the heads of a (sequence, kv head) pair get ids that are equal
    // mod 8 - one XCD, one HBM fetch, six L2 hits.  Same arithmetic per (b, h): same bits.
    const int gsz = p.heads / p.kv_heads;
    int b, h;
    {
        const int id = blockIdx.x, np = p.nb * p.kv_heads;
        if ((np & 7) == 0) {
            const int x = id & 7, q = id >> 3, pair = (q / gsz) * 8 + x;
            b = pair / p.kv_heads; h = (pair % p.kv_heads) * gsz + q % gsz;
        } else { b = id / p.heads; h = id % p.heads; }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, grp = lane >> 4;
    const int g = h / gsz;
    const DecodeState* st = p.st + b;
    const float* qkv = p.qkv + (long long)b * p.ldqkv;
    float* kcache = p.kcache + (long long)b * p.cache_stride;
    float* vcache = p.vcache + (long long)b * p.cache_stride;
    const int pos = st->pos;                         // the new token sits at index `pos`
    const int L = pos + 1;
    const float* kc = kcache + (long long)g * p.max_len * 64;
    const float* vc = vcache + (long long)g * p.max_len * 64;
    float4 k4[NS], v4[NS], kn[NS], vn[NS];
    auto load_pass = [&](int base, float4* kd, float4* vd) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + wave * WPASS + sl * 4 + grp;
            const long long o = (long long)(j < pos ? j : 0) * 64 + sub * 4;      // unconditional, clamped
            kd[sl] = *reinterpret_cast<const float4*>(kc + o);
            vd[sl] = *reinterpret_cast<const float4*>(vc + o);
        }
    };
    load_pass(0, k4, v4);
    const float* qraw = qkv + h * 64;
    const float* kq = qkv + p.heads * 64 + g * 64;
    const float* vq = qkv + (p.heads + p.kv_heads) * 64 + g * 64;
    const int d0 = sub * 4, dp = (d0 + 32) & 63;
    const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + pos * 32 + (d0 & 31));
    const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + pos * 32 + (d0 & 31));
    const float4 qa = *reinterpret_cast<const float4*>(qraw + d0), qb = *reinterpret_cast<const float4*>(qraw + dp);
    const float4 ka = *reinterpret_cast<const float4*>(kq + d0), kp = *reinterpret_cast<const float4*>(kq + dp);
    const float4 vn4 = *reinterpret_cast<const float4*>(vq + d0);
    const float sg = d0 < 32 ? -1.f : 1.f;
    const float4 q4 = make_float4(qa.x * c4.x + sg * qb.x * s4.x, qa.y * c4.y + sg * qb.y * s4.y, qa.z * c4.z + sg * qb.z * s4.z, qa.w * c4.w + sg * qb.w * s4.w);
    const float4 kn4 = make_float4(ka.x * c4.x + sg * kp.x * s4.x, ka.y * c4.y + sg * kp.y * s4.y, ka.z * c4.z + sg * kp.z * s4.z, ka.w * c4.w + sg * kp.w * s4.w);
    if (!st->done && h % gsz == 0 && wave == 0 && grp == 0) {            // KV-cache append: one wave per kv head
        *reinterpret_cast<float4*>(kcache + ((long long)g * p.max_len + pos) * 64 + d0) = kn4;
        *reinterpret_cast<float4*>(vcache + ((long long)g * p.max_len + pos) * 64 + d0) = vn4;
    }
    const float NEG = -__builtin_huge_valf();
    float m_run = NEG, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < L; base += PASS) {     // workgroup-uniform
        const bool more = base + PASS < L;
        if (more) load_pass(base + PASS, kn, vn);     // longer contexts: the next pass is in flight under this pass's arithmetic
        float sc[NS];
        float mt = NEG;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + wave * WPASS + sl * 4 + grp;
            const float4 kk = (j == pos) ? kn4 : k4[sl];
            float a = q4.x * kk.x + q4.y * kk.y + q4.z * kk.z + q4.w * kk.w;
            a = group16_sum(a) * 0.125f;
            sc[sl] = j < L ? a : NEG;
            mt = fmaxf(mt, sc[sl]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
        if (mt != NEG) {                             // wave-uniform: false when this wave holds no key of the pass
            const float m_new = fmaxf(m_run, mt);
            const float scale = (m_run == NEG) ? 0.f : expf(m_run - m_new);
            acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
            float lt = 0.f;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const int j = base + wave * WPASS + sl * 4 + grp;
                const float e = (sc[sl] == NEG) ? 0.f : expf(sc[sl] - m_new);
                const float4 vv = (j == pos) ? vn4 : v4[sl];
                acc.x += e * vv.x; acc.y += e * vv.y; acc.z += e * vv.z; acc.w += e * vv.w;
                lt += e;
            }
            l_run = l_run * scale + lt;              // per-group partial (identical on the 16 lanes of a group)
            m_run = m_new;
        }
        if (more) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) { k4[sl] = kn[sl]; v4[sl] = vn[sl]; }
        }
    }
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    l_run += __shfl_xor(l_run, 16); l_run += __shfl_xor(l_run, 32);
    if (grp == 0) *reinterpret_cast<float4*>(&pw[wave][sub * 4]) = acc;
    if (lane == 0) { pw[wave][64] = (l_run > 0.f) ? m_run : 0.f; pw[wave][65] = l_run; }
    __syncthreads();
    if (tid < 16) {                                  // merge the waves (fixed order), normalise
        float M = NEG;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) if (pw[ww][65] > 0.f) M = fmaxf(M, pw[ww][64]);
        float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float wgt = (pw[ww][65] > 0.f) ? expf(pw[ww][64] - M) : 0.f;
            const float4 t = *reinterpret_cast<const float4*>(&pw[ww][tid * 4]);
            den += wgt * pw[ww][65];
            a.x += wgt * t.x; a.y += wgt * t.y; a.z += wgt * t.z; a.w += wgt * t.w;
        }
        const float inv = 1.f / den;
        *reinterpret_cast<float4*>(p.out + (long long)b * p.ldo + h * 64 + tid * 4) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}

// The same attention with the heads of a kv group in ONE workgroup (round 4): workgroup = (sequence b, kv head g), 4 key waves x HS head subsets (wave = hs * 4 + kw);
// a wave loads its key rows once and walks its subset's heads over them.  What it removes: in attn_decode_batch_kernel every one of the gsz query heads of a group
// pulls the group's K / V rows through its own CU (gsz x the L2 -> CU traffic; at 32 slots and contexts of ~600 keys that kernel was 24 - 30 us per launch, the
// second largest of the mixed workload).  Per head the arithmetic is that kernel's, slot for slot (same key -> (wave, slot, lane group) map, same online-softmax
// updates, same merge order): same bits.  No register double buffer for the next pass: the two head subsets of a key wave share a SIMD and cover each other's loads.
template <int HPW>                                   // heads per wave (subset size): gsz <= 2 * HPW
static __global__ __launch_bounds__(512) void attn_decode_batch_gqa_kernel(AttnDecodeBatchArgs p) {
    constexpr int NW = 4, NS = 12, WPASS = 4 * NS, PASS = NW * WPASS;
    __shared__ __attribute__((aligned(16))) float pw[NW][2 * HPW][ATTN_PART];
    const int gsz = p.heads / p.kv_heads;
    const int g = blockIdx.x % p.kv_heads, b = blockIdx.x / p.kv_heads;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wave = wv & 3, hs = wv >> 2, sub = lane & 15, grp = lane >> 4;
    const int h0 = hs * HPW, nh = min(HPW, gsz - h0);         // this wave's heads: g * gsz + h0 .. + nh - 1 (nh <= 0: a subset with nothing to do still joins the barrier)
    const DecodeState* st = p.st + b;
    const float* qkv = p.qkv + (long long)b * p.ldqkv;
    float* kcache = p.kcache + (long long)b * p.cache_stride;
    float* vcache = p.vcache + (long long)b * p.cache_stride;
    const int pos = st->pos;
    const int L = pos + 1;
    const float* kc = kcache + (long long)g * p.max_len * 64;
    const float* vc = vcache + (long long)g * p.max_len * 64;
    float4 k4[NS], v4[NS];
    auto load_pass = [&](int base) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + wave * WPASS + sl * 4 + grp;
            const long long o = (long long)(j < pos ? j : 0) * 64 + sub * 4;      // unconditional, clamped
            k4[sl] = *reinterpret_cast<const float4*>(kc + o);
            v4[sl] = *reinterpret_cast<const float4*>(vc + o);
        }
    };
    load_pass(0);
    const float* kq = qkv + p.heads * 64 + g * 64;
    const float* vq = qkv + (p.heads + p.kv_heads) * 64 + g * 64;
    const int d0 = sub * 4, dp = (d0 + 32) & 63;
    const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + pos * 32 + (d0 & 31));
    const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + pos * 32 + (d0 & 31));
    const float4 ka = *reinterpret_cast<const float4*>(kq + d0), kp = *reinterpret_cast<const float4*>(kq + dp);
    const float4 vn4 = *reinterpret_cast<const float4*>(vq + d0);
    const float sg = d0 < 32 ? -1.f : 1.f;
    const float4 kn4 = make_float4(ka.x * c4.x + sg * kp.x * s4.x, ka.y * c4.y + sg * kp.y * s4.y, ka.z * c4.z + sg * kp.z * s4.z, ka.w * c4.w + sg * kp.w * s4.w);
    float4 q4[HPW];
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        const float* qraw = qkv + (long long)(g * gsz + min(h0 + i, gsz - 1)) * 64;       // (clamped: unused entries of a short subset read a valid head)
        const float4 qa = *reinterpret_cast<const float4*>(qraw + d0), qb = *reinterpret_cast<const float4*>(qraw + dp);
        q4[i] = make_float4(qa.x * c4.x + sg * qb.x * s4.x, qa.y * c4.y + sg * qb.y * s4.y, qa.z * c4.z + sg * qb.z * s4.z, qa.w * c4.w + sg * qb.w * s4.w);
    }
    if (!st->done && wv == 0 && grp == 0) {                  // KV-cache append: one wave per workgroup
        *reinterpret_cast<float4*>(kcache + ((long long)g * p.max_len + pos) * 64 + d0) = kn4;
        *reinterpret_cast<float4*>(vcache + ((long long)g * p.max_len + pos) * 64 + d0) = vn4;
    }
    const float NEG = -__builtin_huge_valf();
    float m_run[HPW], l_run[HPW];
    float4 acc[HPW];
#pragma unroll
    for (int i = 0; i < HPW; ++i) { m_run[i] = NEG; l_run[i] = 0.f; acc[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
    for (int base = 0; base < L; base += PASS) {     // workgroup-uniform
        if (base > 0) load_pass(base);
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            if (i >= nh) break;                      // wave-uniform
            float sc[NS];
            float mt = NEG;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const int j = base + wave * WPASS + sl * 4 + grp;
                const float4 kk = (j == pos) ? kn4 : k4[sl];
                float a = q4[i].x * kk.x + q4[i].y * kk.y + q4[i].z * kk.z + q4[i].w * kk.w;
                a = group16_sum(a) * 0.125f;
                sc[sl] = j < L ? a : NEG;
                mt = fmaxf(mt, sc[sl]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
            if (mt != NEG) {                         // wave-uniform: false when this wave holds no key of the pass
                const float m_new = fmaxf(m_run[i], mt);
                const float scale = (m_run[i] == NEG) ? 0.f : expf(m_run[i] - m_new);
                acc[i].x *= scale; acc[i].y *= scale; acc[i].z *= scale; acc[i].w *= scale;
                float lt = 0.f;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const int j = base + wave * WPASS + sl * 4 + grp;
                    const float e = (sc[sl] == NEG) ? 0.f : expf(sc[sl] - m_new);
                    const float4 vv = (j == pos) ? vn4 : v4[sl];
                    acc[i].x += e * vv.x; acc[i].y += e * vv.y; acc[i].z += e * vv.z; acc[i].w += e * vv.w;
                    lt += e;
                }
                l_run[i] = l_run[i] * scale + lt;
                m_run[i] = m_new;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        if (i >= nh) break;
        float4 a = acc[i]; float l = l_run[i];
        a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
        a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
        l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
        if (grp == 0) *reinterpret_cast<float4*>(&pw[wave][h0 + i][sub * 4]) = a;
        if (lane == 0) { pw[wave][h0 + i][64] = (l > 0.f) ? m_run[i] : 0.f; pw[wave][h0 + i][65] = l; }
    }
    __syncthreads();
    if (tid < 16 * gsz) {                            // merge the key waves of every head (fixed order), normalise
        const int hh = tid >> 4, t = tid & 15;
        float M = NEG;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) if (pw[ww][hh][65] > 0.f) M = fmaxf(M, pw[ww][hh][64]);
        float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float wgt = (pw[ww][hh][65] > 0.f) ? expf(pw[ww][hh][64] - M) : 0.f;
            const float4 tq = *reinterpret_cast<const float4*>(&pw[ww][hh][t * 4]);
            den += wgt * pw[ww][hh][65];
            a.x += wgt * tq.x; a.y += wgt * tq.y; a.z += wgt * tq.z; a.w += wgt * tq.w;
        }
        const float inv = 1.f / den;
        *reinterpret_cast<float4*>(p.out + (long long)b * p.ldo + (g * gsz + hh) * 64 + t * 4) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same attention on the fp32 matrix pipe, K / V of a (sequence, kv head) pair read ONCE for all the heads of the group.
// The per-head kernels above spend ~40 VALU instructions per (key slot of 4 keys, head) on 16-lane dot products, DPP reductions and softmax bookkeeping, once per
// query head, and every head's workgroup pulls the group's rows through its own CU (rocprof, 32 slots, contexts of ~600 keys: 24 us per launch, 41 % of the
// lock-step step, 0.8 TB/s).  Here the group's gsz <= 16 query heads are the 16 COLUMNS of v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation):
//   S^T[key][head] = K[key][:] . Q^T[:][head]     16 MFMAs per tile of 16 keys (A = K rows as loaded, B = the RoPE'd, pre-scaled queries held in registers)
//   O^T[d][head]  += V^T[d][key] . P[key][head]   16 MFMAs per tile (A = V rows as loaded, B = P straight from the S^T accumulators: S^T's C layout IS the B layout)
// The reduction index of an MFMA may be permuted freely as long as A and B agree, and so may the output rows - so both operands are plain float4 loads of the
// cache rows (lane (r = l % 16, q = l / 16): K[key0 + r][16 i + 4 q ..+3], V[key0 + 4 q + i][4 r ..+3], i = 0..3) and nothing goes through LDS before the merge.
// Per lane the softmax is 4 scores per tile for ONE head (column r), with the running max made uniform over the 4 lane groups by two shuffles.
// Workgroup = (sequence, kv head, key slice); its NW waves and the nslice slices deal the 16-key tiles round-robin (tile t -> wave t % (nslice * NW)), two tiles
// per wave and round with the next round's rows in flight.  Waves merge through LDS in fixed order; nslice > 1 leaves un-normalised partials that
// attn_merge_batch_kernel combines (a second launch is the cheap way to make partials visible across XCDs - see the header of this file).
// Arithmetic differs from attn_decode_batch_kernel in summation order only (fp32 rounding): tokens are held to the oracle's by the same tests.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NW>
static __global__ __launch_bounds__(NW * 64) void attn_decode_batch_mfma_kernel(AttnDecodeBatchArgs p) {
    constexpr int RT = 2;                            // key tiles per wave and round
    __shared__ __attribute__((aligned(16))) float pw[NW][16][ATTN_PART];
    const int S = p.nslice, W = S * NW;
    const int slice = blockIdx.x % S, pair = blockIdx.x / S, g = pair % p.kv_heads, b = pair / p.kv_heads;
    const int gsz = p.heads / p.kv_heads;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                          // scalar: the tile loop and its tests are wave-uniform branches, the loads stay counted
    const DecodeState* st = p.st + b;
    const float* qkv = p.qkv + (long long)b * p.ldqkv;
    float* kcache = p.kcache + (long long)b * p.cache_stride;
    float* vcache = p.vcache + (long long)b * p.cache_stride;
    const int pos = __builtin_amdgcn_readfirstlane(st->pos);                            // the new token sits at index `pos`
    const int L = pos + 1, nt = (L + 15) >> 4, tpos = pos >> 4;
    const float* kc = kcache + (long long)g * p.max_len * 64;
    const float* vc = vcache + (long long)g * p.max_len * 64;
    const int wi = slice * NW + wave;                // this wave's place in the round-robin over the key tiles
    auto load_round = [&](int t0, float4 (*kf)[4], float4 (*vf)[4]) {
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int key0 = (t0 + u * W) * 16;      // tiles >= nt re-read row 0 (clamped below; never used)
            const int jk = key0 + r;
            const float* kr = kc + (long long)(jk < pos ? jk : 0) * 64 + 4 * q;
#pragma unroll
            for (int i = 0; i < 4; ++i) kf[u][i] = *reinterpret_cast<const float4*>(kr + 16 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int jv = key0 + 4 * q + i;
                vf[u][i] = *reinterpret_cast<const float4*>(vc + (long long)(jv < pos ? jv : 0) * 64 + 4 * r);
            }
        }
    };
    float4 kA[RT][4], vA[RT][4], kB[RT][4], vB[RT][4];
    load_round(wi, kA, vA);                          // unconditional (tiles beyond the context re-read row 0): a load in a branch costs the compiler its vmcnt count
    // rotate-half RoPE in registers, in the operand layouts: lane (r, q) holds dims 16 i + 4 q .. + 3 (i = 0..3) of query head r (B operand, pre-scaled by 1/8)
    // and of the new key (substituted for the A operand of key row `pos`, which is not in the cache yet)
    const float* kq = qkv + p.heads * 64 + g * 64;
    const float* vq = qkv + (p.heads + p.kv_heads) * 64 + g * 64;
    const float* qraw = qkv + (long long)(g * gsz + min(r, gsz - 1)) * 64;
    float4 qf[4], kn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d0 = 16 * i + 4 * q, dp = (d0 + 32) & 63;
        const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + pos * 32 + (d0 & 31));
        const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + pos * 32 + (d0 & 31));
        const float4 qa = *reinterpret_cast<const float4*>(qraw + d0), qb = *reinterpret_cast<const float4*>(qraw + dp);
        const float4 ka = *reinterpret_cast<const float4*>(kq + d0), kp = *reinterpret_cast<const float4*>(kq + dp);
        const float sg = d0 < 32 ? -1.f : 1.f;
        const float qs = r < gsz ? 0.125f : 0.f;     // columns beyond the group: zero queries (scores 0, never stored)
        qf[i] = make_float4((qa.x * c4.x + sg * qb.x * s4.x) * qs, (qa.y * c4.y + sg * qb.y * s4.y) * qs, (qa.z * c4.z + sg * qb.z * s4.z) * qs, (qa.w * c4.w + sg * qb.w * s4.w) * qs);
        kn[i] = make_float4(ka.x * c4.x + sg * kp.x * s4.x, ka.y * c4.y + sg * kp.y * s4.y, ka.z * c4.z + sg * kp.z * s4.z, ka.w * c4.w + sg * kp.w * s4.w);
    }
    const float4 vn4 = *reinterpret_cast<const float4*>(vq + 4 * r);
    if (!st->done && slice == 0 && wave == 0) {      // KV-cache append: one wave per (sequence, kv head)
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(kcache + ((long long)g * p.max_len + pos) * 64 + 16 * i + 4 * q) = kn[i];
        }
        if (q == 0) *reinterpret_cast<float4*>(vcache + ((long long)g * p.max_len + pos) * 64 + 4 * r) = vn4;
    }
    const float NEG = -__builtin_huge_valf();
    float m_run = NEG, l_run = 0.f;
    v4f o[4];                                        // O^T tiles: o[c][ii] = numerator of dim 16 q + 4 ii + c, head r
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int t0 = wi; t0 < nt; t0 += RT * W) {       // wave-uniform
        load_round(t0 + RT * W, kB, vB);             // the next round's rows are in flight under this round's arithmetic (unconditional, clamped)
        v4f sc[RT];
        float mt = NEG;
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int t = t0 + u * W;
            if (t < nt) {                            // wave-uniform
                if (t == tpos) {                     // the tile that holds the new key: its row comes from registers
                    const bool mine = (t * 16 + r) == pos;
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (mine) kA[u][i] = kn[i];
                }
                v4f c = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(kA[u][i].x, qf[i].x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(kA[u][i].y, qf[i].y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(kA[u][i].z, qf[i].z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(kA[u][i].w, qf[i].w, c, 0, 0, 0);
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {     // C layout: row (key) 4 q + ii, column (head) r
                    const float a = (t * 16 + 4 * q + ii) < L ? c[ii] : NEG;
                    c[ii] = a; mt = fmaxf(mt, a);
                }
                sc[u] = c;
            } else sc[u] = (v4f){NEG, NEG, NEG, NEG};
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));        // over the 4 lane groups: the running max of head r is the same in all of them
        const float m_new = fmaxf(m_run, mt);        // finite: tile t0 holds at least one key
        const float scale = (m_run == NEG) ? 0.f : expf(m_run - m_new);
#pragma unroll
        for (int c = 0; c < 4; ++c) { o[c][0] *= scale; o[c][1] *= scale; o[c][2] *= scale; o[c][3] *= scale; }
        float lt = 0.f;
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int t = t0 + u * W;
            if (t < nt) {                            // wave-uniform
                v4f e;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) { e[ii] = (sc[u][ii] == NEG) ? 0.f : expf(sc[u][ii] - m_new); lt += e[ii]; }
                if (t == tpos) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if ((t * 16 + 4 * q + i) == pos) vA[u][i] = vn4;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {        // step i: reduction slot q <-> key 4 q + i (P of that key is accumulator register i of this lane)
                    o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[u][i].x, e[i], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[u][i].y, e[i], o[1], 0, 0, 0);
                    o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[u][i].z, e[i], o[2], 0, 0, 0);
                    o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[u][i].w, e[i], o[3], 0, 0, 0);
                }
            }
        }
        l_run = l_run * scale + lt;                  // per-lane partial (this lane's keys of head r)
        m_run = m_new;
#pragma unroll
        for (int u = 0; u < RT; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { kA[u][i] = kB[u][i]; vA[u][i] = vB[u][i]; }
        }
    }
    l_run += __shfl_xor(l_run, 16); l_run += __shfl_xor(l_run, 32);
    if (r < gsz) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) *reinterpret_cast<float4*>(&pw[wave][r][16 * q + 4 * ii]) = make_float4(o[0][ii], o[1][ii], o[2][ii], o[3][ii]);
        if (q == 0) { pw[wave][r][64] = (l_run > 0.f) ? m_run : 0.f; pw[wave][r][65] = l_run; }
    }
    __syncthreads();
    for (int e = tid; e < 16 * gsz; e += NW * 64) {  // merge the waves of every head (fixed order)
        const int hh = e >> 4, t = e & 15;
        float M = NEG;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) if (pw[ww][hh][65] > 0.f) M = fmaxf(M, pw[ww][hh][64]);
        float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float wgt = (pw[ww][hh][65] > 0.f) ? expf(pw[ww][hh][64] - M) : 0.f;
            const float4 tq = *reinterpret_cast<const float4*>(&pw[ww][hh][t * 4]);
            den += wgt * pw[ww][hh][65];
            a.x += wgt * tq.x; a.y += wgt * tq.y; a.z += wgt * tq.z; a.w += wgt * tq.w;
        }
        if (S == 1) {
            const float inv = 1.f / den;
            *reinterpret_cast<float4*>(p.out + (long long)b * p.ldo + (g * gsz + hh) * 64 + t * 4) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
        } else {
            float* pr = p.part + (((long long)pair * S + slice) * gsz + hh) * ATTN_PART;
            *reinterpret_cast<float4*>(pr + t * 4) = a;
            if (t == 0) { pr[64] = (den > 0.f) ? M : 0.f; pr[65] = den; }
        }
    }
}

// out[b][head][:] = softmax-merge of the nslice partials a (sequence, kv head) pair's workgroups left (fixed order; slices without a key carry denominator 0)
static __global__ __launch_bounds__(256) void attn_merge_batch_kernel(AttnDecodeBatchArgs p) {
    constexpr int MAXS = 8;
    const int e = blockIdx.x * 256 + threadIdx.x;    // one thread per (sequence, head, 4 dims)
    const int gsz = p.heads / p.kv_heads, S = p.nslice;
    if (e >= p.nb * p.heads * 16) return;
    const int t = e & 15, h = (e >> 4) % p.heads, b = (e >> 4) / p.heads, g = h / gsz, hh = h % gsz;
    const float* pr = p.part + (((long long)(b * p.kv_heads + g) * S) * gsz + hh) * ATTN_PART;
    float4 pa[MAXS]; float2 ml[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {                 // every partial requested before the first is used (unconditional, clamped)
        const float* ps = pr + (long long)min(s, S - 1) * gsz * ATTN_PART;
        pa[s] = *reinterpret_cast<const float4*>(ps + t * 4);
        ml[s] = *reinterpret_cast<const float2*>(ps + 64);
    }
    const float NEG = -__builtin_huge_valf();
    float M = NEG;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) if (s < S && ml[s].y > 0.f) M = fmaxf(M, ml[s].x);
    float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        const float wgt = (s < S && ml[s].y > 0.f) ? expf(ml[s].x - M) : 0.f;
        den += wgt * ml[s].y;
        a.x += wgt * pa[s].x; a.y += wgt * pa[s].y; a.z += wgt * pa[s].z; a.w += wgt * pa[s].w;
    }
    const float inv = 1.f / den;
    *reinterpret_cast<float4*>(p.out + (long long)b * p.ldo + h * 64 + t * 4) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
}

// advance the KV length of every slot after a backbone step
static __global__ void advance_pos_batch_kernel(DecodeState* st, int nb) {
    const int b = threadIdx.x;
    if (b < nb && !st[b].done) st[b].pos += 1;
}

}  // namespace cv
