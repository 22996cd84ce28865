// The row-local part of a flow-estimator transformer block for LARGE passes as ONE launch per 64-row band (round 5; matcha BasicTransformerBlock inside
// CausalConditionalDecoder, cosyvoice/flow/decoder.py:405-494), bf16 mode:
//
//     attention output (bf16) -> out-projection + bias + residual -> LayerNorm(norm3) -> FF1 + bias + GELU(erf) -> FF2 + bias + residual
//       [-> LayerNorm(norm1 of the NEXT block) -> bf16 rows: the operand of the next block's QKV GEMM]
//
// flow_big.h runs this as five launches (out-projection, LayerNorm, FF1, FF2, LayerNorm) of 64 x 64 tiles with K = 256 .. 1024: 70 of a block's 142 us at
// M = 10 784 (8 utterances per pass), each launch a chain of first-load wait -> a few K stages -> stores, each writing and re-reading 11 - 22 MB of activations
// (profiles/r4_flow_phase_stamps.txt: "nothing to pipeline inside one GEMM").  flow_tail.h (round 3) showed the other extreme: a 16-row band per workgroup pulls all
// 2 MB of weights through one CU's load path for 16 rows of MFMA work - 48 us, bound by the L2 -> L1 path (674 bands x 2 MB).  Here a band is 64 rows (4 MFMA row
// tiles): every weight fragment a wave loads (16 bytes per lane straight into the MFMA operand, fragment-ordered stream, weights.py::pack_flow_band) is multiplied
// against FOUR activation fragments read from LDS - 4 x the rows per weight byte, ~170 workgroups = one round over the chip at M = 10 784, 1.3 MB of weights per band.
//
// LDS (147 KB, one workgroup of 8 waves per CU): X1 fp32 residual tile 64 x C (read once, written once), A1 / A2 bf16 operand tiles, the attention tile A0 overlaid
// on A1 + A2 (dead after the out-projection).  FF1 -> FF2 run in FF / C chunks of C hidden columns: the chunk's GELU output (64 x C bf16) is the only part of
// the 64 x FF intermediate that ever exists, FF2's accumulators stay in registers across the chunks - k ascending over the chunks, so every output element sums the
// same products in the same order as the five-launch form (and the four small-tile launches): BIT-IDENTICAL to them, tests/test_flow_big.py.
// Weight loads run one PASS (<= 16 fragments per wave) ahead of the MFMAs in a second register buffer, across phase boundaries and barriers (the stream does not
// depend on activations); there is no global store before the last fragment has been consumed (a pending store costs the compiler its vmcnt bookkeeping).
#pragma once
#include <type_traits>
#include "flow_fused.h"

namespace cv {

template <int I, int N, class F>
__device__ __forceinline__ void band_static_for(F&& f) {          // f(integral_constant<int, I>) for I .. N - 1: the loop index as a compile-time constant (buffer choice per pass)
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); band_static_for<I + 1, N>(f); }
}

struct FlowBandArgs {
    const bf16_t* att; int ld_att;            // attention output [M][INNER] bf16
    float* x; int ldx;                        // residual stream [M][C] fp32, read once, written once
    const u32x4_t* wstream;                   // packed fragments [NW waves][fragments per wave][64 lanes] (pack_flow_band)
    const float* prm;                         // small operands of the block (the layout of flow_tail.h): [b_out C | norm3.g C | norm3.b C | b_ff1 FF | b_ff2 C | norm1.g of the NEXT block C | its norm1.b C]
    float eps; int M;
    bf16_t* xn; int ld_xn;                    // HAS_NEXT: LayerNorm(norm1 of the next block) of the new residual rows, bf16 [M][C]
    bf16_t* qk; int ld_qk;                    // HAS_QKV: Q | K of the NEXT block [M][2 INNER] bf16 (what its flow_gemm_big_kernel<.., 0> launch would have written)
    bf16_t* vt; long long vt_batch; int ldt; int rows_per_batch;      // HAS_QKV: its V^T [B][INNER][ldt] bf16, key-permuted (vt_col)
    long long* dbg;                           // dev tool (tools/ubench/band_probe.hip): clock64() of thread 0 at the phase boundaries, 16 slots per workgroup; null in production
    int stagger;                              // dev probe (option "band_stagger", round 6): band b starts (b mod 4) * stagger * ~0.9 us late (s_sleep), so that the phases of the
                                              // bands of one round - the staging loads, the Q | K | V^T store burst - do not coincide; experiments builds only (measured, a loss: profiles/r6_band_stagger.txt)
};

// one pass: PT 16-column tiles x KS k-steps of 32 against the 4 row tiles of the band.  A: bf16 pairs in LDS, row pitch `pitch` dwords, first dword k0.
// SWAP: the activations as the MFMA "A" operand - a lane ends with 4 consecutive ROWS of one column (the V^T epilogue) instead of 4 consecutive columns of one row.
// KS0: first k-step (the pipelined form multiplies a pass in slices of k-steps between the pieces of its GELU).
template <int PT, int KS, int MODE = 0, int RT = 4, bool SWAP = false, int KS0 = 0>
__device__ __forceinline__ void band_mma(const u32x4_t (&w)[16], const unsigned* A, int pitch, int k0, int lq, int lg, v4f (&acc)[RT][PT]) {
    if constexpr (MODE == 2) {                // probe: consume the fragments without the matrix pipe or LDS
#pragma unroll
        for (int i = PT * KS0; i < PT * KS; ++i) acc[0][0][0] += __uint_as_float(w[i][0] ^ w[i][1] ^ w[i][2] ^ w[i][3]);
        return;
    }
#pragma unroll
    for (int ks = KS0; ks < KS; ++ks) {
        uint4 af[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if constexpr (MODE == 3) af[rt] = make_uint4(0x3f803f80u + rt + ks, 0x3f003f00u + lq, 0x3e803e80u + lg, 0x3f803f80u);      // probe: no fragment reads
            else af[rt] = *reinterpret_cast<const uint4*>(&A[(16 * rt + lq) * pitch + k0 + ks * 16 + lg * 4]);
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (SWAP) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af[rt]), __builtin_bit_cast(v8bf, w[ks * PT + t]), acc[rt][t], 0, 0, 0);
                else                acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, w[ks * PT + t]), __builtin_bit_cast(v8bf, af[rt]), acc[rt][t], 0, 0, 0);
            }
        }
    }
}
template <int N, int MODE = 0>
__device__ __forceinline__ void band_wload(u32x4_t (&dst)[16], const u32x4_t* ws, int frag0) {
    if constexpr (MODE == 1) return;          // probe: the stream is not requested again
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = ws[(long long)(frag0 + i) * 64];
    __builtin_amdgcn_sched_barrier(0);        // the requests stay HERE, a pass ahead of their use (left alone the scheduler sinks them down to the first MFMA that needs them)
}

// LayerNorm of the band's rows parked in X (fp32, pitch PX floats) -> bf16 operand tile (pitch `pa` dwords): the expressions of ln_bf16_kernel / tail_layernorm
template <int C, int PX, int NT, int BM = 64>
__device__ __forceinline__ void band_layernorm(const float* X, unsigned* A, int pa, const float* gamma, const float* beta, float eps, int tid) {
    constexpr int NJ = C / 64;
    const int sub = tid & 15;
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += NT / 16) {
        const int row = r0 + (tid >> 4);
        if ((NT / 16 > BM || BM % (NT / 16) != 0) && row >= BM) break;      // (more 16-lane groups than rows: the 32-row band on 8 waves normalises its rows in one pass with half the groups; 48 rows: a pass and a half)
        float4 v[NJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { v[j] = *reinterpret_cast<const float4*>(&X[row * PX + 4 * sub + 64 * j]); s += v[j].x + v[j].y + v[j].z + v[j].w; }
        const float mean = group16_sum(s) * (1.f / (float)C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean; q += a * a + b * b + c * c + d * d; }
        const float rstd = rsqrtf(group16_sum(q) * (1.f / (float)C) + eps);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = 4 * sub + 64 * j;
            const float4 g = *reinterpret_cast<const float4*>(gamma + k), b = *reinterpret_cast<const float4*>(beta + k);      // gamma / beta: LDS copies
            const float4 y = make_float4((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y, (v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w);
            *reinterpret_cast<uint2*>(&A[row * pa + k / 2]) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
        }
    }
}

// fragments per wave of one block's stream (weights.py::pack_flow_band must agree)
template <int C, int INNER, int FF, int NW>
struct FlowBandShape {
    static constexpr int TA = C / 16 / NW;                          // 16-column tiles per wave in every phase (all three GEMMs are cut to C output columns per pass)
    static constexpr int KA = INNER / 32;                           // out-projection k-steps, in passes of <= 8
    static constexpr int NPA = (KA + 7) / 8;
    static constexpr int KC = C / 32;                               // FF1 chunk: K = C;  FF2 chunk: K = C hidden columns
    static constexpr int NCH = FF / C;
    static constexpr int TOTAL = TA * (KA + 2 * NCH * KC);
    static constexpr int NQ = 3 * INNER / C;                        // HAS_QKV: Q | K | V of the next block in passes of C output columns (one pass = TA tiles x KC k-steps, like an FF1 chunk)
    static constexpr int FQ0 = TOTAL, TOTALQ = TOTAL + NQ * TA * KC;
    static_assert((2 * INNER) % C == 0 && NQ >= 2, "flow_band: a QKV pass may not straddle the K | V boundary");
    static_assert(C % (16 * NW) == 0 && TA >= 1 && TA <= 2 && KC <= 8 && INNER % 32 == 0 && FF % C == 0 && (KA <= 8 || KA % 8 == 0), "flow_band: unsupported dimensions");
};

// MODE (dev tool only, tools/ubench/band_probe.hip; 4 = HAS_QKV without the Q | K | V^T stores, tools/ubench/bandq_probe.hip): 0 = the kernel; 1 = the weight stream is requested once (prologue) and never again; 2 = no MFMA and no fragment
// reads (the stream alone, consumed by a register checksum); 3 = MFMAs on register operands (no LDS fragment reads)
// BM = rows per band: 64 (4 MFMA row tiles per weight fragment; large passes) or 32 (2 row tiles: twice the workgroups for passes whose 64-row bands would leave most of
// the chip idle - 4 utterances per pass, the shared chunk passes of the streaming scheduler - at half the LDS, two workgroups per CU).  Same arithmetic per element.
// HAS_QKV (implies HAS_NEXT; the stream then carries the next block's QKV fragments behind the FF ones, weights.py::pack_flow_band(.., w_qkv_next)): the band also runs
// the next block's QKV GEMM on its LayerNorm rows - Q | K rows and the V^T columns go straight from the accumulators to memory (the values and stores of
// flow_gemm_big_kernel<.., OMODE 0>'s direct epilogue), so a block of a large pass is TWO launches (attention, band) and the bf16 LayerNorm rows never exist in memory.
// PIPE (second session of round 5; bands of at most 48 rows - the second GELU tile does not fit next to a 64-row band): the FF1 -> GELU -> FF2 chunks as a software
// pipeline.  Stage j multiplies FF2 of chunk j - 1 and FF1 of chunk j + 1 in slices of k-steps BETWEEN the pieces of chunk j's GELU, so the matrix pipe works under the
// epilogue's VALU instructions (the rolled form runs them one after the other: 96 MFMAs = 1.5 k cycles per wave, then ~390 VALU instructions = 2.1 k), and a chunk costs
// ONE barrier instead of two: the GELU tiles alternate between two LDS buffers.  Same products in the same order per output element: the same bits.
template <int C, int INNER, int FF, bool HAS_NEXT, int NW, int MODE = 0, int BM = 64, bool HAS_QKV = false, bool PIPE = false>
__global__ __launch_bounds__(NW * 64) void flow_band_kernel(FlowBandArgs p) {
    using S = FlowBandShape<C, INNER, FF, NW>;
    constexpr int NT = NW * 64, TA = S::TA, RT = BM / 16;
    static_assert(BM == 64 || BM == 48 || BM == 32, "flow_band: 64-, 48- or 32-row bands");
    static_assert(!HAS_QKV || HAS_NEXT, "flow_band: the QKV phase belongs to the next block");
    constexpr int PA0 = INNER / 2 + LDS_PAD, PX = C + LDS_PAD, PA1 = C / 2 + LDS_PAD;       // LDS row pitches (dwords / floats): 8 mod 16 (common.h, LDS_PAD)
    constexpr int OPS = (BM * PA0 > 2 * BM * PA1) ? BM * PA0 : 2 * BM * PA1;                // the attention tile overlays A1 | A2
    __shared__ __attribute__((aligned(16))) float X1[BM * PX];
    __shared__ __attribute__((aligned(16))) unsigned OP[OPS];
    unsigned* const A0 = OP; unsigned* const A1 = OP; unsigned* const A2 = OP + BM * PA1;
    static_assert(!PIPE || (BM <= 48 && S::NCH >= 2), "flow_band: the pipelined chunks need a second GELU tile in LDS (bands of at most 48 rows)");
    __shared__ __attribute__((aligned(16))) unsigned A2B[PIPE ? BM * PA1 : 4];      // PIPE: the GELU tile of the odd chunks
    constexpr int O_BOUT = 0, O_G3 = C, O_BE3 = 2 * C, O_BFF1 = 3 * C, O_BFF2 = 3 * C + FF, O_G1N = 4 * C + FF, O_BE1N = 5 * C + FF, NPRM = 6 * C + FF;
    __shared__ __attribute__((aligned(16))) float prm[NPRM];
    const int tid = threadIdx.x, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
#ifdef CV_BUILD_EXPERIMENTS
    if (p.stagger) for (int i = (blockIdx.x & 3) * p.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(32);       // dev probe: 32 x 64 clocks per unit
#endif

    // ---- the band's small operands, its attention tile and its residual rows FIRST (unconditional, clamped), then the first pass of the weight stream
    constexpr int NPV = NPRM / 4, PPT = (NPV + NT - 1) / NT;
    v4f pv[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) pv[i] = *reinterpret_cast<const v4f*>(p.prm + 4 * min(tid + NT * i, NPV - 1));
    constexpr int APC = BM * INNER / 8, APT = (APC + NT - 1) / NT;          // 16-byte pieces of the attention tile per thread
    u32x4_t av[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int v = min(tid + NT * i, APC - 1), r = v / (INNER / 8), c = v % (INNER / 8);
        av[i] = *reinterpret_cast<const u32x4_t*>(p.att + (long long)min(m0 + r, p.M - 1) * p.ld_att + c * 8);
    }
    constexpr int XPC = BM * C / 4, XPT = (XPC + NT - 1) / NT;              // float4 pieces of the residual tile per thread
    // (native vectors, not the float4 struct: as float4 these staging arrays stayed allocas - parked in spare LDS where there was some, in SCRATCH otherwise: 6 - 8
    // scratch stores and loads per lane in the staging phase of every band, found in the second session of round 5 through the pipelined form, which leaves no spare LDS)
    using XV = v4f;
    XV xv[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int v = min(tid + NT * i, XPC - 1), r = v / (C / 4), c = v % (C / 4);
        xv[i] = *reinterpret_cast<const XV*>(p.x + (long long)min(m0 + r, p.M - 1) * p.ldx + c * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    const u32x4_t* ws = p.wstream + (long long)wave * (HAS_QKV ? S::TOTALQ : S::TOTAL) * 64 + lane;
    u32x4_t wb0[16], wb1[16];
    constexpr int FA0 = TA * (S::KA < 8 ? S::KA : 8);                       // fragments of the first out-projection pass
    band_wload<FA0>(wb0, ws, 0);
    if constexpr (MODE == 1) band_wload<16>(wb1, ws, 16);
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 16 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();                                                                // 0: requests issued
#pragma unroll
    for (int i = 0; i < PPT; ++i) { const int v = tid + NT * i; if (v < NPV) *reinterpret_cast<v4f*>(&prm[4 * v]) = pv[i]; }
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int v = tid + NT * i;
        if (v < APC) *reinterpret_cast<u32x4_t*>(&A0[(v / (INNER / 8)) * PA0 + (v % (INNER / 8)) * 4]) = av[i];
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int v = tid + NT * i;
        if (v < XPC) *reinterpret_cast<XV*>(&X1[(v / (C / 4)) * PX + (v % (C / 4)) * 4]) = xv[i];
    }
    __syncthreads();
    stamp();                                                                // 1: operands staged

    // ---- A: out-projection + bias + residual -> X1 (fp32).  Wave w owns the 16-column tiles w + NW t.
    constexpr int FCD = TA * S::KC;                                         // fragments of an FF1 / FF2 chunk pass
    {
        v4f acc[RT][TA];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < TA; ++t) acc[rt][t] = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (S::NPA == 1) {
            band_wload<FCD, MODE>(wb1, ws, TA * S::KA);                           // FF1 chunk 0 in flight under the out-projection
            band_mma<TA, S::KA, MODE, RT>(wb0, A0, PA0, 0, lq, lg, acc);
        } else {
            static_assert(S::NPA <= 2, "flow_band: the out-projection runs in at most two passes (INNER <= 512)");
            band_wload<TA * 8, MODE>(wb1, ws, TA * 8);
            band_mma<TA, 8, MODE, RT>(wb0, A0, PA0, 0, lq, lg, acc);
            band_wload<FCD, MODE>(wb0, ws, TA * S::KA);                           // FF1 chunk 0
            band_mma<TA, 8, MODE, RT>(wb1, A0, PA0, 8 * 16, lq, lg, acc);
        }
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const int n = 16 * (wave + NW * t) + 4 * lg;
            const float4 b = *reinterpret_cast<const float4*>(&prm[O_BOUT + n]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float* xr = &X1[(16 * rt + lq) * PX + n];
                const float4 r = *reinterpret_cast<const float4*>(xr);
                *reinterpret_cast<float4*>(xr) = make_float4(acc[rt][t][0] + b.x + r.x, acc[rt][t][1] + b.y + r.y, acc[rt][t][2] + b.z + r.z, acc[rt][t][3] + b.w + r.w);   // the same lane read this element
            }
        }
    }
    __syncthreads();                                                        // X1 complete, A0 dead
    stamp();                                                                // 2: out-projection done
    // ---- B: LayerNorm(norm3) -> A1 (bf16)
    band_layernorm<C, PX, NT, BM>(X1, A1, PA1, &prm[O_G3], &prm[O_BE3], p.eps, tid);
    __syncthreads();
    stamp();                                                                // 3: LayerNorm done
    // ---- C / D: FF1 chunk j (+ bias + GELU -> A2) and FF2 over that chunk's hidden columns, accumulators across the chunks.  Stream order: FF1_0 FF2_0 FF1_1 FF2_1 ..
    // With NPA == 1 chunk 0's FF1 fragments sit in wb1, otherwise in wb0: the two buffers alternate from there.
    v4f acc2[RT][TA];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < TA; ++t) acc2[rt][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    constexpr int F0 = TA * S::KA;                                          // first fragment of the FF stream
    constexpr bool FF1_IN_WB1 = S::NPA == 1;
    if constexpr (PIPE) {
        // wbF holds the FF1 passes (chunk 0 is in it: loaded under the out-projection), wbG the FF2 passes; a pass is requested as soon as its buffer has been multiplied
        auto& wbF = FF1_IN_WB1 ? wb1 : wb0;
        auto& wbG = FF1_IN_WB1 ? wb0 : wb1;
        constexpr int NU = RT * TA;                                         // GELU pieces of a chunk per wave: one 16 x 16 accumulator tile each
        constexpr int KC = S::KC;
        v4f acc1[2][RT][TA];
        auto zero1 = [&](int b) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < TA; ++t) acc1[b][rt][t] = (v4f){0.f, 0.f, 0.f, 0.f};
        };
        band_wload<FCD, MODE>(wbG, ws, F0 + FCD);                           // FF2 chunk 0
        zero1(0);
        band_mma<TA, KC, MODE, RT>(wbF, A1, PA1, 0, lq, lg, acc1[0]);      // FF1 chunk 0
        band_wload<FCD, MODE>(wbF, ws, F0 + 2 * FCD);                       // FF1 chunk 1
        band_static_for<0, S::NCH>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr bool HAS_F2 = j >= 1, HAS_F1 = j + 1 < S::NCH;        // FF2 of chunk j - 1 (its GELU tile was completed by the last barrier), FF1 of chunk j + 1
            constexpr int NSTEP = (HAS_F2 ? KC : 0) + (HAS_F1 ? KC : 0);    // k-steps of this stage: FF2 first, then FF1
            unsigned* const Gw = (j & 1) ? A2B : A2;                        // this chunk's GELU tile
            const unsigned* const Gr = (j & 1) ? A2 : A2B;                  // the previous chunk's
            if constexpr (HAS_F1) zero1((j + 1) & 1);
            band_static_for<0, NU>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int s0 = NSTEP * u / NU, s1 = NSTEP * (u + 1) / NU;       // this piece's slice of the stage's k-steps
                band_static_for<s0, s1>([&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    if constexpr (HAS_F2 && st < KC) {
                        band_mma<TA, st + 1, MODE, RT, false, st>(wbG, Gr, PA1, 0, lq, lg, acc2);
                        if constexpr (st == KC - 1) {                       // wbG is free: FF2 of chunk j (or, after the last one, QKV pass 1)
                            if constexpr (j < S::NCH) band_wload<FCD, MODE>(wbG, ws, F0 + (2 * j + 1) * FCD);
                        }
                    } else {
                        constexpr int k = st - (HAS_F2 ? KC : 0);
                        band_mma<TA, k + 1, MODE, RT, false, k>(wbF, A1, PA1, 0, lq, lg, acc1[(j + 1) & 1]);
                        if constexpr (k == KC - 1) {                        // wbF is free: FF1 of chunk j + 2, or (HAS_QKV) the first QKV pass
                            if constexpr (j + 2 < S::NCH) band_wload<FCD, MODE>(wbF, ws, F0 + 2 * (j + 2) * FCD);
                            else if constexpr (HAS_QKV) band_wload<FCD, MODE>(wbF, ws, S::FQ0);
                        }
                    }
                });
                __builtin_amdgcn_sched_barrier(0);                          // the slices stay between the pieces (left alone the scheduler gathers the MFMAs in front of the epilogue again)
                {
                    constexpr int t = u / RT, rt = u % RT;
                    const int n = 16 * (wave + NW * t) + 4 * lg;            // column inside the chunk
                    const float4 b = *reinterpret_cast<const float4*>(&prm[O_BFF1 + j * C + n]);
                    const v4f a = acc1[j & 1][rt][t];
                    const float4 y = gelu_erf4(make_float4(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w));
                    *reinterpret_cast<uint2*>(&Gw[(16 * rt + lq) * PA1 + n / 2]) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();                                                // chunk j's GELU tile is complete; the other tile may be overwritten by the next stage
            stamp();
        });
        // FF2 of the last chunk; then (HAS_QKV) QKV pass 1 into wbG
        band_mma<TA, KC, MODE, RT>(wbG, ((S::NCH - 1) & 1) ? A2B : A2, PA1, 0, lq, lg, acc2);
        if constexpr (HAS_QKV) band_wload<FCD, MODE>(wbG, ws, S::FQ0 + FCD);
    } else {
    #pragma unroll 1
        for (int j = 0; j < S::NCH; ++j) {                                      // a real loop: every chunk runs the same code on the same two buffers (and the epilogue's GELU stays one copy)
            v4f acc1[RT][TA];
    #pragma unroll
            for (int rt = 0; rt < RT; ++rt)
    #pragma unroll
                for (int t = 0; t < TA; ++t) acc1[rt][t] = (v4f){0.f, 0.f, 0.f, 0.f};
            // FF1 chunk j is in buffer X (loaded one pass ago); request FF2 chunk j into the other buffer, multiply
            if constexpr (FF1_IN_WB1) { band_wload<FCD, MODE>(wb0, ws, F0 + (2 * j + 1) * FCD); band_mma<TA, S::KC, MODE, RT>(wb1, A1, PA1, 0, lq, lg, acc1); }
            else                      { band_wload<FCD, MODE>(wb1, ws, F0 + (2 * j + 1) * FCD); band_mma<TA, S::KC, MODE, RT>(wb0, A1, PA1, 0, lq, lg, acc1); }
    #pragma unroll
            for (int t = 0; t < TA; ++t) {
                const int n = 16 * (wave + NW * t) + 4 * lg;                    // column inside the chunk
                const float4 b = *reinterpret_cast<const float4*>(&prm[O_BFF1 + j * C + n]);
    #pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float4 y = gelu_erf4(make_float4(acc1[rt][t][0] + b.x, acc1[rt][t][1] + b.y, acc1[rt][t][2] + b.z, acc1[rt][t][3] + b.w));      // = apply_act4(ACT_GELU_ERF, ..), inlined
                    *reinterpret_cast<uint2*>(&A2[(16 * rt + lq) * PA1 + n / 2]) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
                }
            }
            __syncthreads();                                                    // the chunk's hidden tile is complete
            // FF2 chunk j; request FF1 chunk j + 1 (unconditional: after the last chunk the request repeats that chunk's FF1 fragments and is never used)
            // (HAS_QKV: after the last chunk the request is the first QKV pass of the next block instead)
            const int nxt = (HAS_QKV && j == S::NCH - 1) ? S::FQ0 : F0 + 2 * min(j + 1, S::NCH - 1) * FCD;
            if constexpr (FF1_IN_WB1) { band_wload<FCD, MODE>(wb1, ws, nxt); band_mma<TA, S::KC, MODE, RT>(wb0, A2, PA1, 0, lq, lg, acc2); }
            else                      { band_wload<FCD, MODE>(wb0, ws, nxt); band_mma<TA, S::KC, MODE, RT>(wb1, A2, PA1, 0, lq, lg, acc2); }
            __syncthreads();                                                    // before the next chunk overwrites A2 (and before the epilogue below touches X1's neighbours)
            stamp();                                                            // 4 .. 3 + NCH: chunk j done
        }
        if constexpr (HAS_QKV) {                                                // QKV pass 1 into the buffer the last FF2 chunk has just left (pass 0 sits in the other one)
            if constexpr (FF1_IN_WB1) band_wload<FCD, MODE>(wb0, ws, S::FQ0 + FCD); else band_wload<FCD, MODE>(wb1, ws, S::FQ0 + FCD);
        }
    }
    // ---- FF2 epilogue: + bias + residual -> X1 (the new residual stream)
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const int n = 16 * (wave + NW * t) + 4 * lg;
        const float4 b = *reinterpret_cast<const float4*>(&prm[O_BFF2 + n]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float* xr = &X1[(16 * rt + lq) * PX + n];
            const float4 r = *reinterpret_cast<const float4*>(xr);
            *reinterpret_cast<float4*>(xr) = make_float4(acc2[rt][t][0] + b.x + r.x, acc2[rt][t][1] + b.y + r.y, acc2[rt][t][2] + b.z + r.z, acc2[rt][t][3] + b.w + r.w);
        }
    }
    __syncthreads();
    if constexpr (HAS_NEXT) {
        // ---- E: LayerNorm(norm1 of the next block) -> A1, written out as the bf16 operand rows of its QKV GEMM
        band_layernorm<C, PX, NT, BM>(X1, A1, PA1, &prm[O_G1N], &prm[O_BE1N], p.eps, tid);
        __syncthreads();
        if constexpr (!HAS_QKV) {
            constexpr int NP = BM * C / 8;
#pragma unroll
            for (int i = 0; i < (NP + NT - 1) / NT; ++i) {
                const int v = tid + NT * i, r = v / (C / 8), c = v % (C / 8);
                if (v < NP && m0 + r < p.M) *reinterpret_cast<u32x4_t*>(p.xn + (long long)(m0 + r) * p.ld_xn + c * 8) = *reinterpret_cast<const u32x4_t*>(&A1[r * PA1 + c * 4]);
            }
        } else {
            // ---- F: Q | K | V of the next block, NQ passes of C output columns over K = C.  No barrier from here on (A1 is only read, nothing goes through LDS), so loads
            // and stores stay counted: pass c multiplies, requests pass c + 2 into the buffer it has just emptied, THEN stores - the wait of pass c + 1 is for requests
            // older than these stores.  Q | K: 4 columns of a row per lane (8 bytes); V: operands swapped, 4 consecutive rows (keys) of a column per lane - the V^T
            // stores of flow_gemm_big_kernel (an aligned key group of one request = one 8-byte store at vt_col).
            stamp();                                                        // (HAS_QKV) FF2 epilogue + next LayerNorm done
            float qsink = 0.f;
            band_static_for<0, S::NQ>([&](auto ic) {
                constexpr int c = decltype(ic)::value;
                constexpr bool IS_V = c * C >= 2 * INNER;
                constexpr bool IN_WB1 = (c % 2 == 0) == FF1_IN_WB1;
                v4f acc[RT][TA];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < TA; ++t) acc[rt][t] = (v4f){0.f, 0.f, 0.f, 0.f};
                if constexpr (IN_WB1) band_mma<TA, S::KC, MODE, RT, IS_V>(wb1, A1, PA1, 0, lq, lg, acc); else band_mma<TA, S::KC, MODE, RT, IS_V>(wb0, A1, PA1, 0, lq, lg, acc);
                if constexpr (c + 2 < S::NQ) { if constexpr (IN_WB1) band_wload<FCD, MODE>(wb1, ws, S::FQ0 + (c + 2) * FCD); else band_wload<FCD, MODE>(wb0, ws, S::FQ0 + (c + 2) * FCD); }
                if constexpr (MODE == 4) {                   // probe: the QKV products without their stores (a register checksum keeps them)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int t = 0; t < TA; ++t) qsink += acc[rt][t][0] + acc[rt][t][1] + acc[rt][t][2] + acc[rt][t][3];
                } else if constexpr (!IS_V && TA == 2) {
                    // two tiles per wave: the stream pairs them so that a lane's 4 + 4 columns are ADJACENT (weights.py::pack_flow_band: MFMA row 4 g + r of tile t is output
                    // column 32 wave + 8 g + 4 t + r of the pass) - one 16-byte store per row tile instead of two 8-byte ones, 64 contiguous bytes per row and instruction.
                    // The epilogue is bound by the NUMBER of store instructions (cdna_hip_programming.md T21), not by their bytes.  Same dot products: same values.
                    const int n = c * C + 32 * wave + 8 * lg;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int m = m0 + 16 * rt + lq;
                        if (m < p.M) *reinterpret_cast<uint4*>(p.qk + (long long)m * p.ld_qk + n) =
                            make_uint4(pack_bf16x2(acc[rt][0][0], acc[rt][0][1]), pack_bf16x2(acc[rt][0][2], acc[rt][0][3]), pack_bf16x2(acc[rt][1][0], acc[rt][1][1]), pack_bf16x2(acc[rt][1][2], acc[rt][1][3]));
                    }
                } else if constexpr (!IS_V) {
#pragma unroll
                    for (int t = 0; t < TA; ++t) {
                        const int n = c * C + 16 * (wave + NW * t) + 4 * lg;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const int m = m0 + 16 * rt + lq;
                            if (m < p.M) *reinterpret_cast<uint2*>(p.qk + (long long)m * p.ld_qk + n) = make_uint2(pack_bf16x2(acc[rt][t][0], acc[rt][t][1]), pack_bf16x2(acc[rt][t][2], acc[rt][t][3]));
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < TA; ++t) {
                        const int n = c * C - 2 * INNER + 16 * (wave + NW * t) + lq;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const int m = m0 + 16 * rt + 4 * lg;
                            const unsigned lo = pack_bf16x2(acc[rt][t][0] + 0.f, acc[rt][t][1] + 0.f), hi = pack_bf16x2(acc[rt][t][2] + 0.f, acc[rt][t][3] + 0.f);      // "+ 0": flow_gemm_big_kernel adds its (absent) bias here
                            if (m >= p.M) continue;
                            const int b = m / p.rows_per_batch, tt = m - b * p.rows_per_batch;
                            bf16_t* row = p.vt + (long long)b * p.vt_batch + (long long)n * p.ldt;
                            if (m + 3 < p.M && tt + 3 < p.rows_per_batch && (tt & 3) == 0) { *reinterpret_cast<uint2*>(row + vt_col(tt)) = make_uint2(lo, hi); continue; }
                            if (m + 3 < p.M && tt + 3 < p.rows_per_batch && (tt & 1) == 0) {
                                *reinterpret_cast<unsigned*>(row + vt_col(tt)) = lo; *reinterpret_cast<unsigned*>(row + vt_col(tt + 2)) = hi; continue;
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int mr = m + r;
                                if (mr >= p.M) continue;
                                const int br = mr / p.rows_per_batch, tr_ = mr - br * p.rows_per_batch;
                                const unsigned u = r < 2 ? lo : hi;
                                p.vt[(long long)br * p.vt_batch + (long long)n * p.ldt + vt_col(tr_)] = (bf16_t)((r & 1) ? (u >> 16) : (u & 0xffffu));
                            }
                        }
                    }
                }
            });
            if constexpr (MODE == 4) { if (qsink == 12345.678f) p.qk[0] = (bf16_t)0; }
        }
    }
    stamp();                                                                // FF2 epilogue (+ next LayerNorm and its rows) done
    // write-out of the residual stream: whole rows, 16 bytes per lane
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int v = tid + NT * i, r = v / (C / 4), c = v % (C / 4);
        if (v < XPC && m0 + r < p.M) *reinterpret_cast<float4*>(p.x + (long long)(m0 + r) * p.ldx + c * 4) = *reinterpret_cast<const float4*>(&X1[r * PX + c * 4]);
    }
    stamp();
}


// ---------------------------------------------------------------------------------------------------------------------------------
// flow_lnqkv_kernel (round 6): LayerNorm(norm1) + the QKV GEMM of a stage's FIRST transformer block as one launch per row band.  Inside a stage the band launch of
// block j runs block j + 1's QKV (HAS_QKV above); the first block of a stage follows the stage's resnet and had ln_bf16_kernel + flow_gemm_big_kernel<.., 0> to itself
// (5.3 + 31.7 us at 8 utterances of U10 against 16 - 18 us for the same GEMM as a band phase: the LayerNorm rows stay in LDS, the weights arrive fragment-ordered).
// The arithmetic is the band's phase F - the same LayerNorm expressions on the same rows, the same MFMA operands in the same k order, the same stores - so the bits are
// those of the two launches it replaces.  Stream: weights.py::pack_flow_band_qkv (the QKV part of a `bandq` stream on its own).
// ---------------------------------------------------------------------------------------------------------------------------------
struct FlowLnQkvArgs {
    const float* x; int ldx;                  // residual stream [M][C] fp32 (read only)
    const u32x4_t* wstream;                   // [NW waves][NQ passes x TA tiles x KC k-steps][64 lanes]
    const float* gamma; const float* beta; float eps; int M;
    bf16_t* qk; int ld_qk;                    // Q | K [M][2 INNER] bf16
    bf16_t* vt; long long vt_batch; int ldt; int rows_per_batch;      // V^T [B][INNER][ldt] bf16, key-permuted (vt_col)
};

template <int C, int INNER, int NW, int BM>
__global__ __launch_bounds__(NW * 64) void flow_lnqkv_kernel(FlowLnQkvArgs p) {
    using S = FlowBandShape<C, INNER, 4 * C, NW>;
    constexpr int NT = NW * 64, TA = S::TA, RT = BM / 16, PA1 = C / 2 + LDS_PAD, NJ = C / 64, FCD = TA * S::KC;
    static_assert(BM == 64 || BM == 48 || BM == 32, "flow_lnqkv: 64-, 48- or 32-row bands");
    __shared__ __attribute__((aligned(16))) unsigned A1[BM * PA1];
    const int tid = threadIdx.x, lane = tid & 63, lq = lane & 15, lg = lane >> 4, sub = tid & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    // the band's rows first (a 16-lane group owns a row: the register layout of band_layernorm / ln_bf16_kernel), then the first two passes of the weight stream
    constexpr int RP = (BM + NT / 16 - 1) / (NT / 16);              // rows per 16-lane group
    v4f xr[RP][NJ];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int row = i * (NT / 16) + (tid >> 4);
        const float* src = p.x + (long long)min(m0 + min(row, BM - 1), p.M - 1) * p.ldx;
#pragma unroll
        for (int j = 0; j < NJ; ++j) xr[i][j] = *reinterpret_cast<const v4f*>(src + 4 * sub + 64 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
    const u32x4_t* ws = p.wstream + (long long)wave * (S::NQ * FCD) * 64 + lane;
    u32x4_t wb0[16], wb1[16];
    band_wload<FCD>(wb0, ws, 0);
    band_wload<FCD>(wb1, ws, FCD);
    // ---- LayerNorm -> A1 (bf16): the expressions of band_layernorm, operation for operation
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int row = i * (NT / 16) + (tid >> 4);
        if (row >= BM) break;
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) sm += xr[i][j][0] + xr[i][j][1] + xr[i][j][2] + xr[i][j][3];
        const float mean = group16_sum(sm) * (1.f / (float)C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const float a = xr[i][j][0] - mean, b = xr[i][j][1] - mean, c = xr[i][j][2] - mean, d = xr[i][j][3] - mean; q += a * a + b * b + c * c + d * d; }
        const float rstd = rsqrtf(group16_sum(q) * (1.f / (float)C) + p.eps);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = 4 * sub + 64 * j;
            const float4 g = *reinterpret_cast<const float4*>(p.gamma + k), b = *reinterpret_cast<const float4*>(p.beta + k);
            const float4 y = make_float4((xr[i][j][0] - mean) * rstd * g.x + b.x, (xr[i][j][1] - mean) * rstd * g.y + b.y, (xr[i][j][2] - mean) * rstd * g.z + b.z, (xr[i][j][3] - mean) * rstd * g.w + b.w);
            *reinterpret_cast<uint2*>(&A1[row * PA1 + k / 2]) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
        }
    }
    __syncthreads();
    // ---- Q | K | V: NQ passes of C output columns over K = C (phase F of flow_band_kernel with the stream starting at the QKV fragments: pass c sits in wb[c % 2])
    band_static_for<0, S::NQ>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        constexpr bool IS_V = c * C >= 2 * INNER;
        constexpr bool IN_WB1 = c % 2 == 1;
        v4f acc[RT][TA];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < TA; ++t) acc[rt][t] = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (IN_WB1) band_mma<TA, S::KC, 0, RT, IS_V>(wb1, A1, PA1, 0, lq, lg, acc); else band_mma<TA, S::KC, 0, RT, IS_V>(wb0, A1, PA1, 0, lq, lg, acc);
        if constexpr (c + 2 < S::NQ) { if constexpr (IN_WB1) band_wload<FCD>(wb1, ws, (c + 2) * FCD); else band_wload<FCD>(wb0, ws, (c + 2) * FCD); }
        if constexpr (!IS_V && TA == 2) {
            const int n = c * C + 32 * wave + 8 * lg;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int m = m0 + 16 * rt + lq;
                if (m < p.M) *reinterpret_cast<uint4*>(p.qk + (long long)m * p.ld_qk + n) =
                    make_uint4(pack_bf16x2(acc[rt][0][0], acc[rt][0][1]), pack_bf16x2(acc[rt][0][2], acc[rt][0][3]), pack_bf16x2(acc[rt][1][0], acc[rt][1][1]), pack_bf16x2(acc[rt][1][2], acc[rt][1][3]));
            }
        } else if constexpr (!IS_V) {
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const int n = c * C + 16 * (wave + NW * t) + 4 * lg;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int m = m0 + 16 * rt + lq;
                    if (m < p.M) *reinterpret_cast<uint2*>(p.qk + (long long)m * p.ld_qk + n) = make_uint2(pack_bf16x2(acc[rt][t][0], acc[rt][t][1]), pack_bf16x2(acc[rt][t][2], acc[rt][t][3]));
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const int n = c * C - 2 * INNER + 16 * (wave + NW * t) + lq;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int m = m0 + 16 * rt + 4 * lg;
                    const unsigned lo = pack_bf16x2(acc[rt][t][0] + 0.f, acc[rt][t][1] + 0.f), hi = pack_bf16x2(acc[rt][t][2] + 0.f, acc[rt][t][3] + 0.f);      // "+ 0": flow_gemm_big_kernel adds its (absent) bias here
                    if (m >= p.M) continue;
                    const int b = m / p.rows_per_batch, tt = m - b * p.rows_per_batch;
                    bf16_t* row = p.vt + (long long)b * p.vt_batch + (long long)n * p.ldt;
                    if (m + 3 < p.M && tt + 3 < p.rows_per_batch && (tt & 3) == 0) { *reinterpret_cast<uint2*>(row + vt_col(tt)) = make_uint2(lo, hi); continue; }
                    if (m + 3 < p.M && tt + 3 < p.rows_per_batch && (tt & 1) == 0) {
                        *reinterpret_cast<unsigned*>(row + vt_col(tt)) = lo; *reinterpret_cast<unsigned*>(row + vt_col(tt + 2)) = hi; continue;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mr = m + r;
                        if (mr >= p.M) continue;
                        const int br = mr / p.rows_per_batch, tr_ = mr - br * p.rows_per_batch;
                        const unsigned u = r < 2 ? lo : hi;
                        p.vt[(long long)br * p.vt_batch + (long long)n * p.ldt + vt_col(tr_)] = (bf16_t)((r & 1) ? (u >> 16) : (u & 0xffffu));
                    }
                }
            }
        }
    });
}

}  // namespace cv
