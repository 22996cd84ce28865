// The row-local TAIL of a flow-estimator transformer block as ONE launch (round 3; matcha BasicTransformerBlock inside
// CausalConditionalDecoder, cosyvoice/flow/decoder.py:405-494), bf16 mode:
//
//     attention output (bf16) -> out-projection + bias + residual -> LayerNorm(norm3) -> FF1 + bias + GELU(erf) -> FF2 + bias + residual
//       [-> LayerNorm(norm1 of the NEXT block) -> Q | K (bf16, row-major) and V^T (bf16, transposed + key-permuted for the P.V MFMA)]
//
// Everything after the attention of a block is row-local, yet round 2 ran it as four launches (out-projection, LN + FF1, FF2, and the next block's
// LN + QKV) of ~1000 single-tile workgroups each: 37 us of a 50 us block, 3.5 % MFMA-busy, the fp32 residual stream re-read once per N-tile
// through eight non-coherent L2s (3.3x over-fetch, profiles/r2_pmc_flow.json).  Here a workgroup owns a BAND of 16 rows and walks all four
// GEMMs with the activations parked in LDS (74 KB): the residual stream is read once and written once, nothing else of the band touches memory.
//
// What bounds the launch is then the WEIGHT stream: every band needs all 2 MB of the block's (and the next block's QKV) weights, L2-resident but
// pulled through a CU's ~64 B/clk L1 fill path (~14 us; the band count - 85 at T = 674 - only decides how many CUs do it at once).  So the
// weights never touch LDS: they are re-packed at load time into MFMA fragments in the exact order a wave consumes them
// (cosyvoice_amd/weights.py::pack_flow_tail: [wave][fragment][lane] x 16 bytes, one fully coalesced 1 KB read per fragment), each 16-byte load IS
// the matrix-pipe operand, and a wave keeps a ring of D fragments in flight ACROSS the phases - the LayerNorms and barriers run under the next
// phase's weights.  Two rules keep that ring alive (both learnt from the first version, profiles/r3_flow_tail_ab.txt: 26 full drains per launch, 52 us):
// vmcnt retires in issue order, so a load requested BEHIND ring loads cannot be waited for without draining the ring - every small operand
// (biases, LayerNorm affine, residual rows, the attention tile) is therefore requested FIRST and parked in LDS / registers; and a pending global
// STORE makes the compiler's counter model fall back to vmcnt(0) for the next load wait, so nothing is stored until the last fragment has been
// consumed - x, Q and K wait in LDS for one coalesced write-out at the end.  Products: v_mfma_f32_16x16x32_bf16, fp32 accumulate; statistics, residuals and the stream itself fp32: the rounding points are
// exactly those of the unfused bf16 path (operands rounded to bf16 where they are staged), so the two agree to summation order.
#pragma once
#include "flow_fused.h"

namespace cv {

struct FlowTailArgs {
    const bf16_t* att; int ld_att;            // attention output [M][INNER] bf16
    float* x; int ldx;                        // residual stream [M][C] fp32, read once, written once
    const u32x4_t* wstream;                   // packed fragments [4 waves][fragments per wave][64 lanes] (pack_flow_tail)
    const float* prm;                         // small operands of the block, packed (pack_flow_tail_params): [b_out C | norm3.g C | norm3.b C | b_ff1 FF | b_ff2 C |
                                              //  norm1.g of the NEXT block C | its norm1.b C] fp32
    float eps; int M;
    bf16_t* qk; int ld_qk;                    // HAS_QKV: Q | K [M][2 INNER] bf16
    bf16_t* vt; long long vt_batch; int ldt; int rows_per_batch;      // HAS_QKV: V^T [B][INNER][ldt] bf16, key-permuted (vt_col)
    long long* dbg;                           // dev tool (tools/ubench/tail_probe.hip): clock64() of thread 0 at the phase boundaries, 16 slots per workgroup; null in production
};

// One pass of PT 16-column tiles over KS k-steps of 32: operands A[16 rows][32 KS] in LDS (bf16 pairs, row pitch `pitch` dwords) against the next
// PT * KS fragments of the wave's stream (k-step major, tile minor).  `ring` slot i % D holds stream fragment i; a slot is re-requested the moment
// it has been multiplied.  SWAP: activations as the MFMA "A" (a lane ends with 4 consecutive ROWS of one column: the V^T epilogue) instead of the
// weights (4 consecutive COLUMNS of one row).
template <int PT, int KS, int BASE, int TOTAL, int D, bool SWAP>
__device__ __forceinline__ void tail_mma(u32x4_t (&ring)[D], const u32x4_t* ws, const unsigned* A, int pitch, int lq, int lg, v4f (&acc)[PT]) {
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const uint4 af = *reinterpret_cast<const uint4*>(&A[lq * pitch + ks * 16 + lg * 4]);
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int i = BASE + ks * PT + t;                                 // compile-time after unrolling
            const u32x4_t wf = ring[i % D];
            if (i + D < TOTAL) ring[i % D] = ws[(long long)(i + D) * 64];
            if (SWAP) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af), __builtin_bit_cast(v8bf, wf), acc[t], 0, 0, 0);
            else      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf), __builtin_bit_cast(v8bf, af), acc[t], 0, 0, 0);
        }
    }
}

// LayerNorm of the 16 rows parked in X (fp32, pitch PX floats) -> bf16 operand tile (pitch `pa` dwords).  16 threads per row, thread `sub` holds
// channels 4 sub + 64 j; two-pass statistics (mean, then centred variance) like torch.nn.LayerNorm and norm_rows_kernel, DPP row reductions.
template <int C, int PX>
__device__ __forceinline__ void tail_layernorm(const float* X, unsigned* A, int pa, const float* gamma, const float* beta, float eps, int tid) {
    constexpr int NJ = C / 64;
    const int row = tid >> 4, sub = tid & 15;
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

// fragments per wave of one block's stream (weights.py::pack_flow_tail must agree)
template <int C, int INNER, int FF, bool HAS_QKV>
struct FlowTailShape {
    static constexpr int NW = 4;
    static constexpr int TA = C / 16 / NW, KA = INNER / 32;        // out-projection: tiles per wave, k-steps
    static constexpr int TC = FF / 16 / NW, KC = C / 32;           // FF1
    static constexpr int TD = C / 16 / NW, KD = FF / 32;           // FF2
    static constexpr int TQ = INNER / 16 / NW, KQ = C / 32;        // each of Q, K, V
    static constexpr int FA = TA * KA, FC = TC * KC, FD = TD * KD, FQ = HAS_QKV ? 3 * TQ * KQ : 0;
    static constexpr int TOTAL = FA + FC + FD + FQ;
    static constexpr int PC = TC > 8 ? 8 : TC;                     // FF1 tiles per pass (accumulator budget), TC % PC == 0
    static_assert(C % 64 == 0 && INNER % 64 == 0 && FF % 64 == 0 && TC % PC == 0 && TA <= 8 && TQ <= 8, "flow_tail: unsupported dimensions");
};

template <int C, int INNER, int FF, bool HAS_QKV, int D>
__global__ __launch_bounds__(256) void flow_tail_kernel(FlowTailArgs p) {
    using S = FlowTailShape<C, INNER, FF, HAS_QKV>;
    constexpr int BM = 16;
    constexpr int PA0 = INNER / 2 + 4, PX = C + 4, PA1 = C / 2 + 4, PA2 = FF / 2 + 4;           // LDS row pitches (dwords / floats), +16 bytes against bank conflicts
    __shared__ __attribute__((aligned(16))) unsigned A0[BM * PA0];
    __shared__ __attribute__((aligned(16))) float X1[BM * PX];
    __shared__ __attribute__((aligned(16))) unsigned A1[BM * PA1];
    __shared__ __attribute__((aligned(16))) unsigned A2[BM * PA2];
    // small operands, parked once: [b_out C | g3 C | be3 C | b_ff1 FF | b_ff2 C | g1n C | be1n C]
    constexpr int O_BOUT = 0, O_G3 = C, O_BE3 = 2 * C, O_BFF1 = 3 * C, O_BFF2 = 3 * C + FF, O_G1N = 4 * C + FF, O_BE1N = 5 * C + FF, NPRM = 6 * C + FF;
    __shared__ __attribute__((aligned(16))) float prm[NPRM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int mrow = min(m0 + lq, p.M - 1);                                 // the band row this lane's accumulators belong to (clamped; stores are masked)
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 16 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();

    // ---- every small operand FIRST (vmcnt retires in order: what is requested behind the ring can only be had by draining it)
    constexpr int NPV = NPRM / 4, PPT = (NPV + 255) / 256;                  // float4 pieces of the parameter block per thread
    v4f pv[PPT];                                                            // (a native vector: as a float4 struct array this stayed an alloca parked in LDS)
#pragma unroll
    for (int i = 0; i < PPT; ++i) pv[i] = *reinterpret_cast<const v4f*>(p.prm + 4 * min(tid + 256 * i, NPV - 1));
    constexpr int PIECES = BM * INNER / 8, APT = (PIECES + 255) / 256;      // 16-byte pieces of the attention tile per thread
    u32x4_t av[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int v = min(tid + 256 * i, PIECES - 1), r = v / (INNER / 8), c = v % (INNER / 8);
        av[i] = *reinterpret_cast<const u32x4_t*>(p.att + (long long)min(m0 + r, p.M - 1) * p.ld_att + c * 8);
    }
    float4 res[S::TA];
#pragma unroll
    for (int t = 0; t < S::TA; ++t) res[t] = *reinterpret_cast<const float4*>(p.x + (long long)mrow * p.ldx + 16 * (wave + 4 * t) + 4 * lg);
    __builtin_amdgcn_sched_barrier(0);                                      // the scheduler may not sink any of the above below the ring loads
    // ---- then the weight stream of this wave: D fragments in flight from here to the last MFMA
    const u32x4_t* ws = p.wstream + (long long)wave * S::TOTAL * 64 + lane;
    u32x4_t ring[D];
#pragma unroll
    for (int i = 0; i < D; ++i) ring[i] = ws[(long long)(i < S::TOTAL ? i : 0) * 64];
#pragma unroll
    for (int i = 0; i < PPT; ++i) { const int v = tid + 256 * i; if (v < NPV) *reinterpret_cast<v4f*>(&prm[4 * v]) = pv[i]; }
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int v = tid + 256 * i;
        if (v < PIECES) *reinterpret_cast<u32x4_t*>(&A0[(v / (INNER / 8)) * PA0 + (v % (INNER / 8)) * 4]) = av[i];
    }
    __syncthreads();
    stamp();

    // ---- A: out-projection + bias + residual -> X1 (fp32)
    {
        v4f acc[S::TA];
        tail_mma<S::TA, S::KA, 0, S::TOTAL, D, false>(ring, ws, A0, PA0, lq, lg, acc);
#pragma unroll
        for (int t = 0; t < S::TA; ++t) {
            const int n = 16 * (wave + 4 * t) + 4 * lg;
            const float4 b = *reinterpret_cast<const float4*>(&prm[O_BOUT + n]);
            *reinterpret_cast<float4*>(&X1[lq * PX + n]) = make_float4(acc[t][0] + b.x + res[t].x, acc[t][1] + b.y + res[t].y, acc[t][2] + b.z + res[t].z, acc[t][3] + b.w + res[t].w);
        }
    }
    __syncthreads();
    stamp();
    // ---- B: LayerNorm(norm3) -> A1 (bf16)
    tail_layernorm<C, PX>(X1, A1, PA1, &prm[O_G3], &prm[O_BE3], p.eps, tid);
    __syncthreads();
    stamp();
    // ---- C: FF1 + bias + GELU -> A2 (bf16)
#pragma unroll
    for (int ps = 0; ps < S::TC / S::PC; ++ps) {
        v4f acc[S::PC];
        if (ps == 0) tail_mma<S::PC, S::KC, S::FA, S::TOTAL, D, false>(ring, ws, A1, PA1, lq, lg, acc);
        else         tail_mma<S::PC, S::KC, S::FA + S::PC * S::KC, S::TOTAL, D, false>(ring, ws, A1, PA1, lq, lg, acc);
#pragma unroll
        for (int t = 0; t < S::PC; ++t) {
            const int n = 16 * (wave + 4 * (ps * S::PC + t)) + 4 * lg;
            const float4 b = *reinterpret_cast<const float4*>(&prm[O_BFF1 + n]);
            const float4 y = apply_act4(ACT_GELU_ERF, make_float4(acc[t][0] + b.x, acc[t][1] + b.y, acc[t][2] + b.z, acc[t][3] + b.w), 0.f);
            *reinterpret_cast<uint2*>(&A2[lq * PA2 + n / 2]) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
        }
    }
    static_assert(S::TC / S::PC <= 2, "flow_tail: FF1 runs in at most two passes");
    __syncthreads();
    stamp();
    // ---- D: FF2 + bias + residual -> X1 (the new residual stream: written to memory at the END, and the next block's LayerNorm input)
    {
        v4f acc[S::TD];
        tail_mma<S::TD, S::KD, S::FA + S::FC, S::TOTAL, D, false>(ring, ws, A2, PA2, lq, lg, acc);
#pragma unroll
        for (int t = 0; t < S::TD; ++t) {
            const int n = 16 * (wave + 4 * t) + 4 * lg;
            const float4 b = *reinterpret_cast<const float4*>(&prm[O_BFF2 + n]);
            const float4 r = *reinterpret_cast<const float4*>(&X1[lq * PX + n]);
            *reinterpret_cast<float4*>(&X1[lq * PX + n]) = make_float4(acc[t][0] + b.x + r.x, acc[t][1] + b.y + r.y, acc[t][2] + b.z + r.z, acc[t][3] + b.w + r.w);   // the same lane read this element: no hazard inside the phase
        }
    }
    __syncthreads();
    stamp();
    if constexpr (HAS_QKV) {
        // ---- E: LayerNorm(norm1 of the next block) -> A1
        tail_layernorm<C, PX>(X1, A1, PA1, &prm[O_G1N], &prm[O_BE1N], p.eps, tid);
        __syncthreads();
        stamp();
        // ---- F: Q, K (row-major bf16, parked in A2 - free since FF2 - until the write-out) and V^T of the next block
        constexpr int BQ = S::FA + S::FC + S::FD;
        constexpr int PQK = 2 * INNER / 2 + 4;                            // row pitch (dwords) of the parked Q | K tile
        static_assert(BM * PQK <= BM * PA2, "flow_tail: the Q | K tile must fit the FF hidden tile it replaces (2 INNER <= FF)");
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            v4f acc[S::TQ];
            if (which == 0) tail_mma<S::TQ, S::KQ, BQ, S::TOTAL, D, false>(ring, ws, A1, PA1, lq, lg, acc);
            else            tail_mma<S::TQ, S::KQ, BQ + S::TQ * S::KQ, S::TOTAL, D, false>(ring, ws, A1, PA1, lq, lg, acc);
#pragma unroll
            for (int t = 0; t < S::TQ; ++t) {
                const int n = which * INNER + 16 * (wave + 4 * t) + 4 * lg;
                *reinterpret_cast<uint2*>(&A2[lq * PQK + n / 2]) = make_uint2(pack_bf16x2(acc[t][0], acc[t][1]), pack_bf16x2(acc[t][2], acc[t][3]));
            }
        }
        {
            v4f acc[S::TQ];
            tail_mma<S::TQ, S::KQ, BQ + 2 * S::TQ * S::KQ, S::TOTAL, D, true>(ring, ws, A1, PA1, lq, lg, acc);      // the LAST fragments of the stream: stores are free from here
            // swapped operands: lane holds rows m0 + 4 lg + r of column n.  Rows of one request are consecutive (m = b rows_per_batch + t) but a group of four
            // may straddle a 4-aligned key group or two requests when rows_per_batch % 4 != 0, so every element finds its own slot (as flow_gemm_kernel does).
#pragma unroll
            for (int t = 0; t < S::TQ; ++t) {
                const int n = 16 * (wave + 4 * t) + lq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 4 * lg + r;
                    if (m >= p.M) continue;
                    const int b = m / p.rows_per_batch, tt = m - b * p.rows_per_batch;
                    const unsigned u = pack_bf16x2(acc[t][r], 0.f);
                    p.vt[(long long)b * p.vt_batch + (long long)n * p.ldt + vt_col(tt)] = (bf16_t)(u & 0xffffu);
                }
            }
        }
        __syncthreads();
        stamp();
        // write-out of the parked Q | K tile: 16-byte pieces, whole rows (4 KB at INNER = 512) per 256 consecutive lanes
        constexpr int QP = BM * 2 * INNER / 8;
#pragma unroll
        for (int i = 0; i < (QP + 255) / 256; ++i) {
            const int v = tid + 256 * i, r = v / (2 * INNER / 8), c = v % (2 * INNER / 8);
            if (v < QP && m0 + r < p.M) *reinterpret_cast<u32x4_t*>(p.qk + (long long)(m0 + r) * p.ld_qk + c * 8) = *reinterpret_cast<const u32x4_t*>(&A2[r * PQK + c * 4]);
        }
    }
    // write-out of the residual stream (after the last weight fragment has been consumed)
    constexpr int XP = BM * C / 4;
#pragma unroll
    for (int i = 0; i < (XP + 255) / 256; ++i) {
        const int v = tid + 256 * i, r = v / (C / 4), c = v % (C / 4);
        if (v < XP && m0 + r < p.M) *reinterpret_cast<float4*>(p.x + (long long)(m0 + r) * p.ldx + c * 4) = *reinterpret_cast<const float4*>(&X1[r * PX + c * 4]);
    }
    stamp();
}

}  // namespace cv
