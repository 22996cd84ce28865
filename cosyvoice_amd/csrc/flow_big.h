// Large-M forms of the fused bf16 transformer-block pipeline of the flow estimator (flow_fused.h; matcha BasicTransformerBlock inside
// CausalConditionalDecoder, cosyvoice/flow/decoder.py:405-494) for gfx950 - what a flow pass shared by several utterances runs on.
//
// Round-4 profile of the shared pass at 8 utterances (M = 10 784 rows; profiles/r4_rocprof_flow_batch8_kernel_stats.csv): the LayerNorm-prologue
// GEMMs take 54 us per launch (157 TFLOP/s), the residual GEMMs 18.6 us, attention 49 us - 195 of the ~250 us per block evaluation.  The small tiles of
// flow_fused.h are cut for M = 1348 (one round of short dependent chains); at M >= 4096 they re-read and re-normalise the fp32 rows once per 64
// output columns (QKV: 24 x) and put every operand byte through a CU's load path for 1 MFLOP per 64 KB.  Here:
//   ln_bf16_kernel            LayerNorm once per row, bf16 out (the arithmetic of flow_gemm_kernel's prologue, instruction for instruction)
//   flow_gemm_big_kernel      bf16 A, K streamed in 64-wide stages through a double-buffered, XOR-swizzled LDS ring with a register prefetch stage,
//                             128 x 128 / 128 x 64 / 64 x 64 output tiles of 16 x 16 x 32 MFMAs; the epilogues of flow_gemm_kernel
//   attn_flow_kernel<4,2,2,2> (flow_fused.h, QG = 2) 128 queries per workgroup, 32 per wave: every K / V^T fragment read from LDS feeds two MFMAs
// BIT-IDENTICAL to the small-tile path by construction - the same MFMA shape, the same k order into one accumulator chain per output element, the same
// LayerNorm / softmax / merge expressions in the same order - so the kernel choice may follow the row count of a pass without an utterance's
// mel depending on what it shared the pass with (tests/test_flow.py::test_big_m_kernels_are_bit_identical).
#pragma once
#include "flow_fused.h"

namespace cv {

// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm over K <= 256 channels of fp32 rows -> bf16 rows.  A 16-lane group owns a row, lane `sub` holds channels 4 sub + 64 j - the register
// layout, reduction order and expressions of flow_gemm_kernel<.., AMODE 1>'s prologue (keep the two in step: the bf16 values must be the same bits).
// ---------------------------------------------------------------------------------------------------------------------------------
struct LnBf16Args { const float* x; int ldx; const float* gamma; const float* beta; float eps; bf16_t* y; int ldy; int M, K; };

__global__ __launch_bounds__(256) void ln_bf16_kernel(LnBf16Args p) {
    const int tid = threadIdx.x, grp = tid >> 4, sub = tid & 15;
    const int m = blockIdx.x * 16 + grp;
    const float* xr = p.x + (long long)min(m, p.M - 1) * p.ldx;
    float4 x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 4 * sub + 64 * j;
        float4 t = *reinterpret_cast<const float4*>(xr + min(k, p.K - 4));
        if (k >= p.K) t = make_float4(0.f, 0.f, 0.f, 0.f);
        x[j] = t;
    }
    float4 ga[4], be[4];
    if (p.gamma) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = min(4 * sub + 64 * j, p.K - 4);
            ga[j] = *reinterpret_cast<const float4*>(p.gamma + k); be[j] = *reinterpret_cast<const float4*>(p.beta + k);
        }
    }
    const float invK = 1.f / (float)p.K;
    if (p.gamma) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += x[j].x + x[j].y + x[j].z + x[j].w;
        const float mean = group16_sum(s) * invK;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * sub + 64 * j < p.K) { const float a = x[j].x - mean, b = x[j].y - mean, c = x[j].z - mean, d = x[j].w - mean; q += a * a + b * b + c * c + d * d; }
        const float rstd = rsqrtf(group16_sum(q) * invK + p.eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j].x = (x[j].x - mean) * rstd * ga[j].x + be[j].x; x[j].y = (x[j].y - mean) * rstd * ga[j].y + be[j].y;
            x[j].z = (x[j].z - mean) * rstd * ga[j].z + be[j].z; x[j].w = (x[j].w - mean) * rstd * ga[j].w + be[j].w;
        }
    }
    if (m >= p.M) return;
    bf16_t* yr = p.y + (long long)m * p.ldy;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 4 * sub + 64 * j;
        if (k < p.K) *reinterpret_cast<uint2*>(yr + k) = make_uint2(pack_bf16x2(x[j].x, x[j].y), pack_bf16x2(x[j].z, x[j].w));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// out = epi(A W^T): A bf16 [M][lda], W bf16 [N][Kp] (FlowGemmArgs, AMODE 0 fields; OMODE as in flow_gemm_kernel).  CONV: a causal Conv1d over the rows of every
// request (taps x K contraction, tap-major like gemm_conv_kernel's bf16 tiles, which this form replaces for the ResNet convolutions of a large pass).  BM x BN tile per workgroup of
// 2 x 2 waves; K in stages of 64: stage c is multiplied out of LDS buffer c & 1 while stage c + 1 is written into the other buffer and stage c + 2 is
// in flight in registers - one barrier per stage.  LDS rows are 64 bf16 = eight 16-byte slots without padding, slot s of row r stored at s ^ ((r >> 1) & 7):
// the fragment reads (row r, slot 4 kg + g) of a 16-lane LDS group then cover all 64 banks once (common.h, LDS_PAD note), and so does a row's store.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int OMODE, bool CONV = false>
__global__ __launch_bounds__(256) void flow_gemm_big_kernel(FlowGemmArgs p) {
    constexpr int BK = 64, RP = BK / 2;                       // row pitch in dwords
    constexpr int TM = BM / 32, TN = BN / 32;                 // 16 x 16 MFMA tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int AV = BM * 8 / 256, WV = BN * 8 / 256;       // 16-byte pieces per thread and stage
    static_assert(BM % 32 == 0 && BN % 32 == 0 && AV >= 1 && WV >= 1, "tile must split over 2 x 2 waves of 16 x 16 MFMA tiles");
    __shared__ __attribute__((aligned(16))) unsigned Ls[2 * (BM + BN) * RP];
    unsigned* const As0 = Ls; unsigned* const Ws0 = Ls + 2 * BM * RP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int ntn = (p.N + BN - 1) / BN;
    const int bl = xcd_remap((int)blockIdx.x, (int)gridDim.x);        // an XCD walks the N tiles of a band of rows: A crosses the fabric once
    const int m0 = (bl / ntn) * BM, n0 = (bl % ntn) * BN;
    const int spt = (p.K + BK - 1) / BK;                       // stages per tap
    const int nst = CONV ? p.taps * spt : spt;

    bool tr[TN];
    v4f acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) tr[j] = OMODE == 0 && n0 + wn * (BN / 2) + j * 16 >= p.n_row;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    // piece v = tid + 256 i of a stage: row v / 8, slot v % 8 (k = 8 (v % 8)); row bases once, a stage adds its uniform k offset
    const bf16_t* a_ptr[AV]; const bf16_t* w_ptr[WV]; int a_lds[AV], w_lds[WV], a_t[CONV ? AV : 1];
    const long long wpitch = CONV ? (long long)p.taps * p.Kp : p.Kp;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int v = tid + 256 * i, r = v >> 3, s = v & 7, m = min(m0 + r, p.M - 1);
        a_ptr[i] = reinterpret_cast<const bf16_t*>(p.A) + (long long)m * p.lda + 8 * s;
        a_lds[i] = r * RP + ((s ^ ((r >> 1) & 7)) << 2);
        if constexpr (CONV) a_t[i] = m % p.rows_per_batch;   // row of its request: taps never reach into the previous request
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int v = tid + 256 * i, r = v >> 3, s = v & 7;
        w_ptr[i] = p.W + (long long)min(n0 + r, p.N - 1) * wpitch + 8 * s;
        w_lds[i] = r * RP + ((s ^ ((r >> 1) & 7)) << 2);
    }
    u32x4_t ra[AV], rw[WV];
    auto load = [&](int c) {
        int tap = 0, kc = c;
        if constexpr (CONV) { tap = c / spt; kc = c - tap * spt; }
        const int k0 = kc * BK, dr = tap - p.pad_left;        // uniform
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int s8 = 8 * ((tid + 256 * i) & 7);
            if constexpr (CONV) {                             // unconditional load (clamped row) + select: the zero padding before a request's first row
                const int tr_ = a_t[i] + dr;
                u32x4_t v = *reinterpret_cast<const u32x4_t*>(a_ptr[i] + (long long)(max(tr_, 0) - a_t[i]) * p.lda + min(k0, p.K - 8 - s8));
                if (tr_ < 0) v = (u32x4_t){0u, 0u, 0u, 0u};
                ra[i] = v;
            } else ra[i] = *reinterpret_cast<const u32x4_t*>(a_ptr[i] + min(k0, p.K - 8 - s8));      // clamped: steps beyond K are never multiplied
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) { const int s8 = 8 * ((tid + 256 * i) & 7); rw[i] = *reinterpret_cast<const u32x4_t*>(w_ptr[i] + (CONV ? tap * p.Kp : 0) + min(k0, p.Kp - 8 - s8)); }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AV; ++i) *reinterpret_cast<u32x4_t*>(&As0[buf * BM * RP + a_lds[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < WV; ++i) *reinterpret_cast<u32x4_t*>(&Ws0[buf * BN * RP + w_lds[i]]) = rw[i];
    };
    // fragment address of this lane: row (lane & 15) of a 16-row tile, slot 4 kg + (lane >> 4), swizzled by the row (tile bases are multiples of 16)
    const int fr = lane & 15, fx = (fr >> 1) & 7, fg = lane >> 4;
    auto compute = [&](int buf, int ksteps) {
        const unsigned* Ab = &As0[buf * BM * RP + (wm * (BM / 2) + fr) * RP];
        const unsigned* Wb = &Ws0[buf * BN * RP + (wn * (BN / 2) + fr) * RP];
#pragma unroll
        for (int kg = 0; kg < BK / 32; ++kg) {
            if (kg >= ksteps) break;
            const int so = (((kg << 2) + fg) ^ fx) << 2;
            uint4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(Ab + i * 16 * RP + so);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const uint4*>(Wb + j * 16 * RP + so);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (tr[j]) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af[i]), __builtin_bit_cast(v8bf, wf[j]), acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[j]), __builtin_bit_cast(v8bf, af[i]), acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    load(0);
    store(0);
    if (nst > 1) load(1);
    __syncthreads();
    for (int c = 0; c < nst; ++c) {
        if (c + 1 < nst) {
            store((c + 1) & 1);                               // buffer (c + 1) & 1 was last read by stage c - 1: every wave left it before the barrier that ended it
            if (c + 2 < nst) load(c + 2);
        }
        compute(c & 1, min(BK, p.K - (CONV ? c % spt : c) * BK) / 32);
        __syncthreads();
    }

    // ---- epilogues: the expressions of flow_gemm_kernel, element for element
    if constexpr (OMODE == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (BM / 2) + i * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
                if (n >= p.N) continue;
                float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                const long long idx = (long long)m * p.ldc + n;
                if (p.res) { const float4 r = *reinterpret_cast<const float4*>(p.res + idx); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                if constexpr (CONV) { v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f; }      // gemm_conv_kernel's epilogue ends in "+ 0" (its accumulate term): a -0 result becomes +0 there
                *reinterpret_cast<float4*>(p.C + idx) = v;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!tr[j]) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = m0 + wm * (BM / 2) + i * 16 + (lane & 15);
                    const int n = n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
                    if (m >= p.M || n >= p.N) continue;
                    float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                    v = apply_act4(p.act, v, 0.f);
                    *reinterpret_cast<uint2*>(p.out + (long long)m * p.ldo + n) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                }
            } else {
                // V^T section: the lane holds rows m .. m + 3 (m % 4 == 0) of column n.  Same values as flow_gemm_kernel, wider stores: a 4-aligned key
                // group of one request is 4 consecutive V^T columns (vt_col) -> one 8-byte store; an even-aligned pair 2 columns -> one 4-byte store.
                const int n = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
                if (n >= p.N) continue;
                const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4;
                    if (m >= p.M) continue;
                    const int b = m / p.rows_per_batch, t = m - b * p.rows_per_batch;
                    const unsigned lo = pack_bf16x2(acc[i][j][0] + bn, acc[i][j][1] + bn), hi = pack_bf16x2(acc[i][j][2] + bn, acc[i][j][3] + bn);
                    bf16_t* row = p.outT + (long long)b * p.t_batch + (long long)(n - p.n_row) * p.ldt;
                    if (m + 3 < p.M && t + 3 < p.rows_per_batch && (t & 3) == 0) { *reinterpret_cast<uint2*>(row + vt_col(t)) = make_uint2(lo, hi); continue; }
                    if (m + 3 < p.M && t + 3 < p.rows_per_batch && (t & 1) == 0) {
                        *reinterpret_cast<unsigned*>(row + vt_col(t)) = lo; *reinterpret_cast<unsigned*>(row + vt_col(t + 2)) = hi; continue;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mr = m + r;
                        if (mr >= p.M) continue;
                        const int br = mr / p.rows_per_batch, tr_ = mr - br * p.rows_per_batch;
                        const unsigned u = r < 2 ? lo : hi;
                        p.outT[(long long)br * p.t_batch + (long long)(n - p.n_row) * p.ldt + vt_col(tr_)] = (bf16_t)((r & 1) ? (u >> 16) : (u & 0xffffu));
                    }
                }
            }
        }
    }
}

}  // namespace cv
