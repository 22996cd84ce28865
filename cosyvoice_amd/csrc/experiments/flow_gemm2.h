// Round-3 forms of the two GEMMs of the flow estimator's transformer blocks (flow_fused.h has the round-2 forms and the argument struct).
//
// What round 2 left: at M = 1348 rows a launch is ~1000 workgroups that are each ONE dependent chain (load burst -> LayerNorm -> LDS -> barrier ->
// 16 MFMAs -> epilogue), both operands staged through LDS (50 KB per workgroup: three workgroups per CU, the launch runs in two rounds), and a CU takes
// in a few tens of bytes per cycle.  The forms below change what a workgroup asks of its CU, not the arithmetic:
//   flow_ln_gemm_kernel<BM, NW>      LayerNorm-prologue GEMM (K <= 256).  The NW waves sit side by side along N, 16 columns each, and load their
//                                    weight rows STRAIGHT into MFMA fragments (a lane's 8 bf16 of a k-step are 16 contiguous bytes of its row): the
//                                    weights never pass through LDS, the workgroup's LDS is the BM x K activation tile alone (17 KB at BM = 32) and
//                                    every workgroup of the launch is resident at once.  Same products in the same order as flow_gemm_kernel<.,.,1,0>:
//                                    bit-identical output.
//   flow_res_gemm_kernel<BM, NW, KS> bf16 A (K = 256 KS) -> fp32 out + bias + residual.  Single shot: KS wave sets take one 256-wide k range each, all
//                                    of the workgroup's operand bytes are requested up front (A through LDS once, W straight into fragments), one
//                                    barrier, 8 k-steps per wave, the sets are summed through LDS in a fixed order.  KS = 1 reproduces the round-2
//                                    summation order; KS > 1 adds the k ranges pairwise ((k0 + k1) + (k2 + k3) at KS = 4): fp32-rounding distance.
#pragma once
#include "flow_fused.h"

namespace cv {

template <int BM, int NW>
__global__ __launch_bounds__(NW * 64) void flow_ln_gemm_kernel(FlowGemmArgs p) {
    constexpr int KC = 256, LDK = KC / 2 + 4, BN = NW * 16, NT = NW * 64, TM = BM / 16;
    constexpr int G = NT / 16;                                // 16-lane groups of the workgroup: group g normalises rows g, g + G, ...
    constexpr int AR = (BM + G - 1) / G;
    static_assert(BM % 16 == 0, "flow_ln_gemm: BM is a multiple of the 16-row MFMA tile");
    __shared__ __attribute__((aligned(16))) unsigned As[BM * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 15, lg = lane >> 4;
    const int ntn = (p.N + BN - 1) / BN;
    const int bl = xcd_remap((int)blockIdx.x, (int)gridDim.x);        // an XCD walks the N tiles of a band of rows: A crosses the fabric once
    const int m0 = (bl / ntn) * BM, n0 = (bl % ntn) * BN;
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();
    const int nw0 = n0 + wave * 16;                           // this wave's 16 columns
    const bool tr = nw0 >= p.n_row;                           // V^T section: wave-uniform
    const int ksteps = p.K / 32;

    // ---- the wave's weight rows, straight into fragments: lane (lq, lg) holds k = 32 kg + 8 lg .. + 7 of row nw0 + lq
    uint4 wf[8];
    {
        const bf16_t* wr = p.W + (long long)min(nw0 + lq, p.N - 1) * p.Kp + lg * 8;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) wf[kg] = *reinterpret_cast<const uint4*>(wr + min(kg * 32, p.Kp - 32));      // steps beyond K are never multiplied
    }
    // ---- activation rows: a 16-lane group per row, lane `sub` holds channels 4 sub + 64 j (as flow_gemm_kernel: same statistics, same rounding)
    const int grp = tid >> 4, sub = tid & 15;
    float4 x[AR][4];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int m = min(m0 + min(grp + G * r, BM - 1), p.M - 1);
        const float* xr = reinterpret_cast<const float*>(p.A) + (long long)m * p.lda;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * sub + 64 * j;
            float4 t = *reinterpret_cast<const float4*>(xr + min(k, p.K - 4));
            if (k >= p.K) t = make_float4(0.f, 0.f, 0.f, 0.f);
            x[r][j] = t;
        }
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
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        if (p.gamma) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += x[r][j].x + x[r][j].y + x[r][j].z + x[r][j].w;          // channels >= K hold zeros
            const float mean = group16_sum(s) * invK;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * sub + 64 * j < p.K) { const float a = x[r][j].x - mean, b = x[r][j].y - mean, c = x[r][j].z - mean, d = x[r][j].w - mean; q += a * a + b * b + c * c + d * d; }
            const float rstd = rsqrtf(group16_sum(q) * invK + p.eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[r][j].x = (x[r][j].x - mean) * rstd * ga[j].x + be[j].x; x[r][j].y = (x[r][j].y - mean) * rstd * ga[j].y + be[j].y;
                x[r][j].z = (x[r][j].z - mean) * rstd * ga[j].z + be[j].z; x[r][j].w = (x[r][j].w - mean) * rstd * ga[j].w + be[j].w;
            }
        }
        if (grp + G * r < BM) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<uint2*>(&As[(grp + G * r) * LDK + 2 * sub + 32 * j]) = make_uint2(pack_bf16x2(x[r][j].x, x[r][j].y), pack_bf16x2(x[r][j].z, x[r][j].w));
        }
    }
    stamp();
    __syncthreads();
    stamp();

    v4f acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        if (kg < ksteps) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const uint4 af = *reinterpret_cast<const uint4*>(&As[(i * 16 + lq) * LDK + kg * 16 + lg * 4]);
                if (tr) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af), __builtin_bit_cast(v8bf, wf[kg]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[kg]), __builtin_bit_cast(v8bf, af), acc[i], 0, 0, 0);
            }
        }
    }
    stamp();

    // ---- epilogue (the one of flow_gemm_kernel<.,.,1,0>, one 16-column tile per wave)
    if (!tr) {
        const int n = nw0 + lg * 4;
        if (n < p.N) {
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + i * 16 + lq;
                if (m >= p.M) continue;
                float4 v = make_float4(acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w);
                v = apply_act4(p.act, v, 0.f);
                *reinterpret_cast<uint2*>(p.out + (long long)m * p.ldo + n) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            }
        }
    } else {
        const int n = nw0 + lq;
        if (n < p.N) {
            const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + i * 16 + lg * 4 + r;
                    if (m >= p.M) continue;
                    const int b = m / p.rows_per_batch, t = m - b * p.rows_per_batch;
                    const unsigned u = pack_bf16x2(acc[i][r] + bn, 0.f);
                    p.outT[(long long)b * p.t_batch + (long long)(n - p.n_row) * p.ldt + vt_col(t)] = (bf16_t)(u & 0xffffu);
                }
        }
    }
    stamp();
}

// C[m][n] = sum_k A[m][k] W[n][k] + bias[n] (+ res[m][n]), A bf16, K == 256 * KS, fp32 out
template <int BM, int NW, int KS>
__global__ __launch_bounds__(NW * KS * 64) void flow_res_gemm_kernel(FlowGemmArgs p) {
    constexpr int KC = 256, LDK = KC / 2 + 4, BN = NW * 16, NT = NW * KS * 64, TM = BM / 16;
    constexpr int PIECES = KS * BM * (KC / 8);                // 16-byte pieces of the workgroup's A tile
    constexpr int AV = (PIECES + NT - 1) / NT;
    static_assert(BM % 16 == 0, "flow_res_gemm: BM is a multiple of the 16-row MFMA tile");
    __shared__ __attribute__((aligned(16))) unsigned As[KS * BM * LDK];          // [ks][row][k pairs]; re-used as the merge scratch
    static_assert(KS * BM * LDK >= (KS - 1) * NW * 64 * TM * 4, "flow_res_gemm: merge scratch does not fit the A tile");

    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, wave = wave_all % NW, ks = wave_all / NW, lq = lane & 15, lg = lane >> 4;
    const int ntn = (p.N + BN - 1) / BN;
    const int bl = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int m0 = (bl / ntn) * BM, n0 = (bl % ntn) * BN;
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();
    const int nw0 = n0 + wave * 16;

    // ---- A tile: piece v -> (k range v / (BM * 32), row (v / 32) % BM, k = 8 (v % 32)); requested first (it has the LDS round trip ahead of it)
    u32x4_t ra[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int v = min(tid + NT * i, PIECES - 1), kk = v / (BM * 32), row = (v / 32) % BM, kc = 8 * (v % 32);
        ra[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(p.A) + (long long)min(m0 + row, p.M - 1) * p.lda + kk * KC + kc);
    }
    // ---- the wave's weight rows of its k range, straight into fragments
    uint4 wf[8];
    {
        const bf16_t* wr = p.W + (long long)min(nw0 + lq, p.N - 1) * p.Kp + ks * KC + lg * 8;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) wf[kg] = *reinterpret_cast<const uint4*>(wr + kg * 32);
    }
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int v = tid + NT * i;
        if (v < PIECES) *reinterpret_cast<u32x4_t*>(&As[(v / 32) * LDK + 4 * (v % 32)]) = ra[i];          // (kk * BM + row) == v / 32
    }
    stamp();
    __syncthreads();
    stamp();

    v4f acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint4 af = *reinterpret_cast<const uint4*>(&As[(ks * BM + i * 16 + lq) * LDK + kg * 16 + lg * 4]);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[kg]), __builtin_bit_cast(v8bf, af), acc[i], 0, 0, 0);
        }
    }
    if constexpr (KS > 1) {
        // fixed-order merge of the k ranges: pairwise tree over the sets (ks ^ 1 first, then ks ^ 2): every set writes, the lower partner adds
        float* mg = reinterpret_cast<float*>(As);
        __syncthreads();                                      // every wave is done reading the A tile
#pragma unroll
        for (int step = 1; step < KS; step *= 2) {
            if ((ks & (2 * step - 1)) == step) {
                float* d = mg + (((ks - step) / (2 * step)) * NW * 64 + wave * 64 + lane) * TM * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) *reinterpret_cast<float4*>(d + 4 * i) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            }
            __syncthreads();
            if ((ks & (2 * step - 1)) == 0) {
                const float* d = mg + ((ks / (2 * step)) * NW * 64 + wave * 64 + lane) * TM * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) { const float4 t = *reinterpret_cast<const float4*>(d + 4 * i); acc[i][0] += t.x; acc[i][1] += t.y; acc[i][2] += t.z; acc[i][3] += t.w; }
            }
            if (2 * step < KS) __syncthreads();
        }
        if (ks > 0) { stamp(); stamp(); return; }
    }
    stamp();
    {
        const int n = nw0 + lg * 4;
        if (n < p.N) {
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + i * 16 + lq;
                if (m >= p.M) continue;
                float4 v = make_float4(acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w);
                const long long idx = (long long)m * p.ldc + n;
                if (p.res) { const float4 r = *reinterpret_cast<const float4*>(p.res + idx); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                *reinterpret_cast<float4*>(p.C + idx) = v;
            }
        }
    }
    stamp();
}

}  // namespace cv
