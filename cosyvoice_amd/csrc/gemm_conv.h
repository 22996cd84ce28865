// Implicit-GEMM linear / conv1d / transposed-conv kernel for gfx950 (fp32-accurate: fp32 MFMA chain, or - bf16 weights - the exact three-term split, AX3).
//
//   C[b][m][n] = epi( sum_{tap<taps} sum_{k<K}  pro(A_b[m*lda + a_off0 + tap*tap_step + k]) * W[n][tap*Kp + k] )
//
// * activations A are fp32, channel-last ([time][channel]); the flat-index form lets one kernel
//   serve Linear (taps=1), causal / "same" / dilated Conv1d (tap_step = dilation*C_in, a_off0 = -pad*C_in),
//   strided Conv1d (lda = stride*C_in, K = k*C_in: an im2col window is contiguous in channel-last),
//   and ConvTranspose1d in polyphase form (tap_step = -C_in, N = stride*C_out, c_off = -pad*C_out).
//   Any flat index outside [0, a_len) reads as zero (that IS the zero padding).
// * weights W are bf16 (LLM / flow: "W16A32") or fp32 (HiFT), row-major [N][taps*Kp], Kp = round_up(K,32)
//   zero padded by the host repacker (cosyvoice_amd/weights.py).
// * math: v_mfma_f32_16x16x4_f32 — bitwise an fp32 fma chain (cdna_hip_programming.md §3), so the
//   result differs from the CPU oracle only by summation order.
// * ABF16 variant ("W16A16 operands, fp32 accumulate", BASELINE.json configs[1] is a bf16 configuration): the prologue'd
//   activations are rounded to bf16 (round-to-nearest-even) when they are staged into LDS, weights stay bf16, the product
//   runs on v_mfma_f32_16x16x32_bf16 (8x the fp32 MFMA rate, half / quarter the LDS bytes).  Activations in HBM, bias,
//   epilogue and accumulation stay fp32.  The oracle mirrors the rounding (oracle/flow.py `bf16_act`).
#pragma once
#include "common.h"

namespace cv {

struct GemmConvArgs {
    // A operand
    const float* A; long long a_batch; long long a_len; int lda; int a_off0; int tap_step; int taps; int K;
    int a_vec;                      // 1: every float4 group is 16B aligned and fully in or fully out of range
    int pro; float pro_p; const float* pro_alpha;   // prologue activation on A (ACT_NONE / ACT_LEAKY / ACT_SNAKE)
    // W operand
    const void* W; int Kp;          // row stride = ldw (0 -> taps*Kp)
    long long ldw; long long w_batch;   // W may itself be an activation (rel-pos scores): explicit row pitch + per-batch offset
    const float* bias;              // [N] or null
    // output
    float* C; long long c_batch; long long c_len; int ldc; long long c_off; int c_vec;
    int M, N;
    int act; float act_p;           // epilogue activation
    const float* res; long long res_batch;          // residual, indexed like C (null = none)
    float out_scale;
    const float* row_scale; long long row_scale_batch;  // per-row multiplier (time mask), null = none
    int accumulate;                 // C += result
    int a_bf16;                     // 1: bf16 x bf16 MFMA (needs bf16 weights); 0: fp32-accurate (fp32 MFMA chain / three-term split)
    long long bias_batch;           // float offset of the bias per batch (grouped convolutions: one bias slice per group); 0 = shared
    const float* col_scale; int col_scale_rows; long long col_scale_stride;   // per-(row block, column) multiplier applied to act(acc + bias) BEFORE the residual:
                                    // the adaLN-zero gates of the DiT (x + gate[b] * f(x), flow/DiT/modules.py:523-528); row m uses block m / col_scale_rows
    const float* act_alpha;         // act == ACT_SNAKE in the EPILOGUE: Snake(acc + bias) with alpha[n] per output column (HiFT ResBlocks, round 3: the activation
                                    // is applied ONCE where a value is produced instead of in the prologue of every tap / N-tile that consumes it)
    float* C2; const float* c2_alpha;   // optional second output, indexed like C: C2 = Snake(final value, c2_alpha[n]) - the next convolution's operand
    const void* W3;                 // optional: the SAME fp32 weights pre-split into three bf16 planes, rows [3 N][taps * Kp] with row 3 n + p = plane p of row n
                                    // (w = w1 + w2 + w3 exactly, weights.py::split3_planes).  With fp32 weights and a_vec the products then run on the bf16 pipe (WX3 below).
    int w3_terms;                   // WX3 only: 0 / 6 = the six plane products of relative weight >= 2^-16 (fp32-exact class, the default); 3 = x1 w1 + x1 w2 + x2 w1 only
                                    // (relative weight >= 2^-8; the dropped terms are <= 2^-16 of a product: 16 mantissa bits per factor - HiFT option "terms", round 6)
    long long* dbg;                 // dev tool (tools/ubench/gemm_probe.hip): per-phase clock64() stamps of wave 0, 64 slots per workgroup; null in production
};

// Pipeline: most GEMMs on this path are small (M ~ 10^3, K = 256..1024) and run ~1 workgroup per CU, so nothing hides global
// latency but the kernel itself: a 2-deep REGISTER prefetch ring keeps the loads of k-tiles it+1 and it+2 in flight while tile
// it is multiplied out of LDS, and the small tiles use BK = 64 to halve the number of barrier-separated iterations.
// WM x WN = wave grid of a workgroup (64 * WM * WN threads); a wave owns a (BM / WM) x (BN / WN) sub-tile.  The default 2 x 2 serves
// every tile from 32 x 32 up; 1 x 2 makes the 16 x 32 tile (two-wave workgroups) that lets the N = 256 GEMMs of the flow run ~3
// workgroups per CU (680 instead of 344 launches' worth at M = 1348).
// AX3 ("exact fp32 on the bf16 pipe", bf16 weights only): an fp32 activation is the exact sum of three bf16 numbers (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = x - x1 - x2: 3 x 8 mantissa bits, every residual exact in fp32), so  x . w = x1 . w + x2 . w + x3 . w  with every product exact and fp32
// accumulation - the accuracy of the fp32 MFMA chain for 3 x 16 cycles per 32 k instead of 8 x 32 (the fp32 tiles are MFMA-bound: one wave per
// SIMD, ~1300 cycles of v_mfma_f32_16x16x4_f32 per 128-wide k-step).  The activation is split ONCE, when it is staged: three bf16 planes in LDS
// (6 bytes per element instead of 4), the weights staged as the raw bf16 they are.  This is what the LLM prefill and the fp32 mode of the flow run on.
// WX3 (round 3, fp32 WEIGHTS - HiFT): both operands split, x = x1 + x2 + x3 and w = w1 + w2 + w3 (the weight planes are prepared once at load time and arrive
// as a [3 N][taps Kp] bf16 matrix, GemmConvArgs::W3).  Of the nine plane products the six of relative weight >= 2^-16 are kept (x1 w1; x1 w2, x2 w1; x2 w2,
// x1 w3, x3 w1), every one exact, accumulated in fp32 smallest first; the dropped x2 w3 + x3 w2 + x3 w3 are <= 2^-23 of a product - the size of ONE fp32
// rounding of it, so the result is as accurate as the fp32 MFMA chain (whose own accumulation error is ~ sqrt(K) roundings) for 6 x 16 instead of
// 8 x 32 matrix-pipe cycles per 32 k.  A WX3 kernel is an AX3 kernel whose weight tile has 3 BN rows.
template <int BM, int BN, int BK, bool WBF16, bool AVEC, int STAGES = 2, bool ABF16 = false, int WM = 2, int WN = 2, bool AX3 = false, bool WX3 = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_conv_kernel(GemmConvArgs p) {
    static_assert(!ABF16 || WBF16, "the bf16 MFMA path takes bf16 weights");
    static_assert(!AX3 || (WBF16 && !ABF16), "the three-term split is the exact path for bf16 weights");
    static_assert(!WX3 || (AX3 && AVEC), "the two-sided split is an AX3 kernel over pre-split weight planes");
    constexpr int NT = 64 * WM * WN;
    constexpr int WR = WX3 ? 3 : 1;                                        // weight-tile rows per output column
    constexpr bool BFT = ABF16 || AX3;                                    // bf16 tiles in LDS, products on v_mfma_f32_16x16x32_bf16
    // LD = LDS row pitch in dwords.  fp32 tiles: BK + LDS_PAD.  bf16 tiles hold BK/2 packed pairs + LDS_PAD dwords of padding.
    constexpr int LD = BFT ? BK / 2 + LDS_PAD : BK + LDS_PAD, KV = BK / 4;   // pitch 8 (mod 16) dwords: conflict-free fragment reads (common.h); KV = groups of 4 k-values per tile row
    constexpr int APL = BM * LD;                                          // dwords per A plane (AX3: three of them)
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);   // 16x16 tiles per wave (wave tile = BM/WM x BN/WN)
    constexpr int AV = BM * KV / NT, WV = WR * BN * KV / NT;  // float4 groups per thread per k-step
    static_assert(TM >= 1 && TN >= 1 && AV >= 1 && WV >= 1 && BM * KV % NT == 0 && WR * BN * KV % NT == 0, "tile does not divide over the workgroup");
    __shared__ __attribute__((aligned(16))) float As[(AX3 ? 3 : 1) * APL];
    __shared__ __attribute__((aligned(16))) float Ws[WR * BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    // XCD-aware tile order: consecutive remapped ids walk the N tiles of one M band, so an XCD owns ~1/8 of the rows of A
    // (A crosses the fabric once) and keeps the whole, much smaller, W panel in its own L2.
    const int bl = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int m0 = (bl / (int)gridDim.y) * BM, n0 = (bl % (int)gridDim.y) * BN, b = blockIdx.z;
    const float* Ab = p.A + (long long)b * p.a_batch;
    const int nit = p.taps * ((p.Kp + BK - 1) / BK);
    const long long ldw = p.ldw ? p.ldw : (long long)p.taps * p.Kp;
    const long long wb = (long long)b * p.w_batch;

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    // STAGES register sets form the prefetch ring (2..4): every tile pays a full L2-miss round trip (its slowest line), measured at
    // several microseconds on these shapes, so the small tiles — which have the registers — keep 4 tiles in flight per wave.
    float4 ra0[AV], rw0[WV], ra1[AV], rw1[WV], ra2[STAGES > 2 ? AV : 1], rw2[STAGES > 2 ? WV : 1], ra3[STAGES > 3 ? AV : 1], rw3[STAGES > 3 ? WV : 1];

    // Loads are UNCONDITIONAL (clamped address + select): a load inside a divergent branch makes the compiler lose count of the
    // outstanding vector-memory operations and fall back to s_waitcnt vmcnt(0), which drains the prefetch ring every iteration.
    // All per-thread address arithmetic (64-bit row bases, validity of the row) is done ONCE here; a tile only adds the uniform
    // offsets of its (tap, k-chunk).  [Measured with tools/ubench/gemm_probe.hip: recomputing it per tile cost ~2500 cycles per
    // load_tile against ~1300 cycles of MFMA work — one wave per SIMD has nothing to hide VALU address math behind.]
    long long a_base[AV]; int a_c4[AV]; bool a_row_ok[AV];
    long long w_base[WV]; int w_c4[WV];
    // vectorised operands go through buffer resources: 32-bit byte offsets, hardware range check (see common.h)
    constexpr int WE = WBF16 ? 2 : 4;                 // bytes per weight element
    int a_boff[AV], w_boff[WV];
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(Ab, (unsigned)(p.a_len * 4));
    const void* Wp = WX3 ? p.W3 : p.W;
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(reinterpret_cast<const char*>(Wp) + wb * WE, (unsigned)(((long long)(WR * p.N - 1) * ldw + (long long)p.taps * p.Kp) * WE));
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int v = tid + i * NT, m = m0 + v / KV;
        a_c4[i] = (v % KV) * 4;
        a_row_ok[i] = m < p.M;
        a_base[i] = (long long)m * p.lda + p.a_off0 + a_c4[i];
        a_boff[i] = a_row_ok[i] ? (int)(a_base[i] * 4) : BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int v = tid + i * NT;
        int n = WR * n0 + v / KV; n = n < WR * p.N ? n : WR * p.N - 1;      // WX3: row 3 n + plane
        w_c4[i] = (v % KV) * 4;
        w_base[i] = wb + (long long)n * ldw + w_c4[i];
        w_boff[i] = (int)(((long long)n * ldw + w_c4[i]) * WE);
    }
    auto load_tile = [&](int tap, int k0, auto& ra, auto& rw) {
        const long long a_off = (long long)tap * p.tap_step + k0;       // uniform
        const long long w_off = (long long)tap * p.Kp + k0;             // uniform
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int kk = k0 + a_c4[i];
            const long long idx = a_base[i] + a_off;
            if (AVEC) {       // K % 4 == 0 and every float4 group is 16B aligned and entirely in or out of [0, a_len)
                ra[i] = buf_load_f4(rsA, kk < p.K ? a_boff[i] + (int)a_off * 4 : BUF_OOB);      // rows >= M, idx outside [0, a_len): hardware zero
            } else {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long ie = idx + e;
                    const bool ok = a_row_ok[i] && kk + e < p.K && ie >= 0 && ie < p.a_len;
                    const float x = Ab[ok ? ie : 0];
                    t[e] = ok ? x : 0.f;
                }
                ra[i] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const bool ok = k0 + w_c4[i] < p.Kp;                // Kp is a multiple of 32: the last k-step may be partly empty
            float4 wv;
            if (AVEC) {                                         // (the host only selects AVEC kernels when W also fits 32-bit offsets)
                const int off = ok ? w_boff[i] + (int)w_off * WE : BUF_OOB;
                if (WBF16) {
                    const uint2 u = buf_load_u2(rsW, off);
                    if (BFT) wv = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
                    else wv = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                          __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
                } else {
                    wv = buf_load_f4(rsW, off);
                }
                rw[i] = wv;
                continue;
            }
            const long long idx = ok ? w_base[i] + w_off : w_base[i];
            if (WBF16) {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.W) + idx);
                if (BFT) wv = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);      // raw pairs, staged as they are
                else wv = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                      __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
            } else {
                wv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.W) + idx);
            }
            if (!ok) wv = make_float4(0.f, 0.f, 0.f, 0.f);
            rw[i] = wv;
        }
    };
    auto store_tile = [&](int k0, const auto& ra, const auto& rw) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int v = tid + i * NT, kk = k0 + a_c4[i];
            float4 x = ra[i];
            if (p.pro != ACT_NONE) {
                if (p.pro == ACT_LEAKY) {
                    x.x = x.x > 0.f ? x.x : x.x * p.pro_p; x.y = x.y > 0.f ? x.y : x.y * p.pro_p;
                    x.z = x.z > 0.f ? x.z : x.z * p.pro_p; x.w = x.w > 0.f ? x.w : x.w * p.pro_p;
                } else if (p.pro == ACT_SNAKE) {
                    if (kk < p.K) {                             // alpha is padded to Kp by the host; Snake(0) == 0 keeps the zero padding
                        const float4 al = *reinterpret_cast<const float4*>(p.pro_alpha + kk);
                        x.x = snake_f(x.x, al.x); x.y = snake_f(x.y, al.y); x.z = snake_f(x.z, al.z); x.w = snake_f(x.w, al.w);
                    }
                } else {                                        // any other activation with act(0) == 0 (Mish, SiLU, ...)
                    x = apply_act4(p.pro, x, p.pro_p);
                }
            }
            if constexpr (AX3) {
                const unsigned a0 = pack_bf16x2(x.x, x.y), a1 = pack_bf16x2(x.z, x.w);
                x.x -= __uint_as_float(a0 << 16); x.y -= __uint_as_float(a0 & 0xffff0000u); x.z -= __uint_as_float(a1 << 16); x.w -= __uint_as_float(a1 & 0xffff0000u);
                const unsigned b0 = pack_bf16x2(x.x, x.y), b1 = pack_bf16x2(x.z, x.w);
                x.x -= __uint_as_float(b0 << 16); x.y -= __uint_as_float(b0 & 0xffff0000u); x.z -= __uint_as_float(b1 << 16); x.w -= __uint_as_float(b1 & 0xffff0000u);
                const int o = (v / KV) * LD + a_c4[i] / 2;
                *reinterpret_cast<uint2*>(&As[o]) = make_uint2(a0, a1);
                *reinterpret_cast<uint2*>(&As[APL + o]) = make_uint2(b0, b1);
                *reinterpret_cast<uint2*>(&As[2 * APL + o]) = make_uint2(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w));
            } else if (ABF16) *reinterpret_cast<uint2*>(&As[(v / KV) * LD + a_c4[i] / 2]) = make_uint2(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w));
            else *reinterpret_cast<float4*>(&As[(v / KV) * LD + a_c4[i]]) = x;
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NT;
            if (BFT) *reinterpret_cast<uint2*>(&Ws[(v / KV) * LD + w_c4[i] / 2]) = make_uint2(__float_as_uint(rw[i].x), __float_as_uint(rw[i].y));
            else *reinterpret_cast<float4*>(&Ws[(v / KV) * LD + w_c4[i]]) = rw[i];
        }
    };
    auto compute_tile = [&]() {
        if constexpr (WX3) {
            if (p.w3_terms == 3) {                              // launch-uniform: two planes of each operand, three products per k
#pragma unroll
                for (int kg = 0; kg < BK / 32; ++kg) {
                    uint4 af[2][TM], wf[2][TN];
                    const int kd = kg * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            af[pl][i] = *reinterpret_cast<const uint4*>(&As[pl * APL + (wm * (BM / WM) + i * 16 + (lane & 15)) * LD + kd]);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            wf[pl][j] = *reinterpret_cast<const uint4*>(&Ws[((wn * (BN / WN) + j * 16 + (lane & 15)) * 3 + pl) * LD + kd]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {          // smallest terms first
                            v4f a = acc[i][j];
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[0][j]), __builtin_bit_cast(v8bf, af[1][i]), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[1][j]), __builtin_bit_cast(v8bf, af[0][i]), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[0][j]), __builtin_bit_cast(v8bf, af[0][i]), a, 0, 0, 0);
                            acc[i][j] = a;
                        }
                }
                return;
            }
#pragma unroll
            for (int kg = 0; kg < BK / 32; ++kg) {
                uint4 af[3][TM], wf[3][TN];
                const int kd = kg * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        af[pl][i] = *reinterpret_cast<const uint4*>(&As[pl * APL + (wm * (BM / WM) + i * 16 + (lane & 15)) * LD + kd]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        wf[pl][j] = *reinterpret_cast<const uint4*>(&Ws[((wn * (BN / WN) + j * 16 + (lane & 15)) * 3 + pl) * LD + kd]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {              // smallest terms first
                        v4f a = acc[i][j];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[0][j]), __builtin_bit_cast(v8bf, af[2][i]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[2][j]), __builtin_bit_cast(v8bf, af[0][i]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[1][j]), __builtin_bit_cast(v8bf, af[1][i]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[0][j]), __builtin_bit_cast(v8bf, af[1][i]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[1][j]), __builtin_bit_cast(v8bf, af[0][i]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[0][j]), __builtin_bit_cast(v8bf, af[0][i]), a, 0, 0, 0);
                        acc[i][j] = a;
                    }
            }
            return;
        }
        if constexpr (AX3) {
#pragma unroll
            for (int kg = 0; kg < BK / 32; ++kg) {
                uint4 af[3][TM], wf[TN];
                const int kd = kg * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        af[pl][i] = *reinterpret_cast<const uint4*>(&As[pl * APL + (wm * (BM / WM) + i * 16 + (lane & 15)) * LD + kd]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[j] = *reinterpret_cast<const uint4*>(&Ws[(wn * (BN / WN) + j * 16 + (lane & 15)) * LD + kd]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {              // smallest term first
                        const v8bf w8 = __builtin_bit_cast(v8bf, wf[j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w8, __builtin_bit_cast(v8bf, af[2][i]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w8, __builtin_bit_cast(v8bf, af[1][i]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w8, __builtin_bit_cast(v8bf, af[0][i]), acc[i][j], 0, 0, 0);
                    }
            }
            return;
        }
        if constexpr (ABF16) {
            // v_mfma_f32_16x16x32_bf16: lane (r = lane & 15, g = lane >> 4) supplies k = 8g .. 8g+7 of row r for both operands
#pragma unroll
            for (int kg = 0; kg < BK / 32; ++kg) {
                uint4 af[TM], wf[TN];
                const int kd = kg * 16 + (lane >> 4) * 4;               // dword offset of the lane's 8 bf16
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const uint4*>(&As[(wm * (BM / WM) + i * 16 + (lane & 15)) * LD + kd]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[j] = *reinterpret_cast<const uint4*>(&Ws[(wn * (BN / WN) + j * 16 + (lane & 15)) * LD + kd]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[j]), __builtin_bit_cast(v8bf, af[i]), acc[i][j], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int kg = 0; kg < BK / 16; ++kg) {
            float4 af[TM], wf[TN];
            const int kc = kg * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const float4*>(&As[(wm * (BM / WM) + i * 16 + (lane & 15)) * LD + kc]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wf[j] = *reinterpret_cast<const float4*>(&Ws[(wn * (BN / WN) + j * 16 + (lane & 15)) * LD + kc]);
            // k-slot permutation: lane group g = lane>>4 feeds k = kc+s at MFMA step s for both operands.
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].x, af[i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].y, af[i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].z, af[i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].w, af[i].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    long long* dbg = p.dbg ? p.dbg + ((long long)(blockIdx.y * gridDim.x + blockIdx.x)) * 64 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0 && dn < 64) dbg[dn++] = clock64(); };
    // (tap, k0) of the tile being LOADED advance incrementally (no integer division in the loop); ks* shadow the tiles being stored
    int l_tap = 0, l_k0 = 0;
    auto advance = [&]() { l_k0 += BK; if (l_k0 >= p.Kp) { l_k0 = 0; ++l_tap; } };
    int ks0 = 0, ks1 = 0, ks2 = 0, ks3 = 0;
    stamp();
    load_tile(l_tap, l_k0, ra0, rw0); ks0 = l_k0; advance();
    if (nit > 1) { load_tile(l_tap, l_k0, ra1, rw1); ks1 = l_k0; advance(); }
    if constexpr (STAGES > 2) { if (nit > 2) { load_tile(l_tap, l_k0, ra2, rw2); ks2 = l_k0; advance(); } }
    if constexpr (STAGES > 3) { if (nit > 3) { load_tile(l_tap, l_k0, ra3, rw3); ks3 = l_k0; advance(); } }
    stamp();
    for (int it = 0; it < nit; it += STAGES) {
        store_tile(ks0, ra0, rw0);
        stamp();
        __syncthreads();
        if (it + STAGES < nit) { load_tile(l_tap, l_k0, ra0, rw0); ks0 = l_k0; advance(); }     // in flight for the next STAGES-1 tiles
        stamp();
        compute_tile();
        stamp();
        __syncthreads();
        if (it + 1 < nit) {
            store_tile(ks1, ra1, rw1);
            __syncthreads();
            if (it + 1 + STAGES < nit) { load_tile(l_tap, l_k0, ra1, rw1); ks1 = l_k0; advance(); }
            compute_tile();
            __syncthreads();
        }
        if constexpr (STAGES > 2) {
            if (it + 2 < nit) {
                store_tile(ks2, ra2, rw2);
                __syncthreads();
                if (it + 2 + STAGES < nit) { load_tile(l_tap, l_k0, ra2, rw2); ks2 = l_k0; advance(); }
                compute_tile();
                __syncthreads();
            }
        }
        if constexpr (STAGES > 3) {
            if (it + 3 < nit) {
                store_tile(ks3, ra3, rw3);
                __syncthreads();
                if (it + 3 + STAGES < nit) { load_tile(l_tap, l_k0, ra3, rw3); ks3 = l_k0; advance(); }
                compute_tile();
                __syncthreads();
            }
        }
    }

    stamp();
    // epilogue: lane holds C[m][n..n+3] with m = ..+(lane&15), n = ..+(lane>>4)*4   (W rows were the MFMA "A")
    float* Cb = p.C + (long long)b * p.c_batch;
    const float* Rb = p.res ? p.res + (long long)b * p.res_batch : nullptr;
    const float* RSb = p.row_scale ? p.row_scale + (long long)b * p.row_scale_batch : nullptr;
    const float* Bb = p.bias ? p.bias + (long long)b * p.bias_batch : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const float rs = RSb ? RSb[m] : 1.f;
        const float* CSr = p.col_scale ? p.col_scale + (long long)(m / p.col_scale_rows) * p.col_scale_stride : nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            const long long idx = (long long)m * p.ldc + n + p.c_off;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const bool full = p.c_vec && (n + 3 < p.N) && idx >= 0 && idx + 3 < p.c_len;
            bool ok[4];
            float bb[4] = {0.f, 0.f, 0.f, 0.f}, rr[4] = {0.f, 0.f, 0.f, 0.f}, oo[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {1.f, 1.f, 1.f, 1.f};
            if (full) {
                ok[0] = ok[1] = ok[2] = ok[3] = true;
                if (Bb) { const float4 t = *reinterpret_cast<const float4*>(Bb + n); bb[0] = t.x; bb[1] = t.y; bb[2] = t.z; bb[3] = t.w; }
                if (CSr) { const float4 t = *reinterpret_cast<const float4*>(CSr + n); cs[0] = t.x; cs[1] = t.y; cs[2] = t.z; cs[3] = t.w; }
                if (Rb) { const float4 t = *reinterpret_cast<const float4*>(Rb + idx); rr[0] = t.x; rr[1] = t.y; rr[2] = t.z; rr[3] = t.w; }
                if (p.accumulate) { const float4 t = *reinterpret_cast<const float4*>(Cb + idx); oo[0] = t.x; oo[1] = t.y; oo[2] = t.z; oo[3] = t.w; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long ie = idx + e;
                    ok[e] = (n + e < p.N) && ie >= 0 && ie < p.c_len;
                    if (ok[e]) {
                        if (Bb) bb[e] = Bb[n + e];
                        if (CSr) cs[e] = CSr[n + e];
                        if (Rb) rr[e] = Rb[ie];
                        if (p.accumulate) oo[e] = Cb[ie];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bb[e];
            if (p.act == ACT_SNAKE) {                        // Snake with a per-column alpha (padded to a multiple of 32 by the host: n + 3 is readable)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = snake_f(v[e], p.act_alpha[n + e]);
            } else if (p.act != ACT_NONE) {                  // the only activation site of the epilogue
                const float4 t = apply_act4(p.act, make_float4(v[0], v[1], v[2], v[3]), p.act_p);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] * cs[e] + rr[e]) * p.out_scale * rs + oo[e];
            if (full) {
                *reinterpret_cast<float4*>(Cb + idx) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (ok[e]) Cb[idx + e] = v[e];
            }
            if (p.C2) {                                      // the same values, activated for their consumer
                float* C2b = p.C2 + (long long)b * p.c_batch;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (ok[e]) C2b[idx + e] = snake_f(v[e], p.c2_alpha[n + e]);
            }
        }
    }
    stamp();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// M = 1 with fp32 weights: y[n] = epi(sum_k x[k] W[n][k]) - the decode step of CosyVoice-300M's TransformerLM (llm/llm.py:162-223: 14 x (qkv4, out, w1, w2) + the
// decoder, 176 M fp32 parameters = 705 MB per token).  Round 3 ran these on the tiled GEMM above: a 16- or 32-row tile with ONE useful row, 64-128 workgroups
// per launch, 1.84 ms per token (0.38 TB/s).  Here a 16-lane group owns an output row and reads it in 256-byte steps (16 B per lane, non-temporal: every
// weight byte is read once per token); the four waves of a workgroup split K between them and combine through LDS in a fixed order; a workgroup = 4 rows, so
// N = 1024 is already 256 workgroups.  All of a wave's weight loads of an 8-step group are requested before the first FMA (gemv_kernel's latency structure,
// llm_kernels.h).  The k order differs from the MFMA chain's, like any other summation order: results agree to fp32 rounding.
// ---------------------------------------------------------------------------------------------------------------------------------
struct GemvF32Args {
    const float* x; const float* W; long long ldw; const float* bias; const float* res; float* y;
    int N, K, Kp, act; float act_p, out_scale;
};
static __global__ __launch_bounds__(256) void gemv_f32_kernel(GemvF32Args p) {
    constexpr int U = 8;                                        // 64-float steps in flight per lane group
    __shared__ float part[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const int row = min((int)blockIdx.x * 4 + grp, p.N - 1);      // clamped: the reductions are wave collectives
    const int steps = (p.Kp + 63) / 64, s0 = wave * steps / 4, s1 = (wave + 1) * steps / 4;
    const float* wr = p.W + (long long)row * p.ldw;
    float acc = 0.f;
    for (int sb = s0; sb < s1; sb += U) {
        v4f w[U]; float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)                             // unconditional loads (clamped step) + select keep the vmcnt bookkeeping exact
            w[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(wr + min(min(sb + u, s1 - 1) * 64 + sub * 4, p.Kp - 4)));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(sb + u, s1 - 1) * 64 + sub * 4;
            x[u] = *reinterpret_cast<const float4*>(p.x + min(k, p.K - 4));
            if (sb + u >= s1 || k >= p.K) x[u] = make_float4(0.f, 0.f, 0.f, 0.f);          // beyond the wave's range / beyond K (the weights are zero padded to Kp, x is not)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += w[u][0] * x[u].x; acc += w[u][1] * x[u].y; acc += w[u][2] * x[u].z; acc += w[u][3] * x[u].w; }
    }
    acc = group16_sum(acc);
    if (sub == 0) part[wave][grp] = acc;
    __syncthreads();
    if (wave != 0 || sub != 0) return;
    const int n = (int)blockIdx.x * 4 + grp;
    if (n >= p.N) return;
    float v = ((part[0][grp] + part[1][grp]) + part[2][grp]) + part[3][grp];
    if (p.bias) v += p.bias[n];
    v = apply_act(p.act, v, p.act_p);
    if (p.res) v += p.res[n];
    p.y[n] = v * p.out_scale;
}

// host-side dispatch (gemm_conv.hip)
void launch_gemm_conv(const GemmConvArgs& a, bool w_bf16, int batch, hipStream_t stream);

}  // namespace cv
