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
#include <type_traits>
#include "flow_fused.h"

namespace cv {

// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm over K <= 256 channels of fp32 rows -> bf16 rows.  A 16-lane group owns a row, lane `sub` holds channels 4 sub + 64 j - the register
// layout, reduction order and expressions of flow_gemm_kernel<.., AMODE 1>'s prologue (keep the two in step: the bf16 values must be the same bits).
// ---------------------------------------------------------------------------------------------------------------------------------
struct LnBf16Args { const float* x; int ldx; const float* gamma; const float* beta; float eps; bf16_t* y; int ldy; int M, K; };

static __global__ __launch_bounds__(256) void ln_bf16_kernel(LnBf16Args p) {
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
// request (taps x K contraction, tap-major like gemm_conv_kernel's bf16 tiles, which this form replaces for the ResNet convolutions of a large pass).
// BM x BN tile per 2 x 2 waves; K in stages of 64 through a two-buffer LDS ring: stage g is multiplied out of buffer g & 1 while stage g + 1 is written into the
// other buffer and stage g + 2 is in flight in registers - one barrier per stage.  LDS rows are 64 bf16 = eight 16-byte slots without padding, slot s of row r
// stored at s ^ ((r >> 1) & 7): the fragment reads (row r, slot 4 kg + g) of a 16-lane LDS group then cover all 64 banks once (common.h, LDS_PAD note), and so
// does a row's store.
// PERSISTENT: a workgroup owns a run of consecutive tiles (the N tiles of a band of rows, then the next band) and its stage sequence runs ACROSS them - the
// first stages of tile t + 1 are loaded and parked under the last MFMAs and the epilogue of tile t.  With K = 256 a tile is four stages: launched one tile per
// workgroup (first form of this kernel, profiles/r4_flow_big_ab.txt) a tile was a latency chain of first-load wait -> 4 stages -> epilogue stores, 26 - 37 us
// per launch at M = 10 784 whatever the tile.  The grid is what fits the chip at once (host: resident workgroups per CU x 256), not the tile count.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int OMODE, bool CONV = false, bool GLDS = false>
__global__ __launch_bounds__(256) void flow_gemm_big_kernel(FlowGemmArgs p) {
    constexpr int BK = 64, RP = BK / 2;                       // row pitch in dwords
    constexpr int TM = BM / 32, TN = BN / 32;                 // 16 x 16 MFMA tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int AV = BM * 8 / 256, WV = BN * 8 / 256;       // 16-byte pieces per thread and stage
    static_assert(BM % 32 == 0 && BN % 32 == 0 && AV >= 1 && WV >= 1, "tile must split over 2 x 2 waves of 16 x 16 MFMA tiles");
    __shared__ __attribute__((aligned(16))) unsigned Ls[2 * (BM + BN) * RP];
    unsigned* const As0 = Ls; unsigned* const Ws0 = Ls + 2 * BM * RP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int ntn = (p.N + BN - 1) / BN, ntiles = ((p.M + BM - 1) / BM) * ntn;
    // this workgroup's run of tiles: consecutive in (row band, N tile) order, runs handed out in XCD order (an XCD's workgroups walk neighbouring bands: A crosses
    // the fabric once, the weights stay in its L2)
    const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;       // dev tool (flow option gemm_dbg): clock64() of thread 0 at the phase boundaries
    int dn = 0;
    auto stamp = [&]() { if (dbg && threadIdx.x == 0) dbg[dn++] = clock64(); };
    stamp();                                                                       // 0: start
    const int tq = ntiles / (int)gridDim.x, trm = ntiles % (int)gridDim.x;
    const int t_begin = wg * tq + min(wg, trm), t_count = tq + (wg < trm ? 1 : 0);
    const int spt = (p.K + BK - 1) / BK;                       // stages per tap
    const int nst = CONV ? p.taps * spt : spt;                 // stages per tile
    const int total = t_count * nst;
    const long long wpitch = CONV ? (long long)p.taps * p.Kp : p.Kp;

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    // piece v = tid + 256 i of a stage: row v / 8, slot v % 8 (k = 8 (v % 8))
    int a_lds[AV], w_lds[WV];
#pragma unroll
    for (int i = 0; i < AV; ++i) { const int v = tid + 256 * i, r = v >> 3, s = v & 7; a_lds[i] = r * RP + ((s ^ ((r >> 1) & 7)) << 2); }
#pragma unroll
    for (int i = 0; i < WV; ++i) { const int v = tid + 256 * i, r = v >> 3, s = v & 7; w_lds[i] = r * RP + ((s ^ ((r >> 1) & 7)) << 2); }
    const int ps8 = 8 * (tid & 7), pr = tid >> 3;             // this thread's slot (in bf16) and first piece row; piece i is row pr + 32 i
    // Register ring: the loads of D stages are in flight at once.  Round-4 finding (profiles/r4_flow_big_ab.txt): with ONE stage ahead a K stage cost ~1.2 us whatever it
    // did - 8 MFMAs per wave do not cover an L2 round trip under load - and tile shape, persistence, LDS-DMA staging and row-wise stores all measured neutral.
    constexpr int D = GLDS ? 1 : 2;                            // (four stages in flight measured WORSE: 118.5 vs 107 ms per pass at 8 utterances - 122 registers, three workgroups per CU instead of five)
    u32x4_t ra[D][AV], rw[D][WV];
    int a_t[CONV ? AV : 1];                                   // CONV: row of its request of every A piece of the tile being loaded (one modulo per tile, not per stage)
    // (tile, stage) of the NEXT load, advanced incrementally: no division in the loop
    int l_tile = t_begin, l_c = 0;
    int a_tile = -1;                                          // CONV: the tile a_t[] belongs to
    auto load = [&](int slot) {                                // `slot` is a compile-time constant after unrolling (the ring stays in registers)
        const bool past = l_tile >= t_begin + t_count;        // beyond the last stage: re-request it (no branch around a load: the vmcnt bookkeeping stays exact)
        const int lt = past ? t_begin + t_count - 1 : l_tile, lc = past ? nst - 1 : l_c;
        const int m0 = (lt / ntn) * BM, n0 = (lt % ntn) * BN;
        int tap = 0, kc = lc;
        if constexpr (CONV) { tap = lc / spt; kc = lc - tap * spt; }
        const int k0 = kc * BK, dr = tap - p.pad_left;        // uniform
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int m = min(m0 + pr + 32 * i, p.M - 1);
            const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (long long)m * p.lda + ps8;
            if constexpr (CONV) {                             // unconditional load (clamped row) + select: the zero padding before a request's first row
                if (a_tile != lt) a_t[i] = m % p.rows_per_batch;      // (uniform branch, once per tile)
                const int t = a_t[i], tr_ = t + dr;                   // row of its request: taps never reach into the previous request
                u32x4_t v = *reinterpret_cast<const u32x4_t*>(ap + (long long)(max(tr_, 0) - t) * p.lda + min(k0, p.K - 8 - ps8));
                if (tr_ < 0) v = (u32x4_t){0u, 0u, 0u, 0u};
                ra[slot][i] = v;
            } else ra[slot][i] = *reinterpret_cast<const u32x4_t*>(ap + min(k0, p.K - 8 - ps8));      // clamped: steps beyond K are never multiplied
        }
        if constexpr (CONV) a_tile = lt;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int n = min(n0 + pr + 32 * i, p.N - 1);
            rw[slot][i] = *reinterpret_cast<const u32x4_t*>(p.W + (long long)n * wpitch + ps8 + (CONV ? tap * p.Kp : 0) + min(k0, p.Kp - 8 - ps8));
        }
        if (!past && ++l_c == nst) { l_c = 0; ++l_tile; }
    };
    auto store = [&](int slot, int buf) {
#pragma unroll
        for (int i = 0; i < AV; ++i) *reinterpret_cast<u32x4_t*>(&As0[buf * BM * RP + a_lds[i]]) = ra[slot][i];
#pragma unroll
        for (int i = 0; i < WV; ++i) *reinterpret_cast<u32x4_t*>(&Ws0[buf * BN * RP + w_lds[i]]) = rw[slot][i];
    };
    // fragment address of this lane: row (lane & 15) of a 16-row tile, slot 4 kg + (lane >> 4), swizzled by the row (tile bases are multiples of 16)
    const int fr = lane & 15, fx = (fr >> 1) & 7, fg = lane >> 4;
    auto compute = [&](int buf, int ksteps, int n0) {
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
                // V^T section (OMODE 0): decided per 16-column MFMA tile (wave-uniform).  Activations as the MFMA "A": a lane ends with 4 consecutive ROWS of one column
                if (OMODE == 0 && n0 + wn * (BN / 2) + j * 16 >= p.n_row) {
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
    // ---- epilogues: the expressions of flow_gemm_kernel, element for element; the accumulators are cleared for the next tile
    auto epilogue = [&](int m0, int n0) {
        if constexpr (OMODE == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * (BM / 2) + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
                    if (m < p.M && n < p.N) {
                        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                        if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                        const long long idx = (long long)m * p.ldc + n;
                        if (p.res) { const float4 r = *reinterpret_cast<const float4*>(p.res + idx); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                        if constexpr (CONV) { v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f; }      // gemm_conv_kernel's epilogue ends in "+ 0" (its accumulate term): a -0 result becomes +0 there
                        *reinterpret_cast<float4*>(p.C + idx) = v;
                    }
                    acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!(n0 + wn * (BN / 2) + j * 16 >= p.n_row)) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int m = m0 + wm * (BM / 2) + i * 16 + (lane & 15);
                        const int n = n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
                        if (m < p.M && n < p.N) {
                            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                            v = apply_act4(p.act, v, 0.f);
                            *reinterpret_cast<uint2*>(p.out + (long long)m * p.ldo + n) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                        }
                        acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
                    }
                } else {
                    // V^T section: the lane holds rows m .. m + 3 (m % 4 == 0) of column n.  Same values as flow_gemm_kernel, wider stores: a 4-aligned key
                    // group of one request is 4 consecutive V^T columns (vt_col) -> one 8-byte store; an even-aligned pair 2 columns -> one 4-byte store.
                    const int n = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
                    const float bn = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int m = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4;
                        const unsigned lo = pack_bf16x2(acc[i][j][0] + bn, acc[i][j][1] + bn), hi = pack_bf16x2(acc[i][j][2] + bn, acc[i][j][3] + bn);
                        acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
                        if (m >= p.M || n >= p.N) continue;
                        const int b = m / p.rows_per_batch, t = m - b * p.rows_per_batch;
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
    };

    // GLDS: the stage goes from global memory straight into LDS (common.h, CV_GLDS16): piece v of a stage lands at LDS offset 16 v - row v / 8, PHYSICAL slot
    // v % 8 - so the lane fetches the logical slot that belongs there, (v % 8) ^ ((row >> 1) & 7): the swizzle sits on the source address (the same 128-byte line)
    auto stage = [&](int buf) {
        const int m0 = (l_tile / ntn) * BM, n0 = (l_tile % ntn) * BN;
        int tap = 0, kc = l_c;
        if constexpr (CONV) { tap = l_c / spt; kc = l_c - tap * spt; }
        const int k0 = kc * BK, dr = tap - p.pad_left;        // uniform
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int r = pr + 32 * i, ks = min(k0 + 8 * ((tid & 7) ^ ((r >> 1) & 7)), p.K - 8), m = min(m0 + r, p.M - 1);
            const bf16_t* src = reinterpret_cast<const bf16_t*>(p.A) + (long long)m * p.lda + ks;
            if constexpr (CONV) {                             // a row before its request's first one: the DMA fetches zeros
                if (l_c == 0) a_t[i] = m % p.rows_per_batch;
                src = a_t[i] + dr < 0 ? reinterpret_cast<const bf16_t*>(p.zeros) : src + (long long)dr * p.lda;
            }
            CV_GLDS16(src, &As0[buf * BM * RP + (i * 256 + wave * 64) * 4]);
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int r = pr + 32 * i, ks = min(k0 + 8 * ((tid & 7) ^ ((r >> 1) & 7)), p.Kp - 8);
            CV_GLDS16(p.W + (long long)min(n0 + r, p.N - 1) * wpitch + (CONV ? tap * p.Kp : 0) + ks, &Ws0[buf * BN * RP + (i * 256 + wave * 64) * 4]);
        }
        if (++l_c == nst) { l_c = 0; ++l_tile; }
    };

    // ---- the same epilogues with the tile staged through LDS (one tile per workgroup: the ring is free after the last stage) and stored ROW-WISE: 16 bytes per
    // lane, a tile row's 128 / 256 bytes contiguous.  Round-4 measurement (profiles/r4_flow_big_ab.txt): these launches take the time of their OUTPUT - 30 us
    // for the 33 / 22 MB of the bf16-out GEMMs, 15 us for the 11 MB of the residual ones whatever their K, tile or staging - because the accumulator layout
    // hands a lane 4 columns of one row: 8-byte stores to 16 rows per instruction, 32-byte runs (store-issue bound, guide T21).  Values unchanged.
    constexpr int CP = OMODE == 1 ? BN + 4 : BN / 2 + 4;       // dwords per staged row: fp32, or packed bf16 pairs (+4: rows stay 16-byte aligned)
    constexpr bool LDS_EPI = BM * CP <= 2 * (BM + BN) * RP;    // the staged tile fits the ring
    auto epilogue_lds = [&](int m0, int n0) {
        unsigned* Cs = Ls;
        // which 16-column MFMA tiles of this wave are row-major (the V^T section keeps its direct stores)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bool trj = OMODE == 0 && n0 + wn * (BN / 2) + j * 16 >= p.n_row;
            if (trj) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rl = wm * (BM / 2) + i * 16 + (lane & 15), cl = wn * (BN / 2) + j * 16 + (lane >> 4) * 4, n = n0 + cl;
                float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                if (p.bias && n < p.N) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if constexpr (OMODE == 1) *reinterpret_cast<float4*>(&Cs[rl * CP + cl]) = v;
                else { v = apply_act4(p.act, v, 0.f); *reinterpret_cast<uint2*>(&Cs[rl * CP + cl / 2]) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
            }
        }
        __syncthreads();
        if constexpr (OMODE == 1) {
            constexpr int CH = BN / 4;                        // 16-byte chunks (4 floats) per row
#pragma unroll
            for (int q = tid; q < BM * CH; q += 256) {
                const int rl = q / CH, cl = (q % CH) * 4, m = m0 + rl, n = n0 + cl;
                if (m >= p.M || n >= p.N) continue;
                float4 v = *reinterpret_cast<const float4*>(&Cs[rl * CP + cl]);
                const long long idx = (long long)m * p.ldc + n;
                if (p.res) { const float4 r = *reinterpret_cast<const float4*>(p.res + idx); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                if constexpr (CONV) { v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f; }
                *reinterpret_cast<float4*>(p.C + idx) = v;
            }
        } else {
            constexpr int CH = BN / 8;                        // 16-byte chunks (8 bf16) per row
            const int n_lim = min(p.N, p.n_row);              // row-major columns end here (multiple of 16)
#pragma unroll
            for (int q = tid; q < BM * CH; q += 256) {
                const int rl = q / CH, cl = (q % CH) * 8, m = m0 + rl, n = n0 + cl;
                if (m >= p.M || n >= n_lim) continue;
                *reinterpret_cast<uint4*>(p.out + (long long)m * p.ldo + n) = *reinterpret_cast<const uint4*>(&Cs[rl * CP + cl / 2]);
            }
            // V^T section: direct stores, as in epilogue()
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!(n0 + wn * (BN / 2) + j * 16 >= p.n_row)) continue;
                const int n = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
                const float bn = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4;
                    const unsigned lo = pack_bf16x2(acc[i][j][0] + bn, acc[i][j][1] + bn), hi = pack_bf16x2(acc[i][j][2] + bn, acc[i][j][3] + bn);
                    if (m >= p.M || n >= p.N) continue;
                    const int b = m / p.rows_per_batch, t = m - b * p.rows_per_batch;
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
    };
    const bool lds_epi = LDS_EPI && t_count == 1 && p.lds_epilogue;     // (uniform; persistent runs keep the next tile's stages in the ring)

    if (total == 0) return;
    int c = 0, tile = t_begin;                                 // (tile, stage) being multiplied
    if constexpr (GLDS) {
        stage(0);
        CV_VMCNT0();
        __syncthreads();
        for (int g = 0; g < total; ++g) {
            if (g + 1 < total) stage((g + 1) & 1);            // buffer (g + 1) & 1 was last read by stage g - 1: every wave left it before the barrier that ended it
            const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
            compute(g & 1, min(BK, p.K - (CONV ? c % spt : c) * BK) / 32, n0);
            if (++c == nst) { if (!lds_epi) epilogue(m0, n0); c = 0; ++tile; }
            CV_VMCNT0();                                      // the DMA of the next stage (requested before this stage's MFMAs) has landed; the barrier publishes it
            __syncthreads();
        }
        if (lds_epi) epilogue_lds((t_begin / ntn) * BM, (t_begin % ntn) * BN);
        return;
    }
    if constexpr (!GLDS) {
#pragma unroll
        for (int k = 0; k < D; ++k) load(k);                      // stages 0 .. D - 1 requested at once
        // stage g: its registers (ring slot g % D, requested D stages ago) -> LDS buffer g & 1, stage g + D requested into the freed slot, barrier, MFMAs.  Buffer g & 1
        // was last read by stage g - 2, which every wave finished before the barrier of stage g - 1.  The stage count is rounded up to D: the extra ones
        // re-request the last stage, park it and multiply nothing.
        for (int g0 = 0; g0 < total; g0 += D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int g = g0 + k;
                store(k, g & 1);
                load(k);
                __syncthreads();
                if (g == 0) stamp();                                  // 1: first stage parked (first-load latency)
                if (g < total) {
                    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
                    compute(g & 1, min(BK, p.K - (CONV ? c % spt : c) * BK) / 32, n0);
                    if (++c == nst) { if (!lds_epi) epilogue(m0, n0); c = 0; ++tile; }   // stores only: the next tile's stages are already in flight
                }
            }
        }
        __syncthreads();                                          // every wave has left the ring (the staged epilogue reuses it)
        stamp();                                                  // 2: all stages multiplied
    }
    if (lds_epi) epilogue_lds((t_begin / ntn) * BM, (t_begin % ntn) * BN);
    stamp();                                                      // 3: output stored (with per-lane stores the epilogue sits inside the stage loop: 2 -> 3 is empty)
    stamp();
}

}  // namespace cv
