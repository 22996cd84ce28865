// Fused bf16 pipeline of the flow estimator's transformer blocks (matcha BasicTransformerBlock inside CausalConditionalDecoder,
// cosyvoice/flow/decoder.py:405-494) for gfx950, "bf16" precision mode only (the fp32 mode keeps the exact-fp32 kernels).
//
// Per block the round-1 path launched LN, QKV, attention, out, LN, FF1, FF2 with fp32 activations between them (7 launches, every
// operand rounded to bf16 while it was staged).  Here the rounding points are the same, but the rounded value is what travels:
//   flow_gemm<LN prologue, bf16 out>   x (fp32 residual stream) -> LayerNorm in the prologue -> Q | K (bf16, row-major) and V^T (bf16,
//                                      transposed + key-permuted the way the P.V MFMA wants it), or -> GELU(FF1) (bf16)
//   attn_flow                          bf16 Q / K / V^T in, flash attention with fp32 softmax, bf16 out
//   flow_gemm<bf16 A, fp32 out + res>  out-projection and FF2, accumulated onto the fp32 residual stream
// 5 launches per block, half the activation bytes, no conversion or transposition work inside the attention's K/V loop.
// All products: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Statistics, softmax, residuals: fp32.
#pragma once
#include "common.h"
#include "attention.h"

namespace cv {

struct FlowGemmArgs {
    const void* A; int lda;                       // AMODE 0: bf16 [M][lda]     AMODE 1: fp32 [M][lda]
    const float* gamma; const float* beta; float eps;   // AMODE 1: LayerNorm over the K channels in the prologue (gamma == null: none)
    const bf16_t* W; int Kp;                      // [N][Kp] bf16 (cosyvoice_amd/weights.py layout), K % 32 == 0 so Kp == K
    const float* bias;                            // [N] or null
    int M, N, K;
    int act;                                      // OMODE 0: epilogue activation
    bf16_t* out; int ldo; int n_row;              // OMODE 0: columns [0, n_row) -> out[m][n] (row-major bf16)
    bf16_t* outT; long long t_batch; int ldt; int rows_per_batch;   // columns [n_row, N) -> outT[m / rpb][n - n_row][perm(m % rpb)]
    float* C; int ldc; const float* res;          // OMODE 1: C[m][n] = acc + bias (+ res[m][n]), fp32
    long long* dbg;                               // dev tool (tools/ubench/flow_gemm_probe.hip): clock64() of thread 0 at the phase boundaries, 8 slots per workgroup; null in production
    int lds_epilogue;                             // flow_gemm_big_kernel: 1 = the output tile goes through LDS and is stored row-wise, 16 bytes per lane (one tile per workgroup only)
    const void* zeros;                            // flow_gemm_big_kernel<.., CONV, GLDS>: 16 readable zero bytes (what the DMA of a padded row fetches)
    int taps, pad_left;                           // flow_gemm_big_kernel<.., CONV>: causal Conv1d over channel-last rows, W rows [taps][Kp]; tap j of output row m reads input row
                                                  // m + j - pad_left of the SAME request (rows_per_batch rows each), zero before the request's first row
};

// key (time) index -> column of the transposed V tile: inside every 32-key block, key 16 s + 4 g + r sits at column 8 g + 4 s + r, which
// makes the 8 bf16 a lane reads for the P.V MFMA (k-slot g) exactly the keys whose probabilities that lane holds after S^T = K.Q^T.
__device__ __forceinline__ int vt_col(int t) { return (t & ~31) + ((t >> 2) & 3) * 8 + ((t >> 4) & 1) * 4 + (t & 3); }

// BM x BN output tile per workgroup (256 threads, 2 x 2 waves), K staged in chunks of 256 (the whole K for the LN-prologue GEMMs).
// NTILE (AMODE 1 only, round 3): a workgroup walks NTILE consecutive BN-column tiles of the SAME rows - the LayerNorm prologue and the A tile in LDS
// are paid once, the next tile's weights are requested before the current tile's MFMAs and land under them and under its epilogue.  The QKV GEMM
// at T = 674 (1032 single-tile workgroups = 1.34 rounds of the 768 resident ones) becomes 516 two-tile workgroups in ONE round.
template <int BM, int BN, int AMODE, int OMODE, int NTILE = 1>
__global__ __launch_bounds__(256) void flow_gemm_kernel(FlowGemmArgs p) {
    constexpr int KC = 256, LDK = KC / 2 + LDS_PAD;           // LDS row pitch in dwords (bf16 pairs + padding: 8 mod 16, common.h)
    constexpr int TM = BM / 32, TN = BN / 32;                 // 16 x 16 MFMA tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int AR = BM / 16;                               // AMODE 1: rows per 16-lane group
    constexpr int AV = BM / 8, WV = BN / 8;                   // 16-byte chunks per thread and K chunk (AMODE 0 A tile / W tile)
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tile must split over 2 x 2 waves of 16 x 16 MFMA tiles");
    __shared__ __attribute__((aligned(16))) unsigned As[BM * LDK];
    __shared__ __attribute__((aligned(16))) unsigned Ws[BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    static_assert(NTILE == 1 || AMODE == 1, "several N tiles per workgroup: LayerNorm-prologue GEMMs only");
    const int ntn = (p.N + BN * NTILE - 1) / (BN * NTILE);
    const int bl = xcd_remap((int)blockIdx.x, (int)gridDim.x);        // an XCD walks the N tiles of a band of rows: A crosses the fabric once
    const int m0 = (bl / ntn) * BM;
    int n0 = (bl % ntn) * BN * NTILE;                                 // first column of the CURRENT tile
    const int nchunks = (p.K + KC - 1) / KC;
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();
    // V^T section (OMODE 0): decided per 16-column MFMA tile (wave-uniform), so that n_row only has to be a multiple of 16
    bool tr[TN];
    v4f acc[TM][TN];
    auto begin_tile = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) tr[j] = OMODE == 0 && n0 + wn * (BN / 2) + j * 16 >= p.n_row;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    };
    begin_tile();

    // ---- W chunk: thread t stages 16-byte pieces v = t + 256 i: row v / 32, k = 8 (v % 32)
    u32x4_t rw[WV];
    auto load_w = [&](int kc0, int nb = -1) {
        const int nbase = nb < 0 ? n0 : nb;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + 256 * i, n = min(nbase + v / 32, p.N - 1), k = kc0 + 8 * (v % 32);
            rw[i] = *reinterpret_cast<const u32x4_t*>(p.W + (long long)n * p.Kp + min(k, p.Kp - 8));      // clamped: steps beyond Kp are never multiplied
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WV; ++i) { const int v = tid + 256 * i; *reinterpret_cast<u32x4_t*>(&Ws[(v / 32) * LDK + 4 * (v % 32)]) = rw[i]; }
    };
    // ---- A chunk, AMODE 0 (bf16 in memory): same piece mapping as W
    u32x4_t ra[AMODE == 0 ? AV : 1];
    auto load_a = [&](int kc0) {
        if constexpr (AMODE == 0) {
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                const int v = tid + 256 * i, m = min(m0 + v / 32, p.M - 1), k = kc0 + 8 * (v % 32);
                ra[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(p.A) + (long long)m * p.lda + min(k, p.K - 8));
            }
        }
    };
    auto store_a = [&]() {
        if constexpr (AMODE == 0) {
#pragma unroll
            for (int i = 0; i < AV; ++i) { const int v = tid + 256 * i; *reinterpret_cast<u32x4_t*>(&As[(v / 32) * LDK + 4 * (v % 32)]) = ra[i]; }
        }
    };
    auto compute = [&](int ksteps) {
        for (int kg = 0; kg < ksteps; ++kg) {                 // k-step = 32 values = 16 dwords; lane (r = lane & 15, g = lane >> 4) supplies k = 8 g .. 8 g + 7
            uint4 af[TM], wf[TN];
            const int kd = kg * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(&As[(wm * (BM / 2) + i * 16 + (lane & 15)) * LDK + kd]);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const uint4*>(&Ws[(wn * (BN / 2) + j * 16 + (lane & 15)) * LDK + kd]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (tr[j]) {                                  // activations are the MFMA "A": a lane ends with 4 consecutive ROWS m of one column n
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af[i]), __builtin_bit_cast(v8bf, wf[j]), acc[i][j], 0, 0, 0);
                } else {                                      // weights are the MFMA "A": a lane ends with 4 consecutive COLUMNS n of one row m
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wf[j]), __builtin_bit_cast(v8bf, af[i]), acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    // ---- epilogue (of the current tile: n0)
    auto epilogue = [&]() {
        if constexpr (OMODE == 1) {
    #pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * (BM / 2) + i * 16 + (lane & 15);
                if (m >= p.M) continue;
    #pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
                    if (n >= p.N) continue;                       // N % 4 == 0 (host check): a group of 4 columns is entirely in or out
                    float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                    const long long idx = (long long)m * p.ldc + n;
                    if (p.res) { const float4 r = *reinterpret_cast<const float4*>(p.res + idx); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
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
                    // V^T section: lane holds rows m .. m + 3 of column n.  Rows of one request are consecutive (m = b rows_per_batch + t) but a group
                    // of four may straddle a 4-aligned key group or two requests when rows_per_batch % 4 != 0, so every element finds its own slot.
                    const int n = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
                    if (n >= p.N) continue;
                    const float bn = p.bias ? p.bias[n] : 0.f;
    #pragma unroll
                    for (int i = 0; i < TM; ++i)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
                            if (m >= p.M) continue;
                            const int b = m / p.rows_per_batch, t = m - b * p.rows_per_batch;
                            const unsigned u = pack_bf16x2(acc[i][j][r] + bn, 0.f);
                            p.outT[(long long)b * p.t_batch + (long long)(n - p.n_row) * p.ldt + vt_col(t)] = (bf16_t)(u & 0xffffu);
                        }
                }
            }
        }
    };

    if constexpr (AMODE == 1) {
        // Whole K (<= 256) at once.  A 16-lane group owns AR rows; lane `sub` holds channels 4 sub + 64 j.  Every load of the tile (W included)
        // is requested before the first reduction; the LayerNorm statistics are two-pass on registers (mean, then centred variance,
        // like torch.nn.LayerNorm and norm_rows_kernel), reduced over the 16 lanes with DPP row operations.
        const int grp = tid >> 4, sub = tid & 15;
        float4 x[AR][4];
        load_w(0);
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int m = min(m0 + grp + 16 * r, p.M - 1);
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
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<uint2*>(&As[(grp + 16 * r) * LDK + 2 * sub + 32 * j]) = make_uint2(pack_bf16x2(x[r][j].x, x[r][j].y), pack_bf16x2(x[r][j].z, x[r][j].w));
        }
        store_w();
        stamp();
        __syncthreads();
        stamp();
        if constexpr (NTILE > 1) {
#pragma unroll
            for (int t = 0; t + 1 < NTILE; ++t) {
                load_w(0, n0 + BN);                           // the next tile's weights: in flight under this tile's MFMAs and epilogue
                compute(p.K / 32);
                epilogue();
                __syncthreads();                              // every wave has read this tile's Ws
                store_w();
                n0 += BN;
                begin_tile();
                __syncthreads();
            }
        }
        compute(p.K / 32);
    } else {
        // K streamed in chunks of 256: the next chunk's loads are in flight under the MFMAs of the current one (register prefetch)
        load_a(0); load_w(0);
        for (int c = 0; c < nchunks; ++c) {
            store_a(); store_w();
            __syncthreads();
            if (c + 1 < nchunks) { load_a((c + 1) * KC); load_w((c + 1) * KC); }
            compute(min(KC, p.K - c * KC) / 32);
            __syncthreads();
        }
        stamp(); stamp();
    }
    stamp();

    epilogue();
    stamp();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Flash attention over bf16 Q / K (row-major [B * T][ld]) and V^T ([B][H * 64][Tp], key-permuted, see vt_col), head_dim 64, bf16 out.
// One workgroup = NW waves = NW * 16 queries of one (request, head); K and V^T stream through a DOUBLE-buffered LDS ring in 64-key tiles
// with a register prefetch stage: per tile one barrier, four 16-byte loads and four 16-byte LDS stores per thread - no conversion, no
// transposition (the QKV GEMM epilogue delivered both).  Scores / softmax / accumulator fp32; masks are index arithmetic (none / chunk).
// The S^T = K.Q^T -> P -> O^T += V^T.P^T register dataflow is the one of attention_bf16_kernel (attention.h).
// ---------------------------------------------------------------------------------------------------------------------------------
struct AttnFlowArgs {
    const bf16_t* q; const bf16_t* k; int ld;            // q / k: [B * T][ld], head h at column h * 64
    const bf16_t* vt; long long vt_batch; int ldt;        // vt: [B][H * 64][ldt]
    bf16_t* o; int ldo;                                   // [B * T][ldo]
    int B, H, T; float scale; int mask_mode; int chunk;
    const int* klen;                                      // optional [B] key counts of a padded batch
    long long* dbg;                                       // dev tool (flow option attn_dbg): clock64() of thread 0 at the phase boundaries, 8 slots per workgroup; null in production
};

template <int NW, int KT, int KS = 1, int QG = 1>
__global__ __launch_bounds__(NW * KS * 64) void attn_flow_kernel(AttnFlowArgs p) {
    // KT = 64-key tiles per iteration: with one workgroup per CU (176 of them at T = 674) each SIMD runs ONE wave, so nothing overlaps the
    // dependent chain MFMA -> scale -> row max (2 cross-lane exchanges) -> exp2 -> pack -> MFMA of a tile but the tile's own independent
    // work: KT = 2 halves the number of serial softmax steps and doubles the independent MFMA chains in flight.
    // KS = key splits inside the workgroup: the KT tiles of an iteration are shared out over KS wave sets (wave = query group + NW * split), each
    // running its own online softmax over its tiles; the sets are merged through LDS at the end.  The problem is too small for the chip
    // (674 query waves for 1024 SIMDs at T = 674), so the way to shorten a workgroup's serial chain is to put MORE waves on the same queries.
    // QG (round 4) = groups of 16 queries per wave.  At QG = 1 a wave fetches 16 KB of K / V^T fragments from LDS per 16 MFMAs: with every CU busy (passes
    // shared by several utterances) the kernel is bound by LDS reads, not by the matrix pipe.  QG = 2 gives each fragment to two MFMAs (two query groups'
    // B operands): 128 queries per workgroup, half the LDS bytes per flop.  Per query the operations and their order do not depend on QG - a wave stops
    // its key loop where the 64-query workgroup of QG = 1 holding the same queries would (kend_w) - so results are bit-identical across QG.
    constexpr int NT = NW * KS * 64, BQ = NW * 16 * QG, BKV = 64 * KT, LDH = 32 + LDS_PAD, LDV = BKV / 2 + LDS_PAD;     // LDS row pitches in dwords (8 mod 16: conflict-free fragment reads)
    constexpr int KTW = KT / KS;                          // 64-key tiles per wave and iteration
    constexpr int NI = BKV * 8 / NT;                      // 16-byte pieces per thread and operand tile (BKV x 64 bf16 each)
    static_assert(KT % KS == 0 && KTW >= 1 && BKV * 8 % NT == 0, "attn_flow_kernel: tile / split configuration");
    static_assert(QG == 1 || (QG == 2 && NW == 4), "attn_flow_kernel: QG = 2 pairs the two 64-query blocks of the QG = 1 workgroup shape (4 waves)");
    __shared__ __attribute__((aligned(16))) unsigned Ks[2][BKV * LDH];
    __shared__ __attribute__((aligned(16))) unsigned Vt[2][64 * LDV];

    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, wave = wave_all % NW, ks = wave_all / NW;
    const int lq = lane & 15, lg = lane >> 4;
    const int nqb = (p.T + BQ - 1) / BQ, nbl = gridDim.x;
    const int bl = xcd_remap((int)blockIdx.x, nbl);       // the query tiles of one (request, head) share an XCD: its K / V^T stream through ONE L2
    const int qb = bl % nqb, h = (bl / nqb) % p.H, b = bl / (nqb * p.H);
    const float NEG_INF = -__builtin_huge_valf();
    const float scale2 = p.scale * 1.4426950408889634f;
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = clock64(); };
    stamp();                                               // 0: start

    const bf16_t* kb = p.k + (long long)b * p.T * p.ld + h * 64;
    const bf16_t* vb = p.vt + (long long)b * p.vt_batch + (long long)h * 64 * p.ldt;
    const int Tkb = p.klen ? min(p.T, p.klen[b]) : p.T;        // keys of THIS batch row (see AttnArgs::klen)
    // end of the key loop of the QG = 1 workgroup (NW * 16 queries from q0) - the unit the tile sequence of a query is defined by
    auto kend_of = [&](int q0) {
        if (q0 >= p.T) return 0;
        return p.mask_mode == MASK_CHUNK ? min(Tkb, (min(p.T - 1, q0 + NW * 16 - 1) / p.chunk + 1) * p.chunk) : Tkb;
    };
    int kend = kend_of(qb * BQ), kend_w = kend;            // workgroup / this wave
    if constexpr (QG == 2) { kend_w = kend_of(qb * BQ + (wave >> 1) * 64); kend = max(kend, kend_of(qb * BQ + 64)); }
    int qi[QG], klim[QG]; bool qvalid[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        qi[g] = qb * BQ + (wave * QG + g) * 16 + lq;
        qvalid[g] = qi[g] < p.T;
        klim[g] = Tkb;
        if (p.mask_mode == MASK_CHUNK) klim[g] = min(Tkb, (qi[g] / p.chunk + 1) * p.chunk);
        if (!qvalid[g]) klim[g] = 0;
    }

    int klim_min_w = klim[0];                              // smallest key limit of the wave's queries: tiles that end below it need no per-element mask
#pragma unroll
    for (int g = 1; g < QG; ++g) klim_min_w = min(klim_min_w, klim[g]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) klim_min_w = min(klim_min_w, __shfl_xor(klim_min_w, o));
    klim_min_w = __builtin_amdgcn_readfirstlane(klim_min_w);       // uniform after the reduction: in a scalar register the tile test below is a scalar branch

    // stage pieces: K piece v -> key row v / 8, 8 bf16 at column 8 (v % 8);  V^T piece v -> d row v / (BKV / 8), 8 keys at column 8 (v % (BKV / 8))
    u32x4_t rk[NI], rv[NI];
    auto load_kv = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int v = tid + NT * i;
            rk[i] = *reinterpret_cast<const u32x4_t*>(kb + (long long)min(kt0 + (v >> 3), p.T - 1) * p.ld + (v & 7) * 8);     // keys >= T are masked below
            const int vr = v / (BKV / 8), vc = (v % (BKV / 8)) * 8;
            rv[i] = *reinterpret_cast<const u32x4_t*>(vb + (long long)vr * p.ldt + min(kt0 + vc, p.ldt - 8));   // ldt is a multiple of 64; pad columns are finite
        }
    };
    auto store_kv = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int v = tid + NT * i;
            *reinterpret_cast<u32x4_t*>(&Ks[buf][(v >> 3) * LDH + (v & 7) * 4]) = rk[i];
            *reinterpret_cast<u32x4_t*>(&Vt[buf][(v / (BKV / 8)) * LDV + (v % (BKV / 8)) * 4]) = rv[i];
        }
    };
    if (kend > 0) load_kv(0);
    // Q as the B operand of S^T = K.Q^T: lane (q = lq, g = lg) supplies d = 32 dg + 8 g .. + 7
    uint4 qf[QG][2];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const bf16_t* qp = p.q + ((long long)b * p.T + (qvalid[g] ? qi[g] : 0)) * p.ld + h * 64;
#pragma unroll
        for (int dg = 0; dg < 2; ++dg) {
            uint4 t = *reinterpret_cast<const uint4*>(qp + dg * 32 + lg * 8);
            if (!qvalid[g]) t = make_uint4(0u, 0u, 0u, 0u);
            qf[g][dg] = t;
        }
    }
    float m_run[QG], l_run[QG];
    v4f acc[QG][4];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        m_run[g] = NEG_INF; l_run[g] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[g][d] = (v4f){0.f, 0.f, 0.f, 0.f};
    }

    auto compute = [&](int kt0, int buf) {
        const int koff = ks * KTW * 64;                    // this wave set's keys inside the staged block
        v4f s[QG][4 * KTW];                                // s[g][kt][r] <-> key kt0 + koff + kt*16 + lg*4 + r, query lq of group g
#pragma unroll
        for (int kt = 0; kt < 4 * KTW; ++kt) {
            v4f sa[QG];
#pragma unroll
            for (int g = 0; g < QG; ++g) sa[g] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) {
                const uint4 kf = *reinterpret_cast<const uint4*>(&Ks[buf][(koff + kt * 16 + lq) * LDH + dg * 16 + lg * 4]);
#pragma unroll
                for (int g = 0; g < QG; ++g)
                    sa[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, kf), __builtin_bit_cast(v8bf, qf[g][dg]), sa[g], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < QG; ++g) s[g][kt] = sa[g];
        }
        uint4 pb[QG][2 * KTW];
        // Softmax arithmetic, round 4: the VALUES are those of rounds 2-3 (x = masked ? -inf : s * scale2; e = exp2f(x - m), 0 for a masked key or an all-masked
        // row), computed with ~5 instead of ~14 VALU instructions per score - the kernel was bound by them, not by the matrix pipe:
        //  * a tile entirely below every query's key limit (wave-uniform test) skips the per-element compare + select;
        //  * a masked key needs no test of its own in the exponential: x = -inf gives exp2f(-inf - m) = +0 by itself, and an all-masked row (m = -inf, x - m = NaN)
        //    subtracts 0 instead;
        //  * exp2f() is v_exp_f32 wrapped in a range fix for results below 2^-126 (compare, two selects, add, ldexp per element): when no score of the
        //    wave is more than 126 below its row maximum (one min per row, one vote per tile) the bare v_exp_f32 returns the same bits.
        const bool edge = kt0 + koff + KTW * 64 > klim_min_w;
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            float mt = NEG_INF, mn = -NEG_INF;
            if (edge) {
#pragma unroll
                for (int kt = 0; kt < 4 * KTW; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt0 + koff + kt * 16 + lg * 4 + r;
                        const float x = key < klim[g] ? s[g][kt][r] * scale2 : NEG_INF;     // log2 units: softmax on v_exp_f32
                        s[g][kt][r] = x;
                        mt = fmaxf(mt, x); mn = fminf(mn, x);
                    }
            } else {
                // round 5: four scores per multiply on the packed fp32 pipe (v_pk_mul_f32: the same products), maxima / minima as a tree (exact, order-free)
#pragma unroll
                for (int kt = 0; kt < 4 * KTW; ++kt) {
                    const v4f x = s[g][kt] * scale2;
                    s[g][kt] = x;
                    mt = fmaxf(mt, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]))); mn = fminf(mn, fminf(fminf(x[0], x[1]), fminf(x[2], x[3])));
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run[g], mt);
            const float alpha = (m_run[g] == NEG_INF) ? 0.f : exp2f(m_run[g] - m_new);
            const float m_sub = (m_new == NEG_INF) ? 0.f : m_new;
            float rsum = 0.f;
            if (__any(mn - m_sub < -126.f)) {
#pragma unroll
                for (int kt = 0; kt < 4 * KTW; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = exp2f(s[g][kt][r] - m_sub);
                        s[g][kt][r] = e;
                        rsum += e;                         // the denominator sums the UNROUNDED probabilities
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 4 * KTW; ++kt) {
                    const v4f d = s[g][kt] - m_sub;                                  // packed subtraction (same differences)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(d[r]);
                        s[g][kt][r] = e;
                        rsum += e;                         // the denominator keeps its order of additions
                    }
                }
            }
            rsum += __shfl_xor(rsum, 16);
            rsum += __shfl_xor(rsum, 32);
            l_run[g] = l_run[g] * alpha + rsum;
            m_run[g] = m_new;
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[g][d] = acc[g][d] * alpha;
#pragma unroll
            for (int blk = 0; blk < 2 * KTW; ++blk)        // k-slot (g, 4 s + r) of 32-key block blk is key 32 blk + 16 s + 4 g + r on both operands
                pb[g][blk] = make_uint4(pack_bf16x2(s[g][2 * blk][0], s[g][2 * blk][1]), pack_bf16x2(s[g][2 * blk][2], s[g][2 * blk][3]),
                                        pack_bf16x2(s[g][2 * blk + 1][0], s[g][2 * blk + 1][1]), pack_bf16x2(s[g][2 * blk + 1][2], s[g][2 * blk + 1][3]));
            if constexpr (QG > 1) __builtin_amdgcn_sched_barrier(0);     // one query group's softmax at a time: interleaved, the two keep 64 compare masks and both score sets live
        }
#pragma unroll
        for (int blk = 0; blk < 2 * KTW; ++blk)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint4 vf = *reinterpret_cast<const uint4*>(&Vt[buf][(dt * 16 + lq) * LDV + (ks * 2 * KTW + blk) * 16 + lg * 4]);
#pragma unroll
                for (int g = 0; g < QG; ++g)
                    acc[g][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, vf), __builtin_bit_cast(v8bf, pb[g][blk]), acc[g][dt], 0, 0, 0);
            }
    };

    // ring: tile t is multiplied out of buffer t & 1 while tile t + 1 is parked in the other buffer and tile t + 2 is in flight in registers
    if (kend > 0) { store_kv(0); if (BKV < kend) load_kv(BKV); }
    __syncthreads();
    stamp();                                               // 1: the first K / V^T tile is parked (first-load latency)
    for (int kt0 = 0, t = 0; kt0 < kend; kt0 += BKV, ++t) {
        if (QG == 1 || kt0 < kend_w) compute(kt0, t & 1);
        if (t == 0) stamp();                               // 2: first tile multiplied  // QG = 2: wave-uniform - the 64-query workgroup of these queries ends its loop here
        if (kt0 + BKV < kend) {
            store_kv((t + 1) & 1);                          // every wave left buffer (t + 1) & 1 before the barrier that ended iteration t - 1
            if (kt0 + 2 * BKV < kend) load_kv(kt0 + 2 * BKV);
        }
        __syncthreads();
    }

    stamp();                                               // 3: key loop done
    if constexpr (KS > 1) {
        // merge the key splits of every query group (fixed order): sets 1 .. KS-1 park (max, denominator, numerators) in LDS - the K ring is free
        // after the loop's last barrier - and set 0 combines.  Row pitch 19 floats per lane and query group.
        float* mg = reinterpret_cast<float*>(&Ks[0][0]);
        static_assert(2 * BKV * LDH >= (KS - 1) * NW * 64 * 19 * QG, "merge scratch does not fit the K ring");
        if (ks > 0) {
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                float* d = mg + (((ks - 1) * NW * 64 + wave * 64 + lane) * QG + g) * 19;
                d[0] = m_run[g]; d[1] = l_run[g];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { d[2 + dt * 4 + 0] = acc[g][dt][0]; d[2 + dt * 4 + 1] = acc[g][dt][1]; d[2 + dt * 4 + 2] = acc[g][dt][2]; d[2 + dt * 4 + 3] = acc[g][dt][3]; }
            }
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int o2 = 1; o2 < KS; ++o2) {
                const float* d = mg + (((o2 - 1) * NW * 64 + wave * 64 + lane) * QG + g) * 19;
                const float m2 = d[0], l2 = d[1];
                const float m_new = fmaxf(m_run[g], m2);
                const float a1 = (m_run[g] == NEG_INF) ? 0.f : exp2f(m_run[g] - m_new), a2 = (m2 == NEG_INF) ? 0.f : exp2f(m2 - m_new);
                l_run[g] = l_run[g] * a1 + l2 * a2;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    acc[g][dt][0] = acc[g][dt][0] * a1 + d[2 + dt * 4 + 0] * a2; acc[g][dt][1] = acc[g][dt][1] * a1 + d[2 + dt * 4 + 1] * a2;
                    acc[g][dt][2] = acc[g][dt][2] * a1 + d[2 + dt * 4 + 2] * a2; acc[g][dt][3] = acc[g][dt][3] * a1 + d[2 + dt * 4 + 3] * a2;
                }
                m_run[g] = m_new;
            }
    }
#pragma unroll
    for (int g = 0; g < QG; ++g)
        if (qvalid[g]) {
            const float inv = l_run[g] > 0.f ? 1.f / l_run[g] : 0.f;
            bf16_t* op = p.o + ((long long)b * p.T + qi[g]) * p.ldo + h * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<uint2*>(op + dt * 16 + lg * 4) = make_uint2(pack_bf16x2(acc[g][dt][0] * inv, acc[g][dt][1] * inv), pack_bf16x2(acc[g][dt][2] * inv, acc[g][dt][3] * inv));
        }
    stamp();                                               // 4: merged and stored
}

}  // namespace cv
