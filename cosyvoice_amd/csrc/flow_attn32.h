// Flash attention of the flow estimator's transformer blocks (matcha BasicTransformerBlock -> diffusers Attention inside CausalConditionalDecoder,
// cosyvoice/flow/decoder.py:405-494), bf16 mode, on v_mfma_f32_32x32x16_bf16 - round 6, the replacement of attn_flow_kernel (flow_fused.h) on every bf16 pass.
//
// Why a second kernel.  attn_flow_kernel (16 queries per wave on 16x16x32 tiles, K / V^T staged through registers + ds_write, two xor-shuffles through the LDS
// crossbar per row reduction, a rescale of the accumulators on every tile) took 43.9 us for the 14.9 GFLOP of an 8-utterance pass = 0.136 of the bf16 MFMA peak
// (VERDICT r5).  tools/ubench/valu_rate (profiles/r6_valu_rate.txt) priced what a softmax is made of on this chip: a plain fp32 VALU instruction costs a SIMD
// 1.2-2.3 ns, v_exp_f32 3.6 ns (a HALF-rate instruction that shares the issue port - exponentials as polynomials cannot win), ds_bpermute 10 ns, v_permlane32_swap
// 3.5 ns, and the 32x32x16 MFMA delivers 1.65 x the flops per issue slot of the 16x16x32 form.  So this kernel
//   * gives a wave 32 queries: S^T = K.Q^T as 32x32 tiles leaves a lane holding 16 keys of ONE query per tile - row maxima / sums are in-lane trees plus ONE
//     v_permlane32_swap (lane <-> lane + 32), and S^T's accumulator layout is the B operand of O^T += V^T.P^T once the K rows are read in the order
//     (bits 3 and 4 of the row index swapped) that makes it the key order of the V^T layout the QKV epilogues already write (vt_col, flow_fused.h);
//   * brings K and V^T tiles in by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a 3-stage ring, one barrier per 64-key tile, the DMA
//     two tiles ahead behind a COUNTED vmcnt; the 128-byte tile rows are XOR-swizzled through the DMA's source addresses (chunk ^= (row >> 1) & 7), which makes
//     every ds_read_b128 fragment fetch conflict-free at a 128-byte pitch;
//   * one fma + one bare v_exp_f32 per score (exp2(s * c - m * c)), the scaling folded into the fma;
//   * LAZY rescale: a lane moves its running maximum only when the tile's maximum exceeds it by more than 2^8 (in the exponent: probabilities then stay below 256,
//     which costs bf16 no relative precision), and the accumulators are multiplied only in tiles where some lane of the wave moved - by exactly 1.0 in the lanes
//     that did not, so a query's bits do not depend on its wave mates (padded batch rows, other requests: the batch = single bit-identity contracts hold).
// Per query the operations and their order depend on (T, klen, mask) alone - not on the workgroup shape (NW) and not on what else shares the launch.
// Scores / softmax / accumulators fp32; P rounded to bf16 for the second product (as in attn_flow_kernel); the denominator sums the UNROUNDED probabilities.
#pragma once
#include "common.h"
#include "attention.h"
#include "flow_fused.h"

namespace cv {

typedef float v16f __attribute__((ext_vector_type(16)));

// lane <-> lane + 32 exchange on the VALU (v_permlane32_swap: the upper half of the first operand and the lower half of the second trade places).  Called with the same
// value twice it returns {lower half's value, upper half's value} in BOTH halves - any symmetric combination of the two is the full reduction, identical in all lanes.
typedef unsigned cv_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xhalf_pair(float v, float& lo, float& hi) {
    const cv_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ float xhalf_max(float v) { float a, b; xhalf_pair(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xhalf_sum(float v) { float a, b; xhalf_pair(v, a, b); return a + b; }

// LDS-DMA with a scalar base: global address = sbase (an SGPR pair, wave-uniform) + voff (32-bit byte offset per lane) - the per-lane part of a tile's source
// addresses is the same for every tile, so a stream costs two scalar adds per tile instead of 64-bit vector arithmetic per piece.  (cv_glds16's notes apply.)
#ifndef CV_GLDS16S
__device__ __forceinline__ void cv_glds16s(const void* sbase, unsigned voff, const void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
#define CV_GLDS16S(sbase, voff, lds_wave_base) cv_glds16s((sbase), (voff), (lds_wave_base))
#endif

template <int V> struct attn_ic { static constexpr int value = V; };

// NW waves = NW * 32 queries of one (request, head) per workgroup; 3-stage K / V^T ring (the DMA of tile t + 2 overwrites the stage of tile t - 1), the tile loop
// unrolled by the ring so that every LDS address is a lane register + an immediate.
// WPE = waves per SIMD the register allocation must leave room for (4: 128 registers, 3: 168).
// ABL (tools/ubench/attn_probe only; 0 in the library): parts removed one at a time to price them - 1 no softmax arithmetic, 2 no DMA inside the loop, 4 no barrier /
// vmcnt, 8 no LDS fragment reads, 16 no MFMA.  Results are wrong by construction for ABL != 0.
template <int NW, int WPE = 3, int ABL = 0>
__global__ __launch_bounds__(NW * 64) CV_WAVES_PER_EU(WPE, WPE) void attn_flow32_kernel(AttnFlowArgs p) {
    constexpr int NST = 3, BQ = NW * 32, BKV = 64;
    constexpr int PW = 8 / NW;                             // 1 KB DMA pieces per wave, operand and tile (a tile = 64 rows x 128 bytes = 8 pieces per operand)
    static_assert(NW == 2 || NW == 4 || NW == 8, "attn_flow32_kernel: 8 DMA pieces per operand tile are dealt over the waves");
    constexpr float THR = 8.f;                             // lazy rescale threshold, log2 units
    // stage s: K tile at slots [1024 s, 1024 s + 512) as [key row][8 chunks of 16 bytes] with the chunk index XOR-swizzled, V^T tile at [1024 s + 512, 1024 s + 1024) as
    // [d row][8 chunks of 8 (permuted) keys]
    __shared__ __attribute__((aligned(16))) uint4 lds[NST * 1024];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 31, hi = lane >> 5;
    const int nqb = (p.T + BQ - 1) / BQ;
    const int bl = xcd_remap((int)blockIdx.x, (int)gridDim.x);        // the query tiles of one (request, head) share an XCD: its K / V^T stream through ONE L2
    const int qb = bl % nqb, h = (bl / nqb) % p.H, b = bl / (nqb * p.H);
    const float NEG_INF = -__builtin_huge_valf();
    const float c2 = p.scale * 1.4426950408889634f;                   // scores in log2 units: softmax on the bare v_exp_f32
    const float thr_raw = THR / c2;
    long long* dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 8 : nullptr;       // dev tool: wall_clock64() (100 MHz) of thread 0 at the phase boundaries; null in production
    int dn = 0;
    auto stamp = [&]() { if (dbg && tid == 0) dbg[dn++] = wall_clock64(); };
    stamp();                                                          // 0: start

    const bf16_t* kb = p.k + (long long)b * p.T * p.ld + h * 64;
    const bf16_t* vb = p.vt + (long long)b * p.vt_batch + (long long)h * 64 * p.ldt;
    const int Tkb = p.klen ? min(p.T, p.klen[b]) : p.T;               // keys of THIS batch row
    auto kend_of = [&](int q0, int nq) {                              // end of the key loop for queries [q0, q0 + nq)
        if (q0 >= p.T) return 0;
        return p.mask_mode == MASK_CHUNK ? min(Tkb, (min(p.T - 1, q0 + nq - 1) / p.chunk + 1) * p.chunk) : Tkb;
    };
    const int kend = kend_of(qb * BQ, BQ);                            // workgroup (DMA, barriers)

    // ---- DMA: piece i of an operand tile = rows 8 i .. 8 i + 7; lane l lands at LDS chunk l % 8 of row 8 i + l / 8 and FETCHES global chunk (l % 8) ^ swz(row).
    // Source = scalar tile base + the lane's byte offset inside a tile (the same for every tile; K rows >= T - only in the last tile - are clamped to row T - 1 and masked).
    // The first two tiles are requested before anything else is worked out: their flight (the prologue's longest wait) covers the rest of the set-up.
    const int prow = lane >> 3, pch = lane & 7;
    unsigned koff[PW], voff[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int row = 8 * (wave + NW * j) + prow, ch = pch ^ ((row >> 1) & 7);
        koff[j] = (unsigned)(row * p.ld * 2 + ch * 16); voff[j] = (unsigned)(row * p.ldt * 2 + ch * 16);       // ldt is a multiple of 64: a tile's 64 key columns exist (finite pad)
    }
    auto issue_tile = [&](int kt0, int st) {
        const bf16_t* ks = kb + (long long)kt0 * p.ld; const bf16_t* vs = vb + kt0;
        const bool last = kt0 + BKV > p.T;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int i = wave + NW * j;
            unsigned ko = koff[j];
            if (last) { const int row = 8 * i + prow; ko = (unsigned)((min(kt0 + row, p.T - 1) - kt0) * p.ld * 2 + (pch ^ ((row >> 1) & 7)) * 16); }
            CV_GLDS16S(ks, ko, &lds[st * 1024 + i * 64]);
            CV_GLDS16S(vs, voff[j], &lds[st * 1024 + 512 + i * 64]);
        }
    };
    const int ntile = (kend + BKV - 1) / BKV;
    if (ntile > 0) issue_tile(0, 0);
    if (ntile > 1) issue_tile(BKV, 1);

    const int qw0 = qb * BQ + wave * 32;
    const int kend_w = kend_of(qw0, 32);                              // this wave (its MFMAs): wave-uniform
    const int qi = qw0 + lq;
    const bool qvalid = qi < p.T;
    int klim = kend_w;                                                // rows past T compute finite garbage on zero queries, never stored
    if (qvalid && p.mask_mode == MASK_CHUNK) klim = min(Tkb, (qi / p.chunk + 1) * p.chunk);
    // smallest key limit of the wave = its first query's (limits grow with the query index; rows past T carry kend_w, the largest): tiles that end at or below it need
    // no per-element mask.  Wave-uniform by construction - no cross-lane reduction.
    const int klim_min_w = p.mask_mode == MASK_CHUNK ? min(Tkb, (qw0 / p.chunk + 1) * p.chunk) : kend_w;

    // Q^T as the B operand of S^T = K.Q^T: lane (q = lq, hi) supplies d = 16 dk + 8 hi .. + 7
    uint4 qf[4];
    {
        const bf16_t* qp = p.q + ((long long)b * p.T + (qvalid ? qi : 0)) * p.ld + h * 64;
#pragma unroll
        for (int dk = 0; dk < 4; ++dk) {
            uint4 t = *reinterpret_cast<const uint4*>(qp + dk * 16 + hi * 8);
            if (!qvalid) t = make_uint4(0u, 0u, 0u, 0u);
            qf[dk] = t;
        }
    }
    // fragment slots inside a stage: the K row of S^T row m is key perm(m) = m with bits 3 and 4 swapped; + 256 for the second 32-key block / d tile
    const int pm = (lq & 7) | ((lq & 8) << 1) | ((lq & 16) >> 1);
    const int kswz = (pm >> 1) & 7, vswz = (lq >> 1) & 7;
    int kslot[4], vslot[4];
#pragma unroll
    for (int dk = 0; dk < 4; ++dk) kslot[dk] = pm * 8 + ((dk * 2 + hi) ^ kswz);
#pragma unroll
    for (int c = 0; c < 4; ++c) vslot[c] = 512 + lq * 8 + ((2 * c + hi) ^ vswz);              // c = 2 bb + j: V^T chunk 4 bb + 2 j + hi

    float m_run = NEG_INF, m_s = NEG_INF, l_run = 0.f;
    v16f o0 = (v16f)(0.f), o1 = (v16f)(0.f);                                                    // O^T: d rows 0-31 / 32-63 x this lane's query
    stamp();                                                          // 1: set up, two tiles requested

    auto tile = [&](auto stc, int t) {
        constexpr int ST = decltype(stc)::value;
        const int kt0 = t * BKV;
        // own pieces of tile t have landed (tile t + 1's may still fly), then everyone's: the barrier also says that every wave has left tile t - 1's stage
        if constexpr (!(ABL & 4)) {
            if (t + 1 < ntile) CV_VMCNT(2 * PW); else CV_VMCNT0();
            __syncthreads();
        }
        if (t == 0) stamp();                                          // 2: the first tile has landed
        if constexpr (!(ABL & 2)) if (t + 2 < ntile) issue_tile(kt0 + 2 * BKV, (ST + 2) % NST);
        if (kt0 >= kend_w) return;                                                             // wave-uniform (chunk mask: a later wave of the workgroup needs this tile)
        const uint4* S = lds + ST * 1024;
        // ---- S^T = K.Q^T, two 32-key blocks
        v16f s0 = (v16f)(0.f), s1 = (v16f)(0.f);
#pragma unroll
        for (int dk = 0; dk < 4; ++dk) {
            uint4 k0, k1;
            if constexpr (ABL & 8) { k0 = qf[(dk + 1) & 3]; k1 = qf[(dk + 2) & 3]; } else { k0 = S[kslot[dk]]; k1 = S[256 + kslot[dk]]; }
            if constexpr (ABL & 16) { s0[dk] += __uint_as_float(k0.x); s1[dk] += __uint_as_float(k1.y); }
            else {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, k0), __builtin_bit_cast(v8bf, qf[dk]), s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, k1), __builtin_bit_cast(v8bf, qf[dk]), s1, 0, 0, 0);
            }
        }
        // register r of block bb <-> key kt0 + 32 bb + (r & 3) + 4 hi + 16 ((r >> 2) & 1) + 8 (r >> 3)
        if (kt0 + BKV > klim_min_w) {
            const int lim = klim - kt0 - 4 * hi;                      // compared with the register's constant part
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int koff_r = (r & 3) + 16 * ((r >> 2) & 1) + 8 * (r >> 3);
                s0[r] = koff_r < lim ? s0[r] : NEG_INF;
                s1[r] = koff_r + 32 < lim ? s1[r] : NEG_INF;
            }
        }
        uint4 pb[4];                                                  // pb[2 bb + j] = block bb, registers 8 j .. 8 j + 7: P rounded to bf16, the B operand of the second product
        if constexpr (ABL & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const v16f& s = j < 2 ? s0 : s1; const int r = 8 * (j & 1); pb[j] = make_uint4(pack_bf16x2(s[r], s[r + 1]), pack_bf16x2(s[r + 2], s[r + 3]), pack_bf16x2(s[r + 4], s[r + 5]), pack_bf16x2(s[r + 6], s[r + 7])); }
        } else {
            // ---- row maximum of the tile (raw score units; c2 > 0)
            float mt;
            {
                float a = fmaxf(fmaxf(s0[0], s0[1]), s0[2]), bq = fmaxf(fmaxf(s0[3], s0[4]), s0[5]), c = fmaxf(fmaxf(s0[6], s0[7]), s0[8]), d = fmaxf(fmaxf(s0[9], s0[10]), s0[11]);
                a = fmaxf(fmaxf(a, s0[12]), s0[13]); bq = fmaxf(fmaxf(bq, s0[14]), s0[15]);
                c = fmaxf(fmaxf(c, s1[0]), s1[1]); d = fmaxf(fmaxf(d, s1[2]), s1[3]);
                a = fmaxf(fmaxf(a, s1[4]), s1[5]); bq = fmaxf(fmaxf(bq, s1[6]), s1[7]); c = fmaxf(fmaxf(c, s1[8]), s1[9]); d = fmaxf(fmaxf(d, s1[10]), s1[11]);
                a = fmaxf(fmaxf(a, s1[12]), s1[13]); bq = fmaxf(fmaxf(bq, s1[14]), s1[15]);
                mt = xhalf_max(fmaxf(fmaxf(a, bq), fmaxf(c, d)));
            }
            // ---- lazy rescale: this lane moves its maximum only past the threshold; the multiply runs when some lane of the wave moved (by 1.0 in the others: exact)
            const bool move = mt > m_run + thr_raw;                   // m_run = -inf (first tile): true for any finite mt
            if (__any(move)) {
                const float m_new = move ? mt : m_run;
                const float alpha = move ? __builtin_amdgcn_exp2f((m_run - m_new) * c2) : 1.f;     // first tile: exp2(-inf) = 0 on zero accumulators
                m_run = m_new; m_s = m_new * c2;
                l_run *= alpha;
                o0 = o0 * alpha; o1 = o1 * alpha;
            }
            // ---- P = exp2(s c2 - m c2); four partial denominators (fixed order); rounded to bf16 as the B operand of the second product
            float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c2, -m_s)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r + 1], c2, -m_s));
                const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r + 2], c2, -m_s)), e3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r + 3], c2, -m_s));
                const float f0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c2, -m_s)), f1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r + 1], c2, -m_s));
                const float f2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r + 2], c2, -m_s)), f3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r + 3], c2, -m_s));
                l0 += e0; l1 += e1; l2 += e2; l3 += e3; l0 += f0; l1 += f1; l2 += f2; l3 += f3;
                const unsigned a0 = pack_bf16x2(e0, e1), a1 = pack_bf16x2(e2, e3), b0 = pack_bf16x2(f0, f1), b1 = pack_bf16x2(f2, f3);
                if (r & 4) { pb[r >> 3].z = a0; pb[r >> 3].w = a1; pb[2 + (r >> 3)].z = b0; pb[2 + (r >> 3)].w = b1; }
                else { pb[r >> 3].x = a0; pb[r >> 3].y = a1; pb[2 + (r >> 3)].x = b0; pb[2 + (r >> 3)].y = b1; }
            }
            l_run += (l0 + l1) + (l2 + l3);
        }
        // ---- O^T += V^T.P^T: k-slot (hi, i) of product c = 2 bb + j is V^T chunk 4 bb + 2 j + hi, element i = the key of register 8 j + i of block bb
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint4 v0, v1;
            if constexpr (ABL & 8) { v0 = qf[c]; v1 = qf[c ^ 2]; } else { v0 = S[vslot[c]]; v1 = S[256 + vslot[c]]; }
            if constexpr (ABL & 16) { o0[c] += __uint_as_float(v0.x ^ pb[c].x); o1[c] += __uint_as_float(v1.y ^ pb[c].y); }
            else {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, v0), __builtin_bit_cast(v8bf, pb[c]), o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, v1), __builtin_bit_cast(v8bf, pb[c]), o1, 0, 0, 0);
            }
        }
    };
    for (int t = 0; t < ntile; t += 3) {
        tile(attn_ic<0>{}, t);
        if (t + 1 < ntile) tile(attn_ic<1>{}, t + 1);
        if (t + 2 < ntile) tile(attn_ic<2>{}, t + 2);
    }
    stamp();                                                          // 3: key loop done
    // ---- O / l -> bf16.  Register r of d tile dt is d = 32 dt + (r & 3) + 8 (r >> 2) + 4 hi: four consecutive d per register quad.
    const float l = xhalf_sum(l_run);                                 // (lanes l and l + 32 hold the same query: both or neither are valid)
    if (qvalid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        bf16_t* op = p.o + ((long long)b * p.T + qi) * p.ldo + h * 64 + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<uint2*>(op + 8 * g) = make_uint2(pack_bf16x2(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16x2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
            *reinterpret_cast<uint2*>(op + 32 + 8 * g) = make_uint2(pack_bf16x2(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16x2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
        }
    }
    stamp();                                                          // 4: stored
    if (dbg && tid == 0) { dbg[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); dbg[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); dbg[7] = qb; }       // HW_ID, XCC_ID (where the workgroup ran), its query block
}


}  // namespace cv
