// Flash-style attention for head_dim 64 in exact fp32 (v_mfma_f32_16x16x4_f32), gfx950.
//
// One workgroup = 4 waves = 64 query rows of one (batch, head); K/V stream through LDS in 64-key tiles.
// Per wave (16 queries):  S^T = K·Q^T  -> lane l holds scores of query (l&15) for keys (l>>4)*4+r of each
// 16-key sub-tile, which is *already* the B-operand layout the P·V MFMA wants (k-slot = l>>4, step r),
// so P never leaves registers and softmax row-reductions are two xor-shuffles (16, 32).
// O^T = V^T·P^T accumulates [d][query] -> each lane ends with 4 consecutive d of its query: one 16B store.
// Masks (full / causal / block-causal "chunk") are computed from indices in-kernel: no [B,T,T] bias tensor
// (the reference materialises one per block group, flow/decoder.py:439-443).
#pragma once
#include "common.h"

namespace cv {

struct AttnArgs {
    const float* q; long long q_batch; int q_row; int q_head;
    const float* k; long long k_batch; int k_row; int k_head;
    const float* v; long long v_batch; int v_row; int v_head;
    float* o; long long o_batch; int o_row; int o_head;
    int B, H, kv_group, Tq, Tk;
    float scale; int mask_mode; int chunk;
    const float* rel_bd; long long bd_batch; long long bd_head; int bd_row;
    const int* klen;   // optional [B]: batch row b attends keys < min(Tk, klen[b]) - the padded rows of a batch of unequal lengths are never keys.
                       // The key loop of a row then ends where it would end for that row alone (same tiles, same order: bit-identical results).
    int bf16;      // 1 (2, 3: forced workgroup shape): q, k, v and the probabilities are rounded to bf16 and both products run on v_mfma_f32_16x16x32_bf16 (fp32 softmax / accumulate)
};

enum { MASK_NONE = 0, MASK_CAUSAL = 1, MASK_CHUNK = 2 };

static __global__ __launch_bounds__(256) void attention_kernel(AttnArgs p) {
    constexpr int BQ = 64, BKV = 64, LD = 68;
    __shared__ __attribute__((aligned(16))) float Ks[BKV * LD];
    __shared__ __attribute__((aligned(16))) float Vs[BKV * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, lg = lane >> 4;
    // XCD-aware order: all query tiles of one (batch, head) get consecutive remapped ids, i.e. the same XCD, so that head's K/V
    // stream through ONE L2 instead of all eight.
    const int nqb = gridDim.x, nbl = gridDim.x * gridDim.y * gridDim.z;
    const int bl = xcd_remap((int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x), nbl);
    const int qb = bl % nqb, h = (bl / nqb) % (int)gridDim.y, b = bl / (nqb * (int)gridDim.y), hk = h / p.kv_group;
    const int qi = qb * BQ + wave * 16 + lq;
    const bool qvalid = qi < p.Tq;
    const float NEG_INF = -__builtin_huge_valf();

    // Q fragments stay in registers: qf[dg] = Q[qi][dg*16 + lg*4 .. +3]
    float4 qf[4];
    {
        const float* qp = p.q + (long long)b * p.q_batch + (long long)(qvalid ? qi : 0) * p.q_row + (long long)h * p.q_head;
#pragma unroll
        for (int dg = 0; dg < 4; ++dg) {
            float4 t = *reinterpret_cast<const float4*>(qp + dg * 16 + lg * 4);
            if (!qvalid) t = make_float4(0.f, 0.f, 0.f, 0.f);
            qf[dg] = t;
        }
    }
    const int qmax_blk = min(p.Tq - 1, qb * BQ + BQ - 1);
    const int Tkb = p.klen ? min(p.Tk, p.klen[b]) : p.Tk;      // keys of THIS batch row (padded batches: rows of different lengths share a launch)
    int kend = Tkb;
    if (p.mask_mode == MASK_CAUSAL) kend = min(Tkb, qmax_blk + (p.Tk - p.Tq) + 1);
    else if (p.mask_mode == MASK_CHUNK) kend = min(Tkb, (qmax_blk / p.chunk + 1) * p.chunk);
    int klim = Tkb;    // per-query key limit (exclusive)
    if (p.mask_mode == MASK_CAUSAL) klim = min(Tkb, qi + (p.Tk - p.Tq) + 1);
    else if (p.mask_mode == MASK_CHUNK) klim = min(Tkb, (qi / p.chunk + 1) * p.chunk);
    if (!qvalid) klim = 0;

    const float* kb = p.k + (long long)b * p.k_batch + (long long)hk * p.k_head;
    const float* vb = p.v + (long long)b * p.v_batch + (long long)hk * p.v_head;
    const float* bd = p.rel_bd ? p.rel_bd + (long long)b * p.bd_batch + (long long)h * p.bd_head + (long long)(qvalid ? qi : 0) * p.bd_row + (p.Tq - 1 - (qvalid ? qi : 0)) : nullptr;

    float m_run = NEG_INF, l_run = 0.f;
    v4f acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[d] = (v4f){0.f, 0.f, 0.f, 0.f};

    // K/V tiles go global -> registers -> LDS; the NEXT tile's loads are issued right after the current tile is parked in LDS,
    // so they are in flight under the 128 MFMAs + softmax of the current tile (one workgroup per CU: nothing else hides them).
    float4 rk[4], rv[4];
    auto load_kv = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vv = tid + i * 256, row = vv >> 4, c4 = (vv & 15) * 4;
            const int key = kt0 + row;
            const bool ok = key < p.Tk;                       // unconditional loads (clamped row) keep the vmcnt bookkeeping exact
            const long long kr = ok ? key : 0;
            float4 kx = *reinterpret_cast<const float4*>(kb + kr * p.k_row + c4);
            float4 vx = *reinterpret_cast<const float4*>(vb + kr * p.v_row + c4);
            if (!ok) { kx = make_float4(0.f, 0.f, 0.f, 0.f); vx = kx; }
            rk[i] = kx; rv[i] = vx;
        }
    };
    if (kend > 0) load_kv(0);
    for (int kt0 = 0; kt0 < kend; kt0 += BKV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vv = tid + i * 256, row = vv >> 4, c4 = (vv & 15) * 4;
            *reinterpret_cast<float4*>(&Ks[row * LD + c4]) = rk[i];
            *reinterpret_cast<float4*>(&Vs[row * LD + c4]) = rv[i];
        }
        __syncthreads();
        if (kt0 + BKV < kend) load_kv(kt0 + BKV);

        // scores: s[kt][r] <-> key kt0 + kt*16 + lg*4 + r, query lq
        v4f s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            v4f sa = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dg = 0; dg < 4; ++dg) {
                const float4 kf = *reinterpret_cast<const float4*>(&Ks[(kt * 16 + lq) * LD + dg * 16 + lg * 4]);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[dg].x, sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[dg].y, sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[dg].z, sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[dg].w, sa, 0, 0, 0);
            }
            s[kt] = sa;
        }
        float mt = NEG_INF;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt0 + kt * 16 + lg * 4 + r;
                float x = s[kt][r];
                if (bd && key < klim) x += bd[key];
                x = key < klim ? x * p.scale : NEG_INF;
                s[kt][r] = x;
                mt = fmaxf(mt, x);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = (m_run == NEG_INF) ? 0.f : expf(m_run - m_new);
        float rsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = s[kt][r];
                const float e = (x == NEG_INF) ? 0.f : expf(x - m_new);
                s[kt][r] = e;
                rsum += e;
            }
        rsum += __shfl_xor(rsum, 16);
        rsum += __shfl_xor(rsum, 32);
        l_run = l_run * alpha + rsum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] = acc[d] * alpha;

        // O^T += V^T P^T : a = V[key][d = dt*16 + lq], b = P[query lq][key slot]
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = s[kt][r];
                const float* vrow = &Vs[(kt * 16 + lg * 4 + r) * LD + lq];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[dt * 16], pv, acc[dt], 0, 0, 0);
            }
        __syncthreads();
    }

    if (qvalid) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        float* op = p.o + (long long)b * p.o_batch + (long long)qi * p.o_row + (long long)h * p.o_head;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<float4*>(op + dt * 16 + lg * 4) = make_float4(acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// bf16-operand variant (the "bf16" precision mode of the flow): same tiling, masks and online softmax, but
//   * K is staged as bf16 [key][d], V as bf16 TRANSPOSED [d][key] (the P.V MFMA wants k = keys contiguous), Q is packed once;
//   * S^T = K.Q^T takes 8 v_mfma_f32_16x16x32_bf16 per 64-key tile and wave instead of 64 fp32 MFMAs, O^T += V^T.P^T another 8
//     (the fp32 kernel is MFMA-bound: 128 x 32 cycles per tile);
//   * the k-slot trick again: the score MFMA leaves lane (q, g) with keys {16s + 4g + r}, s = 0..1, r = 0..3 of each 32-key block -
//     declaring exactly that order to be the k order of the P.V MFMA makes P a register-only repack, and V^T is stored with
//     its key columns permuted the same way (col = 32 blk + 8 g + 4 s + r) so the A operand is one 16-byte LDS read;
//   * two tiles are in flight behind the one being multiplied (2 register sets): with the MFMA time gone the kernel is bound
//     by the L2 -> CU latency of the K/V stream (one workgroup per CU, 32 KB per tile).
// Scores, running max / sum, the output accumulator and everything in HBM stay fp32.
// (register budget pinned to 2 waves per SIMD: left to itself the compiler aims for 3 and spills 144 B per lane into scratch)
// NW = waves per workgroup (16 queries each).  NW = 4 streams K/V once per 64 queries (default); NW = 2 doubles the workgroup count
// (352 for the estimator's T = 674, B = 2, H = 8) so that two or three of them share a CU - measured neutral, kept selectable.
template <int NW>
__global__ __launch_bounds__(NW * 64) CV_WAVES_PER_EU(1, 2) void attention_bf16_kernel(AttnArgs p) {
    constexpr int NT = NW * 64, BQ = NW * 16, BKV = 64, LDH = 36;          // LDS row pitch in dwords: 64 bf16 + 8 pad
    constexpr int NI = 512 / NT;                        // key pairs staged per thread and tile (32 pairs x 16 column groups / NT)
    __shared__ __attribute__((aligned(16))) unsigned Ks[BKV * LDH];
    __shared__ __attribute__((aligned(16))) unsigned Vt[64 * LDH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, lg = lane >> 4;
    const int nqb = gridDim.x, nbl = gridDim.x * gridDim.y * gridDim.z;
    const int bl = xcd_remap((int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x), nbl);
    const int qb = bl % nqb, h = (bl / nqb) % (int)gridDim.y, b = bl / (nqb * (int)gridDim.y), hk = h / p.kv_group;
    const int qi = qb * BQ + wave * 16 + lq;
    const bool qvalid = qi < p.Tq;
    const float NEG_INF = -__builtin_huge_valf();
    const float scale2 = p.scale * 1.4426950408889634f;

    // Q as the B operand of S^T = K.Q^T: lane (q = lq, g = lg) supplies d = 32 dg + 8 g .. + 7
    uint4 qf[2];
    {
        const float* qp = p.q + (long long)b * p.q_batch + (long long)(qvalid ? qi : 0) * p.q_row + (long long)h * p.q_head;
#pragma unroll
        for (int dg = 0; dg < 2; ++dg) {
            float4 a = *reinterpret_cast<const float4*>(qp + dg * 32 + lg * 8), c = *reinterpret_cast<const float4*>(qp + dg * 32 + lg * 8 + 4);
            if (!qvalid) { a = make_float4(0.f, 0.f, 0.f, 0.f); c = a; }
            qf[dg] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(c.x, c.y), pack_bf16x2(c.z, c.w));
        }
    }
    const int qmax_blk = min(p.Tq - 1, qb * BQ + BQ - 1);
    const int Tkb = p.klen ? min(p.Tk, p.klen[b]) : p.Tk;      // keys of THIS batch row (padded batches: rows of different lengths share a launch)
    int kend = Tkb;
    if (p.mask_mode == MASK_CAUSAL) kend = min(Tkb, qmax_blk + (p.Tk - p.Tq) + 1);
    else if (p.mask_mode == MASK_CHUNK) kend = min(Tkb, (qmax_blk / p.chunk + 1) * p.chunk);
    int klim = Tkb;    // per-query key limit (exclusive)
    if (p.mask_mode == MASK_CAUSAL) klim = min(Tkb, qi + (p.Tk - p.Tq) + 1);
    else if (p.mask_mode == MASK_CHUNK) klim = min(Tkb, (qi / p.chunk + 1) * p.chunk);
    if (!qvalid) klim = 0;

    const float* kb = p.k + (long long)b * p.k_batch + (long long)hk * p.k_head;
    const float* vb = p.v + (long long)b * p.v_batch + (long long)hk * p.v_head;
    const float* bd = p.rel_bd ? p.rel_bd + (long long)b * p.bd_batch + (long long)h * p.bd_head + (long long)(qvalid ? qi : 0) * p.bd_row + (p.Tq - 1 - (qvalid ? qi : 0)) : nullptr;

    float m_run = NEG_INF, l_run = 0.f;
    v4f acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[d] = (v4f){0.f, 0.f, 0.f, 0.f};

    // thread t stages the key PAIRS (2j, 2j+1), j = (t >> 4) + (NT/16) i, columns c4 .. c4+3: adjacent keys land in one dword of V^T
    const int c4 = (tid & 15) * 4, j0 = tid >> 4;
    // (native vectors, not the float4 struct: as float4 these staging arrays stayed ALLOCAS - 144 bytes per lane, which the backend parked in LDS (36 KB per workgroup next
    // to the 18 KB declared) or, with three waves per SIMD, in scratch: the "spill" the register budget above was pinned against.  Round 5, found through flow_band.h.)
    v4f rkA[NI][2], rvA[NI][2], rkB[NW == 4 ? NI : 1][2], rvB[NW == 4 ? NI : 1][2];
    auto load_kv = [&](int kt0, auto& rk, auto& rv) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int key = kt0 + 2 * (j0 + (NT / 16) * i) + par;
                const bool ok = key < p.Tk;                   // unconditional loads (clamped row) keep the vmcnt bookkeeping exact
                const long long kr = ok ? key : 0;
                v4f kx = *reinterpret_cast<const v4f*>(kb + kr * p.k_row + c4);
                v4f vx = *reinterpret_cast<const v4f*>(vb + kr * p.v_row + c4);
                if (!ok) { kx = (v4f){0.f, 0.f, 0.f, 0.f}; vx = kx; }
                rk[i][par] = kx; rv[i][par] = vx;
            }
    };
    auto store_kv = [&](const auto& rk, const auto& rv) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k0 = 2 * (j0 + (NT / 16) * i);          // even key of the pair, 0..62
#pragma unroll
            for (int par = 0; par < 2; ++par)
                *reinterpret_cast<uint2*>(&Ks[(k0 + par) * LDH + c4 / 2]) = make_uint2(pack_bf16x2(rk[i][par].x, rk[i][par].y), pack_bf16x2(rk[i][par].z, rk[i][par].w));
            // key k0 = 32 blk + 16 s + 4 g + r  ->  column 32 blk + 8 g + 4 s + r (r even: the pair shares a dword)
            const int col = (k0 & 32) + ((k0 >> 2) & 3) * 8 + ((k0 >> 4) & 1) * 4 + (k0 & 3);
            Vt[(c4 + 0) * LDH + col / 2] = pack_bf16x2(rv[i][0].x, rv[i][1].x);
            Vt[(c4 + 1) * LDH + col / 2] = pack_bf16x2(rv[i][0].y, rv[i][1].y);
            Vt[(c4 + 2) * LDH + col / 2] = pack_bf16x2(rv[i][0].z, rv[i][1].z);
            Vt[(c4 + 3) * LDH + col / 2] = pack_bf16x2(rv[i][0].w, rv[i][1].w);
        }
    };
    auto compute = [&](int kt0) {
        // scores: s[kt][r] <-> key kt0 + kt*16 + lg*4 + r, query lq
        v4f s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            v4f sa = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) {
                const uint4 kf = *reinterpret_cast<const uint4*>(&Ks[(kt * 16 + lq) * LDH + dg * 16 + lg * 4]);
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, kf), __builtin_bit_cast(v8bf, qf[dg]), sa, 0, 0, 0);
            }
            s[kt] = sa;
        }
        float mt = NEG_INF;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt0 + kt * 16 + lg * 4 + r;
                float x = s[kt][r];
                if (bd && key < klim) x += bd[key];
                x = key < klim ? x * scale2 : NEG_INF;            // scores in log2 units: softmax via v_exp_f32
                s[kt][r] = x;
                mt = fmaxf(mt, x);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = (m_run == NEG_INF) ? 0.f : exp2f(m_run - m_new);
        float rsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = s[kt][r];
                const float e = (x == NEG_INF) ? 0.f : exp2f(x - m_new);
                s[kt][r] = e;
                rsum += e;                                   // the denominator sums the UNROUNDED probabilities
            }
        rsum += __shfl_xor(rsum, 16);
        rsum += __shfl_xor(rsum, 32);
        l_run = l_run * alpha + rsum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] = acc[d] * alpha;
        // O^T += V^T P^T: k-slot (g, 4 s + r) of 32-key block blk is key 32 blk + 16 s + 4 g + r on both operands
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const uint4 pb = make_uint4(pack_bf16x2(s[2 * blk][0], s[2 * blk][1]), pack_bf16x2(s[2 * blk][2], s[2 * blk][3]),
                                        pack_bf16x2(s[2 * blk + 1][0], s[2 * blk + 1][1]), pack_bf16x2(s[2 * blk + 1][2], s[2 * blk + 1][3]));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint4 vf = *reinterpret_cast<const uint4*>(&Vt[(dt * 16 + lq) * LDH + blk * 16 + lg * 4]);
                acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, vf), __builtin_bit_cast(v8bf, pb), acc[dt], 0, 0, 0);
            }
        }
    };

    if constexpr (NW == 4) {                         // two tiles in flight behind the one being multiplied
        if (kend > 0) load_kv(0, rkA, rvA);
        if (kend > BKV) load_kv(BKV, rkB, rvB);
        for (int kt0 = 0; kt0 < kend; kt0 += 2 * BKV) {
            store_kv(rkA, rvA);
            __syncthreads();
            if (kt0 + 2 * BKV < kend) load_kv(kt0 + 2 * BKV, rkA, rvA);
            compute(kt0);
            __syncthreads();
            if (kt0 + BKV < kend) {
                store_kv(rkB, rvB);
                __syncthreads();
                if (kt0 + 3 * BKV < kend) load_kv(kt0 + 3 * BKV, rkB, rvB);
                compute(kt0 + BKV);
                __syncthreads();
            }
        }
    } else {                                         // small workgroups overlap through co-residency: one tile in flight, half the registers
        if (kend > 0) load_kv(0, rkA, rvA);
        for (int kt0 = 0; kt0 < kend; kt0 += BKV) {
            store_kv(rkA, rvA);
            __syncthreads();
            if (kt0 + BKV < kend) load_kv(kt0 + BKV, rkA, rvA);
            compute(kt0);
            __syncthreads();
        }
    }

    if (qvalid) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        float* op = p.o + (long long)b * p.o_batch + (long long)qi * p.o_row + (long long)h * p.o_head;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<float4*>(op + dt * 16 + lg * 4) = make_float4(acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv);
    }
}

}  // namespace cv
