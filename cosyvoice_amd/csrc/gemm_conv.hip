// Host dispatch for the implicit-GEMM kernel: picks the tile shape so that a launch fills the
// 256 CUs of an MI355X where the problem allows it (most of this path's GEMMs are small).
#include "gemm_conv.h"
#include <cstdlib>
#include "api_common.h"

namespace cv {

// (A twice-as-deep k-tile for the bf16 variants - their LDS tiles are half the size - was measured on MI355X: +6 ms per utterance,
// the extra registers cost more than the saved barriers.)
template <int BM, int BN, int BK, int ST = 2>
static void launch_cfg(const GemmConvArgs& a, bool w_bf16, int batch, hipStream_t stream) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, batch), block(256);
    if (a.a_bf16 && w_bf16 && a.a_vec) {        // bf16 x bf16 MFMA (fp32 accumulate); other layouts keep the fp32-accurate paths
        hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK, true, true, ST, true>), grid, block, 0, stream, a);
        return;
    }
    // fp32 activations x fp32 weights whose three bf16 planes were prepared at load time (HiFT): both operands split, six exact products per k on the bf16
    // matrix pipe (gemm_conv.h, WX3).  BK <= 64: the weight tile has 3 BN rows.  CV_GEMM_WX3=0 pins the fp32 MFMA chain (A/B knob, read at every launch).
    if (a.a_vec && !w_bf16 && a.W3) {
        const char* e = getenv("CV_GEMM_WX3");
        if (!(e && e[0] == '0')) {
            constexpr int BK3 = BK > 64 ? 64 : BK;
#ifdef CV_BUILD_EXPERIMENTS
            // dev knob (experiments builds): half-depth k tiles (less LDS and fewer staging registers per workgroup: more of them per CU)
            if constexpr (BM <= 64 && BK3 == 64) if (const char* bk = getenv("CV_GEMM_WX3_BK"); bk && bk[0] == '3') {
                hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, 32, true, true, ST, false, 2, 2, true, true>), grid, block, 0, stream, a); return;
            }
            // dev knob (experiments builds): a deeper register ring for the two-sided split (3 or 4 tiles in flight per wave)
            if constexpr (BM <= 64) if (const char* st = getenv("CV_GEMM_WX3_STAGES")) {
                if (st[0] == '3') { hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK3, true, true, 3, false, 2, 2, true, true>), grid, block, 0, stream, a); return; }
                if (st[0] == '4') { hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK3, true, true, 4, false, 2, 2, true, true>), grid, block, 0, stream, a); return; }
            }
#endif
            hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK3, true, true, ST, false, 2, 2, true, true>), grid, block, 0, stream, a);
            return;
        }
    }
    // fp32 activations x bf16 weights: the exact three-term split on the bf16 matrix pipe (gemm_conv.h, AX3).  CV_GEMM_X3=0 pins the fp32 MFMA chain
    // (A/B knob, read at every launch); results agree to fp32 rounding either way.
    if (a.a_vec && w_bf16) {
        const char* e = getenv("CV_GEMM_X3");
        if (!(e && e[0] == '0')) { hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK, true, true, ST, false, 2, 2, true>), grid, block, 0, stream, a); return; }
    }
    if (a.a_vec) {
        if (w_bf16) hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK, true, true, ST>), grid, block, 0, stream, a);
        else        hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK, false, true, ST>), grid, block, 0, stream, a);
    } else {
        if (w_bf16) hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK, true, false, ST>), grid, block, 0, stream, a);
        else        hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, BK, false, false, ST>), grid, block, 0, stream, a);
    }
}

// The 16 x 32 two-wave tile (680 workgroups for the flow's N = 256 GEMMs instead of 344) measured neutral-to-worse on MI355X
// (222.6 ms per utterance without, 225.4 with, interleaved runs on one box): it stays out of the size-based choice unless
// CV_GEMM_TWO_WAVE_TILE=1, and stays covered through CV_GEMM_FORCE_TILE=5 (tests/test_ops.py::test_every_gemm_tile_shape).
static bool use_two_wave_tile() {
    static const bool on = [] { const char* e = getenv("CV_GEMM_TWO_WAVE_TILE"); return e && e[0] == '1'; }();
    return on;
}

// test hook: CV_GEMM_FORCE_TILE=0..5 pins the tile shape (read at every launch), so that tests/test_ops.py can hold EVERY instantiation
// to the oracle on small problems - the size-based choice below would only ever give them the smallest tile.
static int forced_tile() {
    const char* e = getenv("CV_GEMM_FORCE_TILE");
    return (e && e[0] >= '0' && e[0] <= '5' && e[1] == 0) ? e[0] - '0' : -1;
}

void launch_gemm_conv(const GemmConvArgs& a, bool w_bf16, int batch, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0 || batch <= 0) return;
    struct Cfg { int bm, bn; };
    static const Cfg cfgs[] = {{128, 128}, {128, 64}, {64, 64}, {32, 64}, {32, 32}, {16, 32}};
    // bf16 tiles are small (LDS and registers) and their launches are latency-bound: ask for ~3 co-resident workgroups per CU (A/B on MI355X: 240 -> 480 -> 720 -> 1024 minimum workgroups gave 238 -> 224 -> 221 -> 226 ms per utterance), so one
    // workgroup's load wait overlaps another's LDS / MFMA phases; the fp32 tiles ask for ~2 per CU.
    const bool bf16_path = a.a_bf16 && w_bf16 && a.a_vec;
    long long min_blocks = bf16_path ? 720 : 480;     // fp32 tiles: ~2 per CU (HiFT, 500 frames: 10.05 ms at 240, 8.66 at 480, 9.03 at 720, 9.23 at 1500: profiles/r2_batch_decode_ab.txt)
    // A tile taller than 32 rows may not pad M by more than 25 %: the LLM prefill of U10 has M = 131 = 128 + 3, and the 128-row tile ran half of its
    // workgroups on 3 useful rows (prefill 5.14 -> 4.38 ms with this rule; asking for more workgroups on top gave 4.2: profiles/r2_batch_decode_ab.txt).
    // CV_GEMM_MAX_WASTE / CV_GEMM_MIN_BLOCKS_F32: dev knobs for such sweeps, read at every launch.
    // The two-sided split (WX3: HiFT's fp32 weights) stages 3 BN weight rows per k-step and splits its activations on the VALU: its launches are bound by that staging, not by the
    // matrix pipe, and want ~3 workgroups per CU like the bf16 tiles (round 6, HiFT at 500 frames: 4.47 ms at 480, 4.14 at 720 - 1000, 4.23 - 4.28 from 1260, 4.95 - 6.6 below 480:
    // profiles/r6_hift.txt section 3; the tile shape does not change a result bit - every output element sums the same products in the same order).
    if (!bf16_path && a.a_vec && !w_bf16 && a.W3) min_blocks = 720;
    double max_waste = 1.25;
    if (const char* e = getenv("CV_GEMM_MIN_BLOCKS_F32"); e && !bf16_path) min_blocks = atoll(e);
    if (const char* e = getenv("CV_GEMM_MAX_WASTE")) max_waste = atof(e);
    const int ncfg = (bf16_path && a.Kp >= 128 && use_two_wave_tile()) ? 6 : 5;   // the 16 x 32 two-wave tile exists for the bf16 path only
    int pick = ncfg - 1;                             // nothing reaches the target: the smallest tile = the most workgroups
    for (int c = 0; c < ncfg; ++c) {
        const long long blocks = (long long)((a.M + cfgs[c].bm - 1) / cfgs[c].bm) * ((a.N + cfgs[c].bn - 1) / cfgs[c].bn) * batch;
        if (cfgs[c].bn > 32 && a.N <= cfgs[c].bn / 2) continue;       // don't pad N by 2x or more
        if (cfgs[c].bm > 32 && a.M <= cfgs[c].bm / 2) continue;
        if (cfgs[c].bm > 32 && (double)((a.M + cfgs[c].bm - 1) / cfgs[c].bm * cfgs[c].bm) > max_waste * a.M) continue;
        if (blocks >= min_blocks) { pick = c; break; }
    }
    if (const int f = forced_tile(); f >= 0) { if (f == 5 && !(bf16_path && a.Kp >= 128)) throw Error("CV_GEMM_FORCE_TILE=5 needs the bf16 path and Kp >= 128"); pick = f; }
    // Small tiles run ~1 workgroup per CU and are bound by memory latency per k-iteration (activations written by another XCD,
    // weights from the Infinity Cache: ~3 us), so they take the biggest BK that fits 64 KB of LDS: fewer, fatter iterations.
    const bool bigk = a.Kp >= 128;
    switch (pick) {
        case 0: launch_cfg<128, 128, 32>(a, w_bf16, batch, stream); break;
        case 1: if (a.Kp >= 64) launch_cfg<128, 64, 64>(a, w_bf16, batch, stream); else launch_cfg<128, 64, 32>(a, w_bf16, batch, stream); break;
        case 2: if (bigk) launch_cfg<64, 64, 128>(a, w_bf16, batch, stream); else launch_cfg<64, 64, 64>(a, w_bf16, batch, stream); break;
        case 3: if (bigk) launch_cfg<32, 64, 128>(a, w_bf16, batch, stream); else launch_cfg<32, 64, 64>(a, w_bf16, batch, stream); break;
        case 4: if (bigk) launch_cfg<32, 32, 128>(a, w_bf16, batch, stream); else launch_cfg<32, 32, 64>(a, w_bf16, batch, stream); break;
        default: {                                   // 16 x 32, wave grid 1 x 2 (128 threads), bf16 MFMA only
            dim3 grid((a.M + 15) / 16, (a.N + 31) / 32, batch);
            hipLaunchKernelGGL((gemm_conv_kernel<16, 32, 128, true, true, 2, true, 1, 2>), grid, dim3(128), 0, stream, a);
            break;
        }
    }
}

}  // namespace cv
