# Dev tool (round 6, CV_BUILD_EXPERIMENTS build): HiFT at 500 frames with 2 / 3 / 4 register stages in the two-sided-split convolutions (CV_GEMM_WX3_STAGES) x the tile rule
for st in 2 3 4 2; do for mb in 720 480; do echo "stages $st min_blocks $mb: $(CV_BUILD_EXPERIMENTS=1 CV_GEMM_WX3_STAGES=$st CV_GEMM_MIN_BLOCKS_F32=$mb python tools/probe_hift_busy.py 500 40 2>/dev/null | tail -1)"; done; done
