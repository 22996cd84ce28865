# Dev tool (round 6, CV_BUILD_EXPERIMENTS build): HiFT at 500 frames with half-depth k tiles in the two-sided-split convolutions (CV_GEMM_WX3_BK=32) x the tile rule
for bk in 64 32 64 32; do for mb in 720 480 1000; do echo "BK $bk min_blocks $mb: $(CV_BUILD_EXPERIMENTS=1 CV_GEMM_WX3_BK=$bk CV_GEMM_MIN_BLOCKS_F32=$mb python tools/probe_hift_busy.py 500 40 2>/dev/null | tail -1)"; done; done
