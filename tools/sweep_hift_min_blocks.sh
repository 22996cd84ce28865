# Dev tool (round 6): HiFT at 500 frames against the fp32-path tile rule's minimum workgroup count (CV_GEMM_MIN_BLOCKS_F32, read at every launch; gemm_conv.hip)
for mb in ${SWEEP:-720 940 1000 1260 1900 2600 4000 480}; do echo "min_blocks $mb: $(CV_GEMM_MIN_BLOCKS_F32=$mb python tools/probe_hift_busy.py 500 40 2>/dev/null | tail -1)"; done
