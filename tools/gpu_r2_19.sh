for mb in 240 480 720 1500 3000; do
echo "== CV_GEMM_MIN_BLOCKS_F32=$mb"
CV_GEMM_MIN_BLOCKS_F32=$mb timeout 200 python tools/probe_all.py hift 2>&1 | grep "hift rep" | tail -2
done | tee gpurun_out/r2_probe_hift_minblocks.txt
