mkdir -p gpurun_out/s3c
for rep in 1 2; do
for L in "" desktop2stereo_amd/libd2s_hip_plainc.so; do
  echo "== lib: ${L:-default}"
  D2S_LIB=$L python tools/warp_bench.py --batch 32 --modes Full-SBS Full-TAB Half-TAB --no-dibr --digest
  D2S_LIB=$L python tools/warp_bench.py --batch 1 --modes Full-SBS --no-dibr --digest
  D2S_LIB=$L python tools/warp_bench.py --batch 4 --hw 2160 3840 --modes Full-SBS Half-TAB --no-dibr --digest
done; done > gpurun_out/s3c/warp_ab.txt 2>&1
cat gpurun_out/s3c/warp_ab.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "warp or sbs or pipeline" 2>&1 | tail -3
