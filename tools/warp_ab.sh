# same-box A/B of warp builds: tools/warp_ab.sh "<lib> <lib> ..."  ("" = the regular library)
mkdir -p gpurun_out/s3c
for rep in 1 2; do
for L in ${LIBS:-default plainc}; do
  [ $L = default ] && P="" || P=desktop2stereo_amd/libd2s_hip_$L.so
  echo "== lib: $L"
  D2S_LIB=$P python tools/warp_bench.py --batch 32 --modes ${MODES:-Full-SBS Full-TAB Half-TAB} --no-dibr --digest
  D2S_LIB=$P python tools/warp_bench.py --batch 1 --modes Full-SBS --no-dibr --digest
  [ -n "${QUICK:-}" ] || D2S_LIB=$P python tools/warp_bench.py --batch 4 --hw 2160 3840 --modes Full-SBS Half-TAB --no-dibr --digest
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s3c/warp_ab.txt
