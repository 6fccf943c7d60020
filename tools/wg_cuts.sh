cd ${GRAFT_REPO_ROOT:-.}
L=gpurun_out/r6n_warp.log; : > $L
echo "== strict lib" >> $L
D2S_LIB=$PWD/desktop2stereo_amd/libd2s_hip_wg_strict.so python tools/dbg_rows.py >> $L 2>&1
echo "== normal lib" >> $L
python tools/dbg_rows.py >> $L 2>&1
python tools/warp_ab.py >> $L 2>&1
python tools/warp_ab.py --batches 32 >> $L 2>&1
for wps in 4 5; do for wpc in $((wps*4)) $((wps*8)); do echo "WPS=$wps WPC=$wpc" >> $L; D2S_WARP_WPS=$wps D2S_WARP_WPC=$wpc python tools/warp_ab.py --batches 32 --modes Full-SBS 2>&1 | grep Full >> $L; done; done
python tools/warp_ab.py --batches 8 --ratio 40 --kind structured --n 20 >> $L 2>&1
python tools/warp_ab.py --batches 2 --hw 2160 3840 --n 30 >> $L 2>&1
python tools/warp_ab.py --batches 5 --hw 1440 2560 --n 30 >> $L 2>&1
grep -v amdgpu.ids $L
