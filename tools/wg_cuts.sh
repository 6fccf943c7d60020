cd ${GRAFT_REPO_ROOT:-.}
L=gpurun_out/r6p_warp.log; : > $L
python tools/dbg_rows.py >> $L 2>&1
D2S_WARP_WPS=5 D2S_WARP_WPC=20 python tools/dbg_rows.py >> $L 2>&1
for wps in 4 5; do for wpc in $((wps*4)) $((wps*8)); do echo "WPS=$wps WPC=$wpc" >> $L; D2S_WARP_WPS=$wps D2S_WARP_WPC=$wpc python tools/warp_ab.py --batches 32 2>&1 | grep 1920 >> $L; done; done
D2S_WARP_WPS=5 D2S_WARP_WPC=20 python tools/warp_ab.py --batches 1 --n 300 >> $L 2>&1
D2S_WARP_WPS=5 D2S_WARP_WPC=20 python tools/warp_ab.py --batches 2 --hw 2160 3840 --n 30 >> $L 2>&1
grep -v amdgpu.ids $L
