timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp8.py -x -q -k "pingpong_kernel or gemm_pp_probe or fp8" 2>&1 | tail -2
timeout 600 python tools/pp_check.py 2>&1 | tail -3
D2S_PROF_DUMP=1 timeout 300 python tools/launch_dump.py --batch 32 > /tmp/dump32.log 2>&1
grep "d2s-prof" /tmp/dump32.log | awk '{print $2, $3, $4, $6, $8}' | sed -n 4,13p
tail -1 /tmp/dump32.log
D2S_LNF_PP=0 D2S_PROF_DUMP=1 timeout 300 python tools/launch_dump.py --batch 32 2>&1 | grep "d2s-prof" | awk '{print $2, $3, $4, $6, $8}' | sed -n 4,8p
