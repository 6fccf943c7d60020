mkdir -p gpurun_out/r4j
timeout 1200 python tools/soak_pp.py --reps 200 > gpurun_out/r4j/soak.log 2>&1; tail -25 gpurun_out/r4j/soak.log
D2S_LIB=$GRAFT_REPO_ROOT/desktop2stereo_amd/libd2s_hip_poison.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "pingpong_kernel or gemm_pp_probe or config5 or config2 or mixed" 2>&1 | tail -4
