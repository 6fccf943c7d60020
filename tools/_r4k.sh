timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_vda.py -x -q -k "full_size or config or vitb or conv_kernel" 2>&1 | tail -4
D2S_PROF_DUMP=1 timeout 300 python tools/launch_dump.py --batch 32 > /tmp/dump.log 2>&1
grep "d2s-prof" /tmp/dump.log | awk '{print $2, $3, $4, $6, $8}' | awk '$2=="gemm_linear" && $4 < 25 {printf "%s:%s us (%s GF); ", $1, $3, $4}'; echo; tail -1 /tmp/dump.log
for i in 1 2; do for v in 0 1; do echo "NO_SK=$v"; D2S_NO_SK=$v timeout 300 python tools/launch_dump.py --batch 1 2>&1 | tail -1 | sed 's/.*sum ms/sum ms/'; done; done
for v in 0 1; do echo "NO_SK=$v"; D2S_NO_SK=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity-class --sink-quality 0 --also-batch 0 --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch1', d['value'], d['ms_per_step'])"; done
