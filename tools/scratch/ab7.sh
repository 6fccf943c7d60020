python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
for B in 1 1 4 32; do
echo "B=$B: $(python bench.py --batch $B --steps 200 --warmup 30 --no-cpu-baseline --no-parity-class --no-profile --also-batch 0 --sink-quality 0 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["value"])')"
done
echo "bx3 B=1: $(python bench.py --precision bf16x3 --batch 1 --steps 200 --warmup 30 --no-cpu-baseline --no-parity-class --no-profile --also-batch 0 --sink-quality 0 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["value"])')"
