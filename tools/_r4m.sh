for bpc in 4 6 9 18; do echo "BPC=$bpc"; D2S_WARP_BPC=$bpc python tools/warp_bench.py --batch 1 --modes Full-SBS --no-dibr 2>&1 | tail -1; done
for bpc in 4 6; do echo "BPC=$bpc b32"; D2S_WARP_BPC=$bpc python tools/warp_bench.py --batch 32 --modes Full-SBS --no-dibr 2>&1 | tail -1; done
