D2S_PROF_DUMP=1 timeout 300 python tools/launch_dump.py --batch 32 > /tmp/dump.log 2>&1
grep "d2s-prof" /tmp/dump.log | awk '{print $2, $3, $4, $6, $8}' | awk '$2=="gemm_linear" && $4 < 80 {printf "%s:%s us (%s GF); ", $1, $3, $4}'
echo
