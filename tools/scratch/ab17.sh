for T in 0 3264 64648 641288; do
echo "CONV_TILE=$T"
D2S_CONV_TILE=$T D2S_PROF_DUMP=1 python tools/launch_dump.py --batch 1 2>&1 | grep "d2s-prof" | awk '$2>=69 && $3=="gemm_conv3x3" {print $2, $4}' | tr '\n' ';'; echo
done
