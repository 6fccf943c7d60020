#!/bin/bash
# compile ONE csrc file alone -- host AND device pass, as the build does (errors shown) -- keep the device asm under /tmp/t and print the
# resource lines of the kernels matching $2:   tools/cc_one.sh dibr rows_kernel
mkdir -p /tmp/t && cd /root/repo/desktop2stereo_amd/csrc || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-tautological-compare -Wno-int-to-pointer-cast -Wno-ignored-attributes -mllvm -amdgpu-kernarg-preload-count=16 -ffp-contract=off $D2S_HIPCC_DEFS"
/opt/rocm/bin/hipcc $F -c $1.hip -o /tmp/t/$1.o 2>&1 | grep -B2 -A8 "error" | head -40
/opt/rocm/bin/hipcc $F -S --cuda-device-only $1.hip -o /tmp/t/$1.s 2>/dev/null
grep -n "; NumVgprs\|; ScratchSize\|; Occupancy\|codeLenInByte\|^_Z.*:" /tmp/t/$1.s | grep -A4 "${2:-kernel}" | grep -v "^--"
