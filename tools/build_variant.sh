#!/bin/bash
# Build a VARIANT of libd2s_hip.so for A/B or instrumentation runs: one source recompiled with extra defines, the other objects of the
# regular build reused.   tools/build_variant.sh <name> <source.hip> "<defines>"   ->  desktop2stereo_amd/libd2s_hip_<name>.so
# (select it with D2S_LIB=...; *.so files are git-ignored but travel with gpurun snapshots)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; defs=$3
B=desktop2stereo_amd/csrc/_build
python -m desktop2stereo_amd.build > /dev/null
extra=""
case $src in frame_ops.hip|post.hip|ingest.hip|dibr.hip) extra="-ffp-contract=off";; attention.hip) extra="-fno-honor-nans -mno-amdgpu-ieee";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 $extra $defs -c desktop2stereo_amd/csrc/$src -o /tmp/variant_$name.o
objs=""
for o in $B/*.o; do [ "$(basename $o .o)" = "$(basename $src .hip)" ] && objs="$objs /tmp/variant_$name.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o desktop2stereo_amd/libd2s_hip_$name.so $objs
echo desktop2stereo_amd/libd2s_hip_$name.so
