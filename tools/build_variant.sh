#!/bin/bash
# An experimental build of the HIP library with extra compiler flags for one source file: pathtracing_amd/libptamd_<tag>.so (for tools/ab_lib.sh).
# usage: tools/build_variant.sh <tag> <file stem: pt_exact | pt_fast | pt_abi | pt_refine> <flags...>
set -e
TAG=$1; STEM=$2; shift 2
D=pathtracing_amd/csrc/device
C=-ffp-contract=off; [ "$STEM" = pt_fast ] && C=-ffp-contract=fast
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall $C "$@" -c -o /tmp/${STEM}_$TAG.o $D/$STEM.hip
OBJS=""
for s in pt_abi pt_exact pt_fast pt_refine; do if [ $s = $STEM ]; then OBJS="$OBJS /tmp/${STEM}_$TAG.o"; else OBJS="$OBJS $D/$s.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o pathtracing_amd/libptamd_$TAG.so $OBJS
echo built pathtracing_amd/libptamd_$TAG.so
