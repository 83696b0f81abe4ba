#!/bin/bash
# On a GPU box, with pathtracing_amd/libptamd_tclk.so built by `tools/build_variant.sh tclk pt_fast -DPT_TRACE_CLOCKS`:
# when the waves of the closest-hit launches begin and end (percentiles, microseconds after the first wave's start).
# usage: tools/trace_exit.sh [bench.py arguments; default --rb --spp 16]
ARGS=${*:---rb --spp 16}
cp pathtracing_amd/libptamd.so /tmp/libptamd_base.so && cp pathtracing_amd/libptamd_tclk.so pathtracing_amd/libptamd.so
timeout -k 10 300 python3 bench.py $ARGS --steps 6 --warmup 2 --no-cpu-baseline --no-rm-leg 2> gpurun_out/trace_clk.err > /dev/null
rc=$?
cp /tmp/libptamd_base.so pathtracing_amd/libptamd.so
[ $rc -eq 0 ] || { tail gpurun_out/trace_clk.err; exit 1; }
grep trace_clk gpurun_out/trace_clk.err | tail -6
