#!/bin/bash
# A/B of library builds on one GPU box: bench.py with pathtracing_amd/libptamd.so, then with each pathtracing_amd/libptamd_<tag>.so
# put in its place (experimental builds made beforehand; *.so files are not tracked).  Two rounds, to see the run-to-run noise.
# usage: tools/ab_lib.sh "<bench args>" <tag> [<tag> ...]      -> gpurun_out/ab_lib.txt
set -u
ARGS=$1; shift
OUT=gpurun_out/ab_lib.txt; mkdir -p gpurun_out; : > $OUT
cp pathtracing_amd/libptamd.so /tmp/libptamd_base.so
one() {
  line=$(timeout -k 10 240 python3 -u bench.py --no-cpu-baseline --no-rm-leg $ARGS 2>/dev/null | grep '^{' | tail -n 1) || { echo "$1: failed" >> $OUT; return 1; }
  python3 -c "
import json, sys
j = json.loads(sys.argv[2])
print('%-24s value %8.1f  ms/step %7.2f  kernel %8.1f' % (sys.argv[1], j['value'], j['ms_per_step'], j['roofline'].get('kernel_mrays_per_s') or 0))" "$1" "$line" >> $OUT
}
for round in 1 2; do
  cp /tmp/libptamd_base.so pathtracing_amd/libptamd.so; one base || break
  for t in "$@"; do cp pathtracing_amd/libptamd_$t.so pathtracing_amd/libptamd.so; one $t || break 2; done
done
cp /tmp/libptamd_base.so pathtracing_amd/libptamd.so
cat $OUT
