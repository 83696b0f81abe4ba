#!/bin/bash
# On a GPU box: rate of the closest-hit kernel against the size of its launch (samples per pixel of a one-pass view).
# usage: tools/spp_sweep.sh [bench.py arguments, e.g. --workload car]
for spp in 4 8 16 32 64 128 256; do
timeout -k 10 200 python3 bench.py "$@" --spp $spp --steps 24 --warmup 4 --no-cpu-baseline --no-rm-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
n=r['rays_per_launch']; k=r['kernel_mrays_per_s']
print('spp %4d  rays/launch %9d  kernel %8.1f Mrays/s  %8.1f us/launch   whole job %8.1f Mrays/s  %6.3f ms/view' % ($spp, n, k, n / k, d['value'], d['ms_per_step']))" || exit 1
done
