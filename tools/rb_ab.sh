#!/bin/bash
# A/B aid: views/s of the -gi -rb multi-view job at the reference's settings (car, cbox) and of the atrium bench with -rb at 16 spp
for s in car cbox; do timeout -k 10 300 python3 -m pathtracing_amd.rm $s -gi -rb --views 128 --out gpurun_out/rm_tmp | tail -n 1 || exit 1; done; rm -rf gpurun_out/rm_tmp
timeout -k 10 400 python3 bench.py --rb --spp 16 --steps 32 --warmup 4 --no-cpu-baseline --no-rm-leg | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('atrium -rb 16 spp: views/s', d['views_per_s'], 'ms/view', d['ms_per_step'])"
