#!/bin/bash
# A/B aid: views/s of `ptmain -gi -rb -rm --gpus all` with k view workers per GPU (PT_TUNE_WORKERS; 0 = the library's own choice).
# usage (on a GPU box): tools/workers_ab.sh "<scenes>" "<ks>" [views]      -> gpurun_out/workers_ab.txt
set -u
OUT=gpurun_out/workers_ab.txt; : > $OUT
for k in $2; do for s in $1; do
  mkdir -p gpurun_out/wab_o
  PT_TUNE_WORKERS=$k timeout -k 10 300 pathtracing_amd/ptmain $s -gi -rb -rm --views ${3:-160} --gpus all --out gpurun_out/wab_o > gpurun_out/wab.log 2>&1 || { echo "failed: $k $s"; tail -3 gpurun_out/wab.log; exit 1; }
  echo "workers $k $s: $(tail -n 1 gpurun_out/wab.log)" >> $OUT
  rm -rf gpurun_out/wab_o
done; done
cat $OUT
