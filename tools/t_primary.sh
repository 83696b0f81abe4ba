#!/bin/bash
# A/B aid for the fp64 primary pass on a GPU box: the GPU tests, views/s of the -gi -rb job, per-kernel times of the atrium -rb view.
# usage: tools/t_primary.sh <tag> [notest]
TAG=$1
if [ "$2" != "notest" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/${TAG}_pytest.log; exit 1; }
  tail -2 gpurun_out/${TAG}_pytest.log
fi
bash tools/rb_ab.sh > gpurun_out/${TAG}_rb_ab.log 2>&1 || { tail gpurun_out/${TAG}_rb_ab.log; exit 1; }
grep -v amdgpu.ids gpurun_out/${TAG}_rb_ab.log
bash tools/kernel_times.sh ${TAG} --rb --spp 16 --steps 32 --warmup 4 > gpurun_out/${TAG}_kt.log 2>&1 || { tail gpurun_out/${TAG}_kt.log; exit 1; }
head -8 gpurun_out/${TAG}_kt.log
