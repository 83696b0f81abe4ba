#!/bin/bash
# Two SQ counter passes of a short bench.py run, summarised for the kernels whose names contain the given strings.
# usage (on the GPU box): tools/pmc_quick.sh <tag> "<kernel substr> [<kernel substr> ...]" <bench.py arguments...>   -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; KERNELS=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py --no-cpu-baseline --no-rm-leg $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
ARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE && \
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA || exit 1
for k in $KERNELS; do python3 tools/pmc_summary.py "$OUT" "$k"; done > "$OUT/summary.txt"
rm -rf "$OUT/sq1" "$OUT/sq2"
cat "$OUT/summary.txt"
