#!/bin/bash
# PMC counter passes over one short bench run (one rocprofv3 run per pass, each
# within the per-block counter limits).  usage: tools/pmc_passes.sh <workload> <outdir>
set -u
W=${1:-atrium}
OUT=${2:-gpurun_out/pmc_$W}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
    python3 bench.py --workload "$W" --steps 1 --warmup 0 --spp 64 --no-cpu-baseline > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE && \
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr
