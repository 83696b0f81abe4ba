#!/bin/bash
# Round profile: kernel trace + stats and PMC passes of `bench.py` for one workload.
# usage (on the GPU box): tools/make_profile.sh <workload> <tag>     -> gpurun_out/profile_<tag>/
set -u
W=${1:-atrium}
TAG=${2:-$W}
OUT=gpurun_out/profile_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --workload $W --no-cpu-baseline --no-rm-leg ${PROFILE_ARGS:-}"      # the default run of bench.py minus the CPU leg (PROFILE_ARGS adds options)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- python3 $ARGS > "$OUT/trace.log" 2>&1 || { echo "kernel trace failed"; exit 1; }
grep "^{" "$OUT/trace.log" > "$OUT/bench_under_trace.json"
run() {
  name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE && \
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr
# header lines: the digest of the kernel sources the counters belong to (bench.py flags a collection as stale when the
# sources have changed since) and the rays one launch processed
HDR=$(python3 -c "
import json, bench
j = json.load(open('$OUT/bench_under_trace.json'))
print('# kernel_source_hash', bench.kernel_source_hash())
print('# rays_per_launch', j['roofline']['rays_per_launch'])")
{ echo "$HDR"; python3 tools/pmc_summary.py "$OUT" "k_trace_closest<false"; } > "$OUT/pmc_k_trace_closest.txt"
{ echo "$HDR"; for k in "k_primary_search" "k_primary_shade" "k_primary_again" "k_packet_prep" "k_mirror_exact"; do python3 tools/pmc_summary.py "$OUT" "$k"; done; } > "$OUT/pmc_k_primary.txt"
{ echo "$HDR"; python3 tools/pmc_summary.py "$OUT" "k_shade_sum_omp"; python3 tools/pmc_summary.py "$OUT" "k_gi_gen"; } > "$OUT/pmc_k_shade_gen.txt"
ls "$OUT/trace"
