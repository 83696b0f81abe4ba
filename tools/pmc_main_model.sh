set -u
OUT=gpurun_out/pmc_main; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcp -o p -- python3 bench.py --no-cpu-baseline --no-rm-leg --model main --steps 3 --warmup 1 > $OUT/tcp.log 2>&1 || { tail -3 $OUT/tcp.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU TCC_HIT_sum TCC_REQ_sum --output-format csv -d $OUT/sq -o p -- python3 bench.py --no-cpu-baseline --no-rm-leg --model main --steps 3 --warmup 1 > $OUT/sq.log 2>&1 || { tail -3 $OUT/sq.log; exit 1; }
for k in "k_shadow_gi" "k_trace_closest<false"; do python3 tools/pmc_summary.py $OUT "$k"; done
rm -rf $OUT/tcp $OUT/sq
