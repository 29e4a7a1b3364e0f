#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag> <dtype> [extra bench args]
# -> gpurun_out/<tag>/{kernel_stats.csv, pmc_step_by_kernel.csv(+.meta.json), bench.json, bench_breakdown.txt}
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass) and never combined with a trace.
set -u
TAG=$1; DT=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
COMMIT=$(cat .git_head 2>/dev/null || echo unknown)
timeout -k 10 600 python3 bench.py --dtype $DT --breakdown --secondary "" "$@" > $OUT/bench.json 2> $OUT/bench_breakdown.txt || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/stats -- python3 bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --secondary "" --sustained-steps 0 "$@" > /dev/null 2>&1 || exit 1
cp $(ls /tmp/prof_$TAG/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
for P in "sq:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  N=${P%%:*}; C=${P#*:}
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$TAG/$N -- python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --secondary "" --sustained-steps 0 "$@" > /dev/null 2>&1 || exit 1
done
python3 tools/summarize_step_pmc.py /tmp/prof_$TAG/sq /tmp/prof_$TAG/fetch /tmp/prof_$TAG/write $OUT/pmc_step_by_kernel.csv \
  batch=32 height=480 width=640 arch=dlapawdl3new_34 dtype=$DT commit=$COMMIT > $OUT/pmc_summary.txt 2>&1
tail -30 $OUT/pmc_summary.txt
