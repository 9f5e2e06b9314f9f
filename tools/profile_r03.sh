#!/bin/bash
# Round-3 profiles, run ON the GPU box: for every workload one rocprofv3 kernel-trace/stats run and separate --pmc
# passes (SQ issue / wait counters, FETCH_SIZE, WRITE_SIZE: never combined with a trace domain).  Raw output goes to
# gpurun_out/prof_r03/<workload>/; tools/summarize_r03.py condenses it into profiles/r03_<workload>_*.
# Output layouts as the bench measures them (tools/profile_workload.py): cfg2 / cfg4 tiled, cfg3 / cfg5 append.
#   tools/profile_r03.sh [workloads...]      default: cfg2 cfg3 cfg4 cfg5 cfg4b monitor
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r03
WL=${@:-cfg2 cfg3 cfg4 cfg5 cfg4b monitor}
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  mkdir -p $OUT/$w
  P="python3 $ROOT/tools/profile_workload.py $w"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace -- $P > $OUT/$w/trace.log 2>&1 || echo "$w trace failed"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/$w/sq -- $P > $OUT/$w/sq.log 2>&1 || echo "$w sq failed"
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/$w/fetch -- $P > $OUT/$w/fetch.log 2>&1 || echo "$w fetch failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$w/write -- $P > $OUT/$w/write.log 2>&1 || echo "$w write failed"
  echo "$w done: $(tail -1 $OUT/$w/trace.log)"
done
python3 $ROOT/tools/summarize_r03.py $OUT $ROOT/gpurun_out/prof_r03_summary
