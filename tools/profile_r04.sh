#!/bin/bash
# Round-4 profiles, run ON the GPU box.  Raw output goes to gpurun_out/prof_r04/<name>/; tools/summarize_r04.py condenses
# it into profiles/r04_*.  PMC passes are never combined with a trace domain; every pass is its own rocprofv3 run with the
# program itself behind `--`.
#   tools/profile_r04.sh calib                      the issue-calibration microkernels (tools/issue_calibration.hip)
#   tools/profile_r04.sh cfg3 cfg5 ...              workloads of tools/profile_workload.py (kernel trace + SQ / TCC passes)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r04
WL=${@:-calib cfg2 cfg3 cfg4 cfg5 cfg4b}
cd /tmp && export TMPDIR=/tmp
# SQ passes (8 slots each); a pass that names a counter this ROCm build does not know fails alone
SQ_A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
SQ_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
SQ_C="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"
SQ_D="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
SQ_E="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
SQ_F="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_IOPS"
for w in $WL; do
  mkdir -p $OUT/$w
  if [ "$w" = calib ]; then
    P="$ROOT/tools/bin/issue_calibration"
    $P > $OUT/$w/plain.json 2> $OUT/$w/plain.err || echo "calib plain run failed"
    rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
  else
    P="python3 $ROOT/tools/profile_workload.py $w"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace -- $P > $OUT/$w/trace.log 2>&1 || echo "$w trace failed"
    rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/$w/fetch -- $P > $OUT/$w/fetch.log 2>&1 || echo "$w fetch failed"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$w/write -- $P > $OUT/$w/write.log 2>&1 || echo "$w write failed"
  fi
  rocprofv3 --pmc $SQ_A --output-format csv -d $OUT/$w/sq_a -- $P > $OUT/$w/sq_a.log 2>&1 || echo "$w sq_a failed"
  rocprofv3 --pmc $SQ_B --output-format csv -d $OUT/$w/sq_b -- $P > $OUT/$w/sq_b.log 2>&1 || echo "$w sq_b failed"
  rocprofv3 --pmc $SQ_E --output-format csv -d $OUT/$w/sq_e -- $P > $OUT/$w/sq_e.log 2>&1 || echo "$w sq_e failed"
  rocprofv3 --pmc $SQ_F --output-format csv -d $OUT/$w/sq_f -- $P > $OUT/$w/sq_f.log 2>&1 || echo "$w sq_f failed"
  if [ "$w" = calib ] || [ "$w" = cfg3 ] || [ "$w" = cfg5 ] || [ "$w" = cfg4b ]; then
    rocprofv3 --pmc $SQ_C --output-format csv -d $OUT/$w/sq_c -- $P > $OUT/$w/sq_c.log 2>&1 || echo "$w sq_c failed"
    rocprofv3 --pmc $SQ_D --output-format csv -d $OUT/$w/sq_d -- $P > $OUT/$w/sq_d.log 2>&1 || echo "$w sq_d failed"
  fi
  echo "$w done"
done
python3 $ROOT/tools/summarize_r04.py $OUT $ROOT/gpurun_out/prof_r04_summary || echo "summary failed"
