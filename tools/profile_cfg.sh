#!/bin/bash
# PMC profile of one config of tools/bench_configs.py: tools/profile_cfg.sh <tag> ENV=VAL ...
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_configs.py > $OUT/trace.log 2>&1 || echo trace failed
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 $ROOT/tools/bench_configs.py > $OUT/sq.log 2>&1 || echo sq failed
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- python3 $ROOT/tools/bench_configs.py > $OUT/lds.log 2>&1 || echo lds failed
grep -E "^cfg" $OUT/trace.log
