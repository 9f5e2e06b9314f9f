#!/bin/bash
# SQ-counter profile of selected configs of tools/bench_configs.py: tools/profile_sq.sh <tag> ENV=VAL ...
# (counters only; rocprofv3 --pmc passes are kept apart from any trace domain)
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/a -- python3 $ROOT/tools/bench_configs.py > $OUT/a.log 2>&1 || echo a failed
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/b -- python3 $ROOT/tools/bench_configs.py > $OUT/b.log 2>&1 || echo b failed
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_vgpr"] = [float(r["VGPR_Count"])]
        acc[k]["_grid"] = [float(r["Grid_Size"])]
for k, c in acc.items():
    if "k_trace" not in k and "k_gen" not in k:
        continue
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:24s} {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
