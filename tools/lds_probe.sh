#!/bin/bash
# LDS counters of one workload's dominant kernel (run on the GPU box): tools/lds_probe.sh [workload] [ENV=VAL ...]
W=${1:-cfg3}; shift
for kv in "$@"; do export "$kv"; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/ldsp -- python3 $ROOT/tools/profile_workload.py $W > /dev/null 2>&1
python3 - $ROOT <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/gpurun_out/ldsp/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if any(k in r["Kernel_Name"] for k in ("k_trace_", "k_gen_")):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    c = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(k, {n: f"{v:.3g}" for n, v in c.items()}, "LDS busy per CU %.2f" % (c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc),
          "conflict share %.2f" % (c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))
PY
