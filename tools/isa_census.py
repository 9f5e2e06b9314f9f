#!/usr/bin/env python3
"""Static instruction census of one kernel by phase.  Build the assembly with the stamp sites turned into comments:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=on \\
          -DOT_MARK -DOT_REAL=float -DOT_APPEND=1 --cuda-device-only -S optable_amd/csrc/inst_rolling.hip -o /tmp/mark.s
    python tools/isa_census.py /tmp/mark.s 'k_trace_rollingIfLj1180ELb1ELb1ELb1E9SegPlanes'
Instructions are charged to the last marker above them IN FILE ORDER (block placement follows the source only roughly:
read the numbers as a map of where the code is, not as an execution profile)."""
import collections
import re
import sys

path, pattern = sys.argv[1], sys.argv[2]
inside, region = False, "prologue"
order, tally = [], collections.defaultdict(collections.Counter)
for line in open(path):
    if re.match(r"^_Z\w+:", line):
        inside = re.search(pattern, line) is not None
        region = "prologue"
        continue
    if not inside:
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        inside = False
        continue
    m = re.search(r"; OT_MARK (\w+ \d+)", line)
    if m:
        region = m.group(1) + f" @{len(order)}"
        order.append(region)
        continue
    m = re.match(r"^\t([a-z_0-9]+)", line)
    if not m:
        continue
    op = m.group(1)
    if region not in tally:
        if region not in order:
            order.append(region)
    kind = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
            "smem" if op.startswith("s_load") or op.startswith("s_buffer") else
            "wait" if op.startswith(("s_waitcnt", "s_nop")) else "branch" if op.startswith(("s_cbranch", "s_branch")) else
            "salu" if op.startswith("s_") else "other")
    tally[region][kind] += 1
    if op in ("v_readlane_b32", "v_writelane_b32"):
        tally[region]["lane_rw"] += 1
    if op in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_lo_i32") or op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_div_", "v_cvt_")):
        tally[region]["slow_or_cvt"] += 1
cols = ["valu", "salu", "lds", "vmem", "smem", "branch", "wait", "lane_rw", "slow_or_cvt"]
print(f"{'region':24s}" + "".join(f"{c:>12s}" for c in cols))
tot = collections.Counter()
for r in order:
    if r in tally:
        print(f"{r:24s}" + "".join(f"{tally[r][c]:12d}" for c in cols))
        tot.update(tally[r])
print(f"{'total':24s}" + "".join(f"{tot[c]:12d}" for c in cols))
