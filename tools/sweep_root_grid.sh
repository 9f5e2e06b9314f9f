#!/bin/bash
# Top-level grid of cfg 3 (pair-queue kernel, append layout): cells per component x cell aspect.  One line per setting.
# usage (GPU box): bash tools/sweep_root_grid.sh > gpurun_out/sweep_root_grid.log
export ONLY=cfg3 PREC=f32 LAYOUT=append N3=${N3:-10000000}
for rgc in 0.5 0.75 1 1.5 2 3; do
  for rga in 0.5 1 2; do
    echo "RGC=$rgc RGA=$rga: $(RGC=$rgc RGA=$rga python tools/bench_configs.py 2>/dev/null | grep 'cfg3' | tail -1)"
  done
done
