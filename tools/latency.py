#!/usr/bin/env python3
"""Call latency of table.ray_tracing at the reference's own sizes (bench.py `latency`), alone, with a cProfile of the slowest case.
    python tools/latency.py [--profile]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import json

import optable_amd as oa
import bench

for rec in bench.survey_latency(oa):
    print(json.dumps(rec))
if "--profile" in sys.argv:
    import cProfile
    import pstats
    from optable_amd import workloads as W

    comps, rays = W.chromatic_scene(oa)
    table = oa.OpticalTable()
    table.add_components(comps)
    for _ in range(5):
        table.rays = []
        table.ray_tracing(rays)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        table.rays = []
        table.ray_tracing(rays)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
