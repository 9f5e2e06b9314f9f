#!/usr/bin/env python3
"""Call latency of table.ray_tracing on the scenes with user-defined parts (tests/scenes.py g25: user interact_local methods,
g26: user Surface classes), next to the reference's own time for the same call (measured in the build container, 3 calls:
g25 360-380 ms, g26 72-74 ms).  `--profile`: cProfile of the hooked scene.
    python tools/latency_user.py [--profile]"""
import contextlib
import io
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))
import optable_amd as oa
import scenes

REFERENCE_MS = {"g25_user_components": 365.0, "g26_user_surfaces": 73.0}
for name, ref_ms in REFERENCE_MS.items():
    sc = {**scenes.SCENES, **scenes.HOOKED_SCENES}[name](oa)
    table = oa.OpticalTable()
    table.add_components(sc["components"])
    times = []
    for it in range(25):
        table.rays = []
        for c in table.compile().limited:
            c._interact_count = {}
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = table.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])
        times.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"scene": name, "segments": len(out), "first_call_ms": round(times[0], 2),
                      "median_ms_per_call": round(statistics.median(times[5:]), 2), "reference_python_ms": ref_ms,
                      "speedup_vs_reference": round(ref_ms / statistics.median(times[5:]), 1)}))
if "--profile" in sys.argv:
    import cProfile
    import pstats

    sc = scenes.HOOKED_SCENES["g25_user_components"](oa)
    table = oa.OpticalTable()
    table.add_components(sc["components"])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        table.rays = []
        with contextlib.redirect_stdout(io.StringIO()):
            table.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
