#!/usr/bin/env python3
"""Where the time of the object API goes (List[Ray] in, List[Ray] out): cProfile of table.ray_tracing on
BASELINE cfg 2 with 1000 rays and on the 6-ray gaussian_beam scene."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(1, os.path.join(ROOT, "tests"))
import optable_amd as oa
import scenes

def run(build, reps):
    sc = build(oa)
    t = oa.OpticalTable(); t.add_components(sc["components"]); t.add_monitors(sc["monitors"])
    t.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])   # warm (library load, first upload)
    t0 = time.perf_counter()
    for _ in range(reps):
        t.rays = []
        out = t.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])
    dt = (time.perf_counter() - t0) / reps
    print(f"{build.__name__}: {len(sc['rays'])} rays -> {len(out)} segments, {dt * 1e3:.2f} ms per call")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(reps):
        t.rays = []
        t.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

run(scenes.g02_cfg2, 10)
run(scenes.g01_gaussian_beam, 50)
