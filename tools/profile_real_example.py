#!/usr/bin/env python3
"""The reference's largest example (fixture g27) traced five times: the program rocprofv3 profiles (tools/profile_r04.sh style passes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optable_amd as oa
import bench

for _ in range(1):
    rec = bench.real_example_latency(oa)
print({k: rec[k] for k in ("segments", "median_ms_per_call", "segments_match_the_reference")})
