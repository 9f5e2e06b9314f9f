"""f4 on the device path: the files this package writes after a GPU trace against the files the REFERENCE wrote
for the same scenes (fixture g23, tools/make_golden.py): optical_table.py:487-500 export_rays_csv and
monitor.py:255-269 export_rays_npz.  Field by field at the parity tolerance, same rows, same `None`s."""
import os
import time

import numpy as np
import pytest

import optable_amd as oa
from helpers import RTOL, build, golden
from optable_amd import export
from optable_amd import workloads as W
from optable_amd.batch import RayBatch

pytestmark = pytest.mark.gpu


def _write_ref(tmp_path, lines):
    path = os.path.join(tmp_path, "reference.csv")
    open(path, "w").write("\n".join(lines) + "\n")
    return path


def _same_file(got_path, want_path):
    got, want = open(got_path).read().splitlines(), open(want_path).read().splitlines()
    assert len(got) == len(want) and got[0] == want[0]
    assert [ln.count("None") for ln in got] == [ln.count("None") for ln in want]
    a, b = export.parse_rays_csv(got_path), export.parse_rays_csv(want_path)
    for key in a:
        np.testing.assert_allclose(a[key], b[key], rtol=RTOL, atol=1e-9, equal_nan=True, err_msg=key)


@pytest.mark.parametrize("name", ["g01_gaussian_beam", "g06_mirror_pair", "g12_dove"])
def test_exports_match_the_files_the_reference_wrote(name, tmp_path):
    fix = golden("g23_exports")
    table, sc = build(name)
    table.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])
    path = os.path.join(tmp_path, "rays.csv")
    table.export_rays_csv(path)
    _same_file(path, _write_ref(tmp_path, fix[name + "_csv"].tolist()))
    for m, mon in enumerate(table.monitors):
        mpath = os.path.join(tmp_path, f"mon{m}.npz")
        mon.export_rays_npz(mpath)
        got = np.load(mpath)
        for key in ("xList", "yList", "tXList", "tYList", "IList"):
            np.testing.assert_allclose(got[key], fix[f"{name}_mon{m}_{key}"], rtol=RTOL, atol=1e-9, err_msg=f"mon{m} {key}")


def test_batch_exports_equal_the_object_api_files(tmp_path):
    """The same scene through `trace_batch`: CSV from the SegmentBatch columns, npz from MonitorHits tensors."""
    fix = golden("g23_exports")
    name = "g06_mirror_pair"
    table, sc = build(name)
    rays = sc["rays"]
    batch = RayBatch.from_arrays([r.origin for r in rays], [r.direction for r in rays])
    batch.flags.zero_()  # these rays carry no Gaussian q
    segs = table.trace_batch(batch, max_segments=16)
    path = os.path.join(tmp_path, "batch.csv")
    table.export_batch_csv(segs, path, rays=batch)
    _same_file(path, _write_ref(tmp_path, fix[name + "_csv"].tolist()))
    for m, mon in enumerate(table.monitors):
        hits = table.record_batch(mon, segs)
        mpath = os.path.join(tmp_path, f"hits{m}.npz")
        hits.export_rays_npz(mpath)
        got = np.load(mpath)
        for key in ("xList", "yList", "tXList", "tYList", "IList"):
            np.testing.assert_allclose(got[key], fix[f"{name}_mon{m}_{key}"], rtol=RTOL, atol=1e-9, err_msg=f"mon{m} {key}")


def test_million_segment_export_takes_seconds(tmp_path):
    n = 200_000
    table = oa.OpticalTable()
    table.add_components(W.cfg2_components(oa))
    o, d = W.cfg2_rays(n, 0)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL)
    segs = table.trace_batch(batch, max_segments=5)
    path = os.path.join(tmp_path, "big.csv")
    t0 = time.perf_counter()
    table.export_batch_csv(segs, path, rays=batch)
    dt = time.perf_counter() - t0
    with open(path) as fh:
        assert sum(1 for _ in fh) == 5 * n + 1
    assert dt < 60, f"{dt:.1f} s for 1e6 segments"
    back = export.parse_rays_csv(path)  # round trip of the first rows
    host = segs.to_host(reference_order=True)
    np.testing.assert_allclose(back["origin"][:1000, 1], host["oy"][:1000], rtol=1e-15, atol=0)
    np.testing.assert_allclose(back["q"][:1000].imag, host["q_im"][:1000], rtol=1e-15)
