"""The single-precision contract of BASELINE cfg 3 / cfg 5 (both quoted in fp32), and the q contract through
aspheres.

fp32 cannot promise the fp64 surface sequence for EVERY ray over 20-50 bounces, but it can promise where rays may
leave it: only at decisions that are marginal in fp64 too.  `optable_amd.fp32_audit.audit` finds, for every ray whose
sequence differs, the first segment that ends on different leaves and measures how far the hit points lie from the
aperture edge of their leaves and how long the segment is.  Round 1 asserted >= 97 % / >= 90 % agreement without
looking at the rest; the audit showed 8 in 10^4 cfg 3 rays re-hitting the plane they had just left at t ~ 1e-5
(beyond the fp32 self-hit guard) — fixed by never testing a ray against the plane it starts on (trace_core.h).
What is left is asserted here: >= 99.95 % identical sequences, and EVERY other ray is an edge case.
Reference rules involved: optical_component.py:184-190 (|t| < 1e-9 guard), surfaces.py:144-171 (apertures)."""
import numpy as np
import pytest

import optable_amd as oa
from optable_amd import abi
from optable_amd import workloads as W
from optable_amd.batch import RayBatch
from optable_amd.fp32_audit import audit

pytestmark = pytest.mark.gpu
EDGE = 1e-4   # model units (scene size ~30): hit point within this of an aperture edge ...
SHORT = 1e-4  # ... or a segment this short (a corner: the next face lies inside the fp32 self-hit guard of 1e-5 x a few)


@pytest.mark.parametrize("name,n", [("cfg3", 100_000), ("cfg5", 30_000)])
def test_fp32_leaves_the_fp64_path_only_at_edge_cases(name, n):
    wl = W.baseline_workloads(oa)[name]
    table = oa.OpticalTable()
    table.add_components(wl.components())
    o, d, lam = wl.rays(n, 0)
    q = 1j * np.pi * W.W0**2 / lam
    K = wl.max_segments
    s64 = table.trace_batch(RayBatch.from_arrays(o, d, wavelength=lam, q=q), max_segments=K)
    s32 = table.trace_batch(RayBatch.from_arrays(o, d, wavelength=lam, q=q, precision="f32"), max_segments=K)
    s64, s32 = s64.as_kray_slots(K), s32.as_kray_slots(K)  # (the default layout of these scenes is the dense append-order list)
    rep = audit(table.compile(), s64, s32, K)
    assert rep["same"].mean() >= 0.9995, rep["same"].mean()
    edge = (rep["margin"] < EDGE) | (np.minimum(rep["len64"], rep["len32"]) < SHORT)
    assert edge.all(), [(int(r), int(k), float(m)) for r, k, m in zip(rep["ray"][~edge], rep["kstar"][~edge], rep["margin"][~edge])]
    # on the rays that agree, every segment start agrees to 2e-3 (20-50 bounces, coordinates ~30, fp32 ulp 2e-6)
    c64 = np.abs(s64.count.cpu().numpy())
    valid = (np.arange(K)[:, None] < c64[None, :]) & rep["same"][None, :]
    for f in ("ox", "oy", "oz"):
        with np.errstate(invalid="ignore"):
            a = s64.field(f).cpu().numpy().reshape(K, n)
            b = s32.field(f).cpu().numpy().reshape(K, n).astype(np.float64)
        assert np.abs(a - b)[valid].max() < 2e-3, f


def test_asphere_q_contract(oracle):
    """q against the oracle on cfg 5: 1e-9 on every segment whose q has not passed an aspheric interface yet; after
    that the reference's own arithmetic is ill-conditioned — ASphere.roc is a 3-point finite difference with
    h = 1e-4 * radius (surfaces.py:355-369), which amplifies rounding by 1/h^2 — so the bound is what that algorithm
    itself does under a last-digit change of its input: the oracle is re-run on inputs moved by a few ulps and the
    spread of ITS q per segment (times a safety factor) is the tolerance, instead of round 1's flat 2e-3."""
    n, K = 3000, 50
    table = oa.OpticalTable()
    table.add_components(W.cfg5_components(oa))
    o, d = W.cfg5_rays(n, 3)
    o[n // 2:, 0] = 8.0  # half of the rays start between the lens and the micro-mirror array: their q meets spherical
    #                      caps (constant ROC) and the flat back of the array before it ever meets an asphere
    q0 = 1j * np.pi * W.W0**2 / W.WL
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=q0)
    scene = table.compile()
    got = table.trace_batch(batch, max_segments=K).to_host(reference_order=True)
    host = batch.to_host()
    ref = oracle.trace(scene, host, max_trace_num=K)
    np.testing.assert_array_equal(got["surface"], ref["surface"])
    # spread of the reference algorithm under input rounding: origins moved by +-2 ulp in y / z
    spread = np.zeros(len(ref["ray"]))
    for sy, sz in ((1, 1), (-1, 1), (1, -1)):
        moved = dict(host)
        moved["oy"] = host["oy"] * (1 + sy * 4.4e-16)
        moved["oz"] = host["oz"] * (1 + sz * 4.4e-16)
        alt = oracle.trace(scene, moved, max_trace_num=K)
        np.testing.assert_array_equal(alt["surface"], ref["surface"])
        spread = np.maximum(spread, np.hypot(alt["q_re"] - ref["q_re"], alt["q_im"] - ref["q_im"]))
    # segments before the ray's q has met an asphere: leaves whose radius of curvature comes from ASphere.roc
    asph = np.array([type(c.surface).__name__ == "ASphere" for c in scene.leaves])
    hit_asph = np.where(ref["surface"] >= 0, asph[np.clip(ref["surface"], 0, None)], False)
    first = np.zeros(len(hit_asph), dtype=bool)  # True once an earlier segment of the same ray ended on an asphere
    seen = {}
    for s, (ray, h) in enumerate(zip(ref["ray"].tolist(), hit_asph.tolist())):
        first[s] = seen.get(ray, False)
        if h:
            seen[ray] = True
    err = np.hypot(got["q_re"] - ref["q_re"], got["q_im"] - ref["q_im"])
    mag = np.hypot(ref["q_re"], ref["q_im"])
    clean = ~first
    assert clean.sum() > 1.5 * n  # segment 0 of every ray, and the first two segments of the rays that start behind the lens
    assert np.all(err[clean] <= 1e-9 * mag[clean] + 1e-9), float((err[clean] / mag[clean]).max())
    assert np.all(err[first] <= 50 * spread[first] + 1e-6 * mag[first]), float((err[first] / (50 * spread[first] + 1e-6 * mag[first])).max())
    for f in abi.SEG_FIELDS:  # everything that is not q stays at 1e-9
        if f not in ("q_re", "q_im"):
            np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)
