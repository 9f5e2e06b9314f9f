"""Shared test plumbing: build a parity scene with optable_amd, pack its rays, compare segment
streams against the committed golden fixtures (tests/golden/*.npz, generated from the
reference by tools/make_golden.py)."""
import ctypes as C
import os

import numpy as np

import optable_amd as oa
from optable_amd import abi
import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-6  # north_star: endpoints / directions / q within 1e-6 relative (fp64)


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def build(name, **kw):
    np.random.seed(12345)
    sc = {**scenes.SCENES, **scenes.HOOKED_SCENES}[name](oa, **kw)
    table = oa.OpticalTable()
    table.add_components(sc["components"])
    table.add_monitors(sc["monitors"])
    return table, sc


def pack_host(rays, scene_unit=1e-2):
    """List[Ray] -> dict of host arrays in the ot_rays layout (+ class ids).  Wavelengths are expressed in
    the scene's length unit (a ray may carry its own `unit`; the reference uses wavelength*ray.unit)."""
    ids = [r._id for r in rays]
    class_of = {}
    cls = np.array([class_of.setdefault(i, len(class_of)) for i in ids], dtype=np.int32)
    n = len(rays)
    o = np.array([r.origin for r in rays], dtype=float).reshape(n, 3)
    d = np.array([r.direction for r in rays], dtype=float).reshape(n, 3)
    has_q = np.array([r.qo is not None for r in rays])
    q = np.array([complex(r.qo) if r.qo is not None else 0j for r in rays])
    host = dict(ox=o[:, 0], oy=o[:, 1], oz=o[:, 2], dx=d[:, 0], dy=d[:, 1], dz=d[:, 2],
                wavelength=np.array([r.wavelength * (r.unit / scene_unit) for r in rays], dtype=float), q_re=q.real.copy(), q_im=q.imag.copy(),
                intensity=np.array([r.intensity for r in rays], dtype=float), n=np.array([r.n for r in rays], dtype=float),
                pathlength=np.array([r._pathlength for r in rays], dtype=float), id=cls,
                flags=(np.where(has_q, abi.RAY_HAS_Q, 0) | np.where([bool(r.alive) for r in rays], 0, abi.RAY_DEAD)).astype(np.int32))
    if any(r.length is not None for r in rays):
        host["length"] = np.array([np.inf if r.length is None else r.length for r in rays], dtype=float)
    return host, len(class_of)


def close(a, b, rtol=RTOL, atol=1e-9):
    return np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def assert_segments_match(got, gold, has_q_of_tree, rtol=RTOL, atol=1e-9):
    """got: dict in SEG_FIELDS layout, reference order.  gold: fixture dict."""
    n = len(gold["seg_tree"])
    assert len(got["ray"]) == n, f"segment count {len(got['ray'])} != reference {n}"
    np.testing.assert_array_equal(got["ray"], gold["seg_tree"])
    np.testing.assert_array_equal(got["surface"] == -1, gold["seg_alive"])
    for k, ax in enumerate("xyz"):
        np.testing.assert_allclose(got["o" + ax], gold["seg_origin"][:, k], rtol=rtol, atol=atol)
        np.testing.assert_allclose(got["d" + ax], gold["seg_direction"][:, k], rtol=rtol, atol=atol)
    np.testing.assert_allclose(got["length"], gold["seg_length"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(got["intensity"], gold["seg_intensity"], rtol=rtol, atol=1e-12)
    np.testing.assert_allclose(got["n"], gold["seg_n"], rtol=rtol)
    np.testing.assert_allclose(got["pathlength"], gold["seg_pathlength"], rtol=rtol, atol=atol)
    hq = has_q_of_tree[gold["seg_tree"]]
    np.testing.assert_array_equal(hq, gold["seg_has_q"])
    q = got["q_re"] + 1j * got["q_im"]
    np.testing.assert_allclose(q[hq], gold["seg_q"][hq], rtol=rtol, atol=atol)


def stored_scene(gold):
    """A CompiledScene from the tables a fixture holds (tools/make_golden.py real_example_fixture: this package's
    compiler run on the reference's own object graph in the build container; the objects do not travel, the tables do)."""
    from optable_amd.scene import CompiledScene

    return CompiledScene.from_tables(gold)


def fixture_rays_host(gold):
    """The fixture's input rays as the dict of host arrays `pack_host` builds from Ray objects."""
    n = len(gold["in_intensity"])
    o, d, q = gold["in_origin"], gold["in_direction"], gold["in_q"]
    has_q = gold["in_has_q"]
    host = dict(ox=o[:, 0], oy=o[:, 1], oz=o[:, 2], dx=d[:, 0], dy=d[:, 1], dz=d[:, 2], wavelength=gold["in_wavelength"],
                q_re=np.where(has_q, q.real, 0.0), q_im=np.where(has_q, q.imag, 0.0), intensity=gold["in_intensity"],
                n=gold["in_n"], pathlength=gold["in_pathlength"])
    host = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in host.items()}
    host["id"] = np.arange(n, dtype=np.int32)
    host["flags"] = np.where(has_q, abi.RAY_HAS_Q, 0).astype(np.int32) | np.where(gold["in_alive"], 0, abi.RAY_DEAD).astype(np.int32)
    return host


EXAMPLES = os.path.join(GOLDEN, "examples")


def example_names():
    return sorted(f[:-4] for f in os.listdir(EXAMPLES) if f.endswith(".npz"))


def example_rays_host(gold):
    """Input rays of an examples/ fixture in the ot_rays layout (class ids; wavelengths in the scene's unit as table._pack does)."""
    host = fixture_rays_host(gold)
    host["id"] = gold["in_class"].astype(np.int32)
    host["wavelength"] = host["wavelength"] * gold["in_unit_scale"]
    return host
