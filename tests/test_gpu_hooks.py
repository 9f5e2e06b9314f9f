"""GPU (MI355X): user-defined components — subclasses that override `interact_local`, the reference's subclassing hook
(optical_component.py:235-240).  The device does what it does for every scene (nearest hit among all leaves, boxes, count
gates, the children of the built-in leaves); the user's Python is called for the rays whose nearest hit is a hooked leaf
(table.py: _trace_hooked).  Checked against a fixture the reference produced from the very same user classes (g25,
tools/make_golden.py) and, for a user class that restates a built-in one, against the built-in kernel path."""
import numpy as np
import pytest

import helpers
import optable_amd as oa
from optable_amd.scene import SceneError
from test_gpu_parity import rays_to_segs

pytestmark = pytest.mark.gpu


def test_user_components_match_the_reference_fixture(capsys):
    table, sc = helpers.build("g25_user_components")
    gold = helpers.golden("g25_user_components")
    assert len(table.compile().hooks) == 4  # two gratings, the mirror subclass, the absorber
    out = table.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])
    assert len(out) == len(gold["seg_tree"])
    got = rays_to_segs(out)
    got["ray"] = gold["seg_tree"]
    np.testing.assert_array_equal(got["has_q"], gold["seg_has_q"])
    helpers.assert_segments_match(got, gold, gold["in_has_q"])
    in_ids = [r._id for r in sc["rays"]]
    assert [r._id for r in out] == [in_ids[t] for t in gold["seg_tree"]]
    scene = table.compile()
    got_counts = np.array([[c._interact_count.get(i, 0) for i in in_ids] for c in scene.limited])
    np.testing.assert_array_equal(got_counts, gold["counts"])
    # four trees ran into the cap of 150 pops (the reference printed its message four times, tools/make_golden.py's log); the
    # fifth ends in the absorber after one
    assert "(4 ray tree(s) truncated)" in capsys.readouterr().out


class PythonMirror(oa.OpticalComponent):
    """What optical_component.py:536-570 does for reflectivity 1, written by a user against the public API."""

    def __init__(self, origin, radius, **kwargs):
        super().__init__(origin, **kwargs)
        self.surface = oa.Circle(radius)

    def interact_local(self, ray):
        P, t = self.intersect_point_local(ray)
        normal = self.surface.normal(P)
        d = ray.direction - 2 * np.dot(ray.direction, normal) * normal
        return [ray.copy(origin=P, direction=d, qo=None if ray.qo is None else ray.q_at_z(t), _pathlength=ray.pathlength(float(t)))]


def _cavity(mirror_cls):
    lens = oa.Lens([3, 0, 0], focal_length=6.0, radius=1.5)
    slab = oa.GlassSlab([5, 0, 0], width=3, height=3, thickness=0.4, n1=1, n2=1.5, reflectivity=0.2).RotZ(0.1)
    if mirror_cls is oa.Mirror:
        a, b = oa.Mirror([8, 0, 0], radius=2.0).RotZ(np.pi + 0.03), oa.Mirror([-1, 0, 0], radius=2.0)
    else:
        a, b = mirror_cls([8, 0, 0], 2.0).RotZ(np.pi + 0.03), mirror_cls([-1, 0, 0], 2.0)
    table = oa.OpticalTable()
    table.add_components([lens, slab, a, b])
    rays = [oa.Ray([0, y, 0.1 * y], [1, 0.02, 0], wavelength=633e-7, w0=50e-4) for y in (-0.5, 0.0, 0.3)]
    return table, rays


def test_a_user_mirror_in_python_equals_the_builtin_mirror():
    """Same scene, the two end mirrors once as the built-in class (everything on the device: k_trace_trees) and once as a
    user class (device search, Python reflection): the same trees in the same order."""
    ref_table, rays = _cavity(oa.Mirror)
    want = rays_to_segs(ref_table.ray_tracing(rays, perfomance_limit={"max_trace_num": 40}))
    table, rays = _cavity(PythonMirror)
    assert len(table.compile().hooks) == 2
    got = rays_to_segs(table.ray_tracing(rays, perfomance_limit={"max_trace_num": 40}))
    assert len(got["ox"]) == len(want["ox"]) > 100
    for f in want:
        np.testing.assert_allclose(got[f], want[f], rtol=1e-9, atol=1e-9, err_msg=f)


def test_single_call_api_and_batch_refusal():
    m = PythonMirror([2, 0, 0], 1.0).RotZ(np.pi + 0.2)
    ray = oa.Ray([0, 0.1, 0], [1, 0, 0], wavelength=633e-7, w0=50e-4)
    t, rays = m.interact(ray)
    t0, rays0 = oa.Mirror([2, 0, 0], radius=1.0).RotZ(np.pi + 0.2).interact(ray)
    assert t == pytest.approx(t0, rel=1e-12) and len(rays) == len(rays0) == 2
    np.testing.assert_allclose(rays[1].direction, rays0[1].direction, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(rays[1].origin, rays0[1].origin, rtol=1e-12, atol=1e-15)
    assert complex(rays[1].qo) == pytest.approx(complex(rays0[1].qo), rel=1e-12)
    assert m.interact(oa.Ray([0, 5, 0], [1, 0, 0], wavelength=633e-7)) == (None, None)
    # whole-trace launches cannot call Python: the batch entry point says where such a scene is traced
    from optable_amd.batch import RayBatch

    table = oa.OpticalTable()
    table.add_components([m])
    batch = RayBatch.from_arrays(np.zeros((4, 3)), np.tile([1.0, 0, 0], (4, 1)), wavelength=633e-7, device="cuda")
    with pytest.raises(SceneError, match="ray_tracing"):
        table.trace_batch(batch, 5)


def test_install_serves_a_foreign_table_class():
    """`optable_amd.install(module)` patches the ORIGINAL package's OpticalTable: its instances are not this package's
    class, so the patched method must not rely on any method of `optable_amd.OpticalTable` being on `self`.  The reference
    does not travel to the GPU box; a stand-in module with the attributes `install` touches (a table class that only holds
    lists, the way optical_table.py:20-55 does) shows that the whole object path — compile, trace, ray trees, user hooks,
    monitors — runs for such an object."""
    import types

    class ForeignTable:
        def __init__(self):
            self.components, self.monitors, self.rays, self.unit = [], [], [], 1e-2

        def ray_tracing(self, rays, perfomance_limit=None):
            raise AssertionError("the original loop must not run")

    class ForeignMonitor(oa.Monitor):
        pass

    module = types.SimpleNamespace(OpticalTable=ForeignTable, Monitor=ForeignMonitor, Ray=oa.Ray)
    undo = oa.install(module)
    try:
        for mirror_cls in (oa.Mirror, PythonMirror):
            native, rays = _cavity(mirror_cls)
            foreign = ForeignTable()
            foreign.components = list(native.components)
            want = rays_to_segs(native.ray_tracing(rays, perfomance_limit={"max_trace_num": 40}))
            got = rays_to_segs(foreign.ray_tracing(rays, perfomance_limit={"max_trace_num": 40}))
            assert len(got["ox"]) == len(want["ox"]) > 100
            for f in want:
                np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    finally:
        undo()
    assert "compile" not in ForeignTable.__dict__
