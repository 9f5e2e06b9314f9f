"""CPU, build container only: scenes built with the ORIGINAL optable classes compile (through
optable_amd.adapter's duck typing) to exactly the same device tables as the same scenes built with this
package's classes.  Skipped where /root/reference does not exist (e.g. the GPU box)."""
import os
import sys

import numpy as np
import pytest

import scenes
import optable_amd as oa
from optable_amd import abi

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optable")), reason="reference not present")


@pytest.fixture(scope="module")
def ref():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        import optable  # the reference package
    finally:
        sys.path.remove(REF)
    assert optable.__file__.startswith(REF)
    return optable


ADAPTABLE = sorted(scenes.SCENES)  # incl. g16: Block with a hole = the reference's closure-based Plane.subtract


@pytest.mark.parametrize("name", ADAPTABLE)
def test_reference_objects_compile_to_the_same_tables(name, ref):
    np.random.seed(12345)
    theirs = oa.compile_scene(scenes.SCENES[name](ref)["components"])
    np.random.seed(12345)
    mine = oa.compile_scene(scenes.SCENES[name](oa)["components"])
    a, b = theirs.node_table(), mine.node_table()
    assert len(a) == len(b)
    for field in a.dtype.names:
        if a[field].dtype.kind == "f":
            np.testing.assert_allclose(a[field], b[field], rtol=0, atol=1e-12, err_msg=field)
        else:
            np.testing.assert_array_equal(a[field], b[field], err_msg=field)
    assert theirs.n_materials == mine.n_materials
    for k in range(mine.n_materials):
        for field, _ in abi.OtMaterial._fields_:
            x, y = getattr(theirs.materials[k], field), getattr(mine.materials[k], field)
            np.testing.assert_allclose(np.array(x[:] if hasattr(x, "__len__") else x), np.array(y[:] if hasattr(y, "__len__") else y), atol=1e-15)
    np.testing.assert_allclose(np.ctypeslib.as_array(theirs.aux)[: theirs.n_aux], np.ctypeslib.as_array(mine.aux)[: mine.n_aux], atol=1e-12)
    assert (theirs.max_children, theirs.root_grid, len(theirs.limited)) == (mine.max_children, mine.root_grid, len(mine.limited))


def test_reference_boolean_apertures_are_recovered_from_their_closures(ref):
    """The reference builds Plane.union / subtract as closures (surfaces.py:100-136); the adapter reads the
    operands from the closure cells, MEASURES the operator and checks the recovered program on a grid."""
    from optable_amd import shapes

    hole = ref.Block([4, 0, 0], hole=ref.Circle(0.3), width=2, height=2)
    mine = oa.Block([4, 0, 0], hole=oa.Circle(0.3), width=2, height=2)
    a, b = oa.compile_scene([hole]), oa.compile_scene([mine])
    np.testing.assert_array_equal(np.ctypeslib.as_array(a.aux)[: a.n_aux], np.ctypeslib.as_array(b.aux)[: b.n_aux])
    assert a.node_table()["shape"][0] == shapes.CSG
    # nested: (rectangle - circle) | small rectangle
    nested_ref = ref.Rectangle(2, 2).subtract(ref.Circle(0.5)).union(ref.Rectangle(0.2, 3.0))
    nested_mine = oa.Rectangle(2, 2).subtract(oa.Circle(0.5)).union(oa.Rectangle(0.2, 3.0))
    from optable_amd import adapter
    assert adapter.lower_surface(nested_ref).aux == nested_mine.lower().aux


def test_unrecognisable_surface_is_rejected_loudly(ref):
    class Blob(ref.Plane):          # a user subclass with its own boundary: no device form
        def within_boundary(self, P):
            return P[1] ** 2 + 3 * P[2] ** 2 < 1
    blk = ref.Block([4, 0, 0], width=2, height=2)
    blk.surface = Blob()
    with pytest.raises(oa.SceneError):
        oa.compile_scene([blk])


def test_install_patches_and_restores(ref):
    before = ref.OpticalTable.ray_tracing
    undo = oa.install(ref)
    assert ref.OpticalTable.ray_tracing is not before and hasattr(ref.OpticalTable, "compile")
    table = ref.OpticalTable()
    table.add_components(scenes.cfg2_components(ref))
    assert table.compile().n_leaves == 3
    undo()
    assert ref.OpticalTable.ray_tracing is before and not hasattr(ref.OpticalTable, "compile")


def test_reference_ray_objects_round_trip_through_pack_and_scatter(ref):
    """install() hands the ORIGINAL Ray objects to this package's packing / unpacking code.  Without a GPU the
    trace in between cannot run here, but both ends can: pack reference rays into a (CPU) RayBatch, then rebuild
    reference Ray objects from a fabricated segment table and use them through the reference's own methods."""
    from optable_amd.table import _pack, _scatter_segments, _clone_rays

    rays = [ref.Ray([0, 0.1 * k, 0], [1, 0.01 * k, 0], wavelength=780e-7, w0=50e-4, id=7 + k) for k in range(3)]
    rays.append(ref.Ray([1, 2, 3], [0, 1, 0], alive=False, length=2.5))          # no wavelength, no q, dead, finite
    batch = _pack(rays, np.arange(4, dtype=np.int32), "cpu")
    np.testing.assert_allclose(batch.oy.numpy(), [0.0, 0.1, 0.2, 2.0])
    np.testing.assert_allclose(batch.dx.numpy()[:3], [r.direction[0] for r in rays[:3]])
    assert batch.flags.tolist() == [abi.RAY_HAS_Q] * 3 + [abi.RAY_DEAD]
    assert batch.length is not None and batch.length.tolist()[3] == 2.5 and np.isinf(batch.length.tolist()[0])
    np.testing.assert_allclose(batch.q_im.numpy()[:3], [complex(r.qo).imag for r in rays[:3]])
    segs = {"ray": np.array([0, 0, 2], dtype=np.int32), "surface": np.array([4, -1, -1], dtype=np.int32),
            "ox": np.array([0.0, 5.0, 0.0]), "oy": np.array([0.0, 0.05, 0.2]), "oz": np.zeros(3),
            "dx": np.array([1.0, -1.0, 1.0]), "dy": np.zeros(3), "dz": np.zeros(3),
            "length": np.array([5.0, np.inf, np.inf]), "intensity": np.array([1.0, 0.5, 1.0]),
            "q_re": np.array([0.0, 5.0, 0.0]), "q_im": np.array([1.0, 1.0, 1.0]),
            "n": np.array([1.0, 1.5, 1.0]), "pathlength": np.array([0.0, 5.0, 0.0])}
    per_ray = [None] * 4
    _scatter_segments(segs, rays, np.arange(4), per_ray)
    assert [len(p) for p in per_ray] == [2, 0, 1, 0]
    first, second = per_ray[0]
    assert type(first) is type(rays[0]) and first._id == rays[0]._id == 7
    assert first.length == 5.0 and first.alive is False and second.length is None and second.alive is True
    assert second.n == 1.5 and second.intensity == 0.5                          # the reference's own `n` property
    assert second.pathlength(2.0) == pytest.approx(5.0 + 2.0 * 1.5)             # and its pathlength(t)
    assert complex(second.q_at_z(1.0)) == pytest.approx(6 + 1j)
    np.testing.assert_allclose(second.direction, [-1, 0, 0])
    clones = _clone_rays(per_ray[0])
    clones[0].origin[0] = 99.0
    assert per_ray[0][0].origin[0] == 0.0                                        # the returned list is independent


def test_reference_objects_with_user_functions_get_the_same_series(ref):
    """ASphericLens(f_asphere = callable) and Material(n = callable) built with the REFERENCE's classes lower to the same
    Chebyshev records as this package's own (optable_amd/cheb.py through adapter.py)."""
    a = oa.compile_scene(scenes.g24_callables(ref)["components"])
    b = oa.compile_scene(scenes.g24_callables(oa)["components"])
    assert a.n_aux == b.n_aux > 100
    np.testing.assert_array_equal(np.ctypeslib.as_array(a.aux)[: a.n_aux], np.ctypeslib.as_array(b.aux)[: b.n_aux])
    assert [m.kind for m in a.materials[: a.n_materials]] == [m.kind for m in b.materials[: b.n_materials]]
    assert list(a.node_table()["shape"]) == list(b.node_table()["shape"]) and 10 in a.node_table()["shape"]
