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
