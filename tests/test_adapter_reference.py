"""CPU, build container only: scenes built with the ORIGINAL optable classes compile (through
optable_amd.adapter's duck typing) to exactly the same device tables as the same scenes built with this
package's classes.  Skipped where /root/reference does not exist (e.g. the GPU box)."""
import os
import sys

import numpy as np
import pytest

import helpers
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


ADAPTABLE = [n for n in sorted(scenes.SCENES) if n != "g16_misc"]  # g16 has a Plane.subtract closure (Block with a hole)


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


def test_reference_boolean_aperture_is_rejected_loudly(ref):
    blk = ref.Block([4, 0, 0], hole=ref.Circle(0.3), width=2, height=2)
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
