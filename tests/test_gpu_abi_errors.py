"""Error behaviour of the C-ABI (include/optable_hip.h): every entry point answers a malformed call
with a negative ot_status and a message from ot_last_error(); nothing reaches a kernel.  The scene
tables are indices into each other (node.end, aux offsets, grid records, polygon / CSG records), so
ot_scene_upload checks every one of them before a ray can follow them on the device."""
import ctypes as C

import pytest

import scenes
from optable_amd import abi
from optable_amd import shapes as sh

pytestmark = pytest.mark.gpu

ERR_INVALID, ERR_UNSUPPORTED, ERR_NOSCENE = -1, -3, -5


@pytest.fixture()
def ctx():
    lib = abi.load()
    c = C.c_void_p()
    assert lib.ot_ctx_create(0, None, C.byref(c)) == 0
    yield lib, c
    assert lib.ot_ctx_destroy(c) == 0


def _compiled(builder):
    import optable_amd as oa

    t = oa.OpticalTable()
    t.add_components(builder(oa)["components"])
    return t.compile()


def _expect(lib, status, code, needle):
    msg = lib.ot_last_error().decode()
    assert status == code, (status, msg)
    assert needle in msg, msg


def test_upload_rejects_broken_tables(ctx):
    lib, c = ctx
    sc = _compiled(scenes.g12_dove)          # polygons in and out of the x = 0 plane
    assert lib.ot_scene_upload(c, C.byref(sc.desc())) == 0
    nodes = sc.node_table()
    poly = [i for i in range(sc.n_nodes) if nodes[i]["kind"] == abi.NODE_LEAF and nodes[i]["shape"] in (sh.POLYGON2D, sh.POLYGON3D)]
    assert poly
    i = poly[0]
    off = int(nodes[i]["aux"])
    # vertex count that runs past the aux table
    keep = sc.aux[off]
    sc.aux[off] = 1e6
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "polygon record")
    sc.aux[off] = keep
    # aux offset outside the table
    sc.nodes[i].aux = sc.n_aux + 5
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "polygon record")
    sc.nodes[i].aux = off
    # skip pointer that leaves the node list / points backwards
    group = next(k for k in range(sc.n_nodes) if nodes[k]["kind"] == abi.NODE_GROUP)
    end = sc.nodes[group].end
    sc.nodes[group].end = sc.n_nodes + 1
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "node.end")
    sc.nodes[group].end = group
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "node.end")
    sc.nodes[group].end = end
    # unknown kinds
    shape = sc.nodes[i].shape
    sc.nodes[i].shape = 99
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_UNSUPPORTED, "shape")
    sc.nodes[i].shape = shape
    assert lib.ot_scene_upload(c, C.byref(sc.desc())) == 0   # restored table is accepted again
    d = sc.desc()
    d.max_children = 3
    _expect(lib, lib.ot_scene_upload(c, C.byref(d)), ERR_INVALID, "max_children")
    d = sc.desc()
    d.unit = 0.0
    _expect(lib, lib.ot_scene_upload(c, C.byref(d)), ERR_INVALID, "unit")
    d = sc.desc()
    d.aux = None
    _expect(lib, lib.ot_scene_upload(c, C.byref(d)), ERR_INVALID, "aux is NULL")
    _expect(lib, lib.ot_scene_upload(c, None), ERR_INVALID, "bad scene sizes")


def test_upload_rejects_broken_materials_and_csg(ctx):
    lib, c = ctx
    import optable_amd as oa

    t = oa.OpticalTable()
    t.add_components([oa.Block([3, 0, 0], hole=oa.Circle(0.3), width=2, height=2),
                      oa.GlassSlab([6, 0, 0], width=2, height=2, thickness=0.5, n1=oa.Vacuum(), n2=oa.Glass_NBK7())])
    sc = t.compile()
    assert lib.ot_scene_upload(c, C.byref(sc.desc())) == 0
    nodes = sc.node_table()
    csg = next(i for i in range(sc.n_nodes) if nodes[i]["kind"] == abi.NODE_LEAF and nodes[i]["shape"] == sh.CSG)
    off = int(nodes[csg]["aux"])
    prog = [sc.aux[off + k] for k in range(sc.n_aux - off)]
    ntok = int(prog[0])
    assert ntok == 3                       # rectangle, circle, subtract
    sc.aux[off] = 2                        # drop the operator: two values left on the stack
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "CSG")
    sc.aux[off] = 500                      # tokens run off the table
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "CSG")
    sc.aux[off] = ntok
    first_kind = sc.aux[off + 1]
    sc.aux[off + 1] = 101                  # operator with an empty stack
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "CSG operator")
    sc.aux[off + 1] = first_kind
    assert lib.ot_scene_upload(c, C.byref(sc.desc())) == 0
    refr = next(i for i in range(sc.n_nodes) if nodes[i]["kind"] == abi.NODE_LEAF and nodes[i]["interaction"] == 1)
    mat2 = sc.nodes[refr].mat2
    sc.nodes[refr].mat2 = sc.n_materials
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "material index")
    sc.nodes[refr].mat2 = mat2
    sc.materials[0].kind = 7
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_UNSUPPORTED, "material kind")


def test_upload_rejects_broken_grids(ctx):
    lib, c = ctx
    sc = _compiled(lambda oa: dict(components=scenes.cfg5_components(oa)))   # MMA: a gridded group
    assert lib.ot_scene_upload(c, C.byref(sc.desc())) == 0
    nodes = sc.node_table()
    grp = next(i for i in range(sc.n_nodes) if nodes[i]["kind"] == abi.NODE_GROUP and nodes[i]["flags"] & abi.NODE_GRID)
    off = int(nodes[grp]["aux"])
    g0, g1 = int(sc.aux[off + 2]), int(sc.aux[off + 3])
    items_at = off + 9 + g0 * g1 + 1
    keep = sc.aux[items_at]
    sc.aux[items_at] = float(grp)          # an item that is not a leaf child of the group
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "grid item")
    sc.aux[items_at] = keep
    keep = sc.aux[off + 9 + 1]
    sc.aux[off + 9 + 1] = -3.0             # cell starts must not decrease
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "grid")
    sc.aux[off + 9 + 1] = keep
    sc.aux[off + 2] = 1e9                  # absurd cell count
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc.desc())), ERR_INVALID, "grid")
    sc.aux[off + 2] = g0
    sc3 = _compiled(lambda oa: dict(components=scenes.cfg3_components(oa)))  # 32 top-level components: root grid
    assert sc3.root_grid >= 0
    d = sc3.desc()
    d.root_grid = sc3.n_aux
    _expect(lib, lib.ot_scene_upload(c, C.byref(d)), ERR_INVALID, "root grid")
    roff = sc3.root_grid
    rg0, rg1 = int(sc3.aux[roff + 2]), int(sc3.aux[roff + 3])
    sc3.aux[roff + 11 + rg0 * rg1 + 1] = 1e6   # first item: no such node
    _expect(lib, lib.ot_scene_upload(c, C.byref(sc3.desc())), ERR_INVALID, "root grid item")


def test_trace_argument_checks(ctx):
    import torch
    from optable_amd.batch import RayBatch, SegmentBatch

    lib, c = ctx
    o, d = scenes.cfg2_rays(64, 0)
    rays = RayBatch.from_arrays(o, d, wavelength=scenes.WL)
    segs = SegmentBatch(64 * 5)
    count = torch.zeros(64, dtype=torch.int32, device="cuda")
    rs, ss = rays.c_struct(), segs.c_struct()
    call = lambda r, n, K, s, cnt, tab=None, ncls=0: lib.ot_trace_f64(c, r, n, K, s, cnt, tab, ncls)
    _expect(lib, call(C.byref(rs), 64, 5, C.byref(ss), count.data_ptr()), ERR_NOSCENE, "ot_scene_upload")
    sc = _compiled(lambda oa: dict(components=scenes.cfg2_components(oa)))
    assert lib.ot_scene_upload(c, C.byref(sc.desc())) == 0
    _expect(lib, call(None, 64, 5, C.byref(ss), count.data_ptr()), ERR_INVALID, "rays is NULL")
    _expect(lib, call(C.byref(rs), 64, 5, None, count.data_ptr()), ERR_INVALID, "segments is NULL")
    _expect(lib, call(C.byref(rs), 64, 0, C.byref(ss), count.data_ptr()), ERR_INVALID, "max_segments")
    _expect(lib, call(C.byref(rs), -1, 5, C.byref(ss), count.data_ptr()), ERR_INVALID, "bad n")
    _expect(lib, call(C.byref(rs), 1 << 31, 5, C.byref(ss), count.data_ptr()), ERR_INVALID, "2^31")
    _expect(lib, call(C.byref(rs), 64, 5, C.byref(ss), None), ERR_INVALID, "seg_count")
    broken = rays.c_struct()
    broken.q_im = None
    _expect(lib, call(C.byref(broken), 64, 5, C.byref(ss), count.data_ptr()), ERR_INVALID, "NULL field")
    # a scene with a count-limited surface needs the table
    lim = _compiled(scenes.g13_count_shadow)
    assert lib.ot_scene_upload(c, C.byref(lim.desc())) == 0
    _expect(lib, call(C.byref(rs), 64, 5, C.byref(ss), count.data_ptr()), ERR_INVALID, "counts table")
    # ids outside the table are not counted and index nothing (header: ot_trace_f64)
    table = torch.zeros((len(lim.limited), 4), dtype=torch.int32, device="cuda")
    rays.id.fill_(1 << 30)
    assert call(C.byref(rs), 64, 5, C.byref(ss), count.data_ptr(), table.data_ptr(), 4) == 0
    assert lib.ot_ctx_synchronize(c) == 0
    assert int(table.abs().sum()) == 0
    assert lib.ot_ctx_synchronize(None) == ERR_INVALID
