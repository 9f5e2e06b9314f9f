"""Scene compiler: object graph -> plain tables the device can stage into LDS.

Depth-first flattening of `OpticalTable.components`.  Order is the contract: the reference
tests top-level components in list order and keeps the first strictly smaller t
(optical_table.py:119-123); a ComponentGroup tests its own AABB, then every child's AABB, then
the children that passed, and returns the first minimum (component_group.py:93-122).  Emitting
`[group, child, child, ...]` in that order, with `end` = index past the last descendant and
`CHECK_AABB` on groups and their children, lets one forward pass with a strict `<` reproduce
the same winner, the same pruning, and the same interact-count side effects.

Anything without a device form (custom Material callables, arbitrary f_asphere closures,
unbounded planes) raises here; there is no host fallback.
"""
import ctypes as C

import numpy as np

from . import abi
from .components import MIRROR, REFRACT, LENS, ROC_INF
from .materials import Material


class SceneError(NotImplementedError):
    pass


class CompiledScene:
    """Owner of the ctypes tables handed to `ot_scene_upload`."""

    def __init__(self, nodes, materials, aux, leaves, limited, max_children, unit, root_grid=-1, always_branches=False,
                 wavelength_range=None):
        self.wavelength_range = wavelength_range  # metres; None when no material is a fitted series
        self.root_grid = root_grid
        self.always_branches = always_branches
        self.nodes = (abi.OtNode * max(len(nodes), 1))(*nodes)
        self.n_nodes = len(nodes)
        self.materials = (abi.OtMaterial * max(len(materials), 1))(*materials)
        self.n_materials = len(materials)
        self.aux = (C.c_double * max(len(aux), 1))(*aux)
        self.n_aux = len(aux)
        self.leaves = leaves        # leaf_id -> component
        self.hooks = {}             # leaf_id -> component whose interact_local is the user's Python (adapter.host_hook)
        self.limited = limited      # count_slot -> component
        self.max_children = max_children
        self.unit = unit

    @property
    def n_leaves(self):
        return len(self.leaves)

    def desc(self):
        d = abi.OtSceneDesc()
        d.nodes, d.n_nodes = self.nodes, self.n_nodes
        d.materials, d.n_materials = self.materials, self.n_materials
        d.aux, d.n_aux = self.aux, self.n_aux
        d.n_count_slots = len(self.limited)
        d.max_children = self.max_children
        d.unit = self.unit
        d.root_grid = self.root_grid
        return d

    # -- a compiled scene as plain arrays (to keep one, or to carry a scene to a machine its Python objects cannot go to) -----
    def to_tables(self):
        """dict of numpy arrays: the node / material records as bytes, the aux table, and the scalars `from_tables` needs."""
        return {"nodes": np.frombuffer(bytes(self.nodes), dtype=np.uint8).copy(),
                "materials": np.frombuffer(bytes(self.materials), dtype=np.uint8).copy(),
                "aux": np.array(self.aux[: self.n_aux], dtype=float),
                "meta": np.array([self.n_nodes, self.n_materials, len(self.limited), self.max_children, self.root_grid,
                                  int(self.always_branches), self.n_leaves], dtype=np.int64),
                "unit": np.array([self.unit]),
                "limited_max": np.array([int(c.max_interact_count) for c in self.limited], dtype=np.int64)}

    @classmethod
    def from_tables(cls, tables):
        """The scene `to_tables` described.  Its `leaves` / `limited` entries are placeholders (there are no component objects
        to write interact counts back to): for `Engine.trace*` / the oracle, which only read the tables."""
        import types

        n_nodes, n_mat, _, max_children, root_grid, always, n_leaves = (int(x) for x in tables["meta"])
        sc = object.__new__(cls)
        sc.nodes = (abi.OtNode * max(n_nodes, 1)).from_buffer_copy(np.ascontiguousarray(tables["nodes"]).tobytes())
        sc.n_nodes = n_nodes
        sc.materials = (abi.OtMaterial * max(n_mat, 1)).from_buffer_copy(np.ascontiguousarray(tables["materials"]).tobytes())
        sc.n_materials = n_mat
        aux = np.ascontiguousarray(tables["aux"], dtype=np.float64)
        sc.aux = (C.c_double * max(len(aux), 1))(*aux.tolist())
        sc.n_aux = len(aux)
        sc.leaves = [None] * n_leaves
        sc.limited = [types.SimpleNamespace(max_interact_count=int(m), _interact_count={}) for m in tables["limited_max"]]
        sc.hooks = {}
        sc.max_children, sc.unit, sc.root_grid = max_children, float(tables["unit"][0]), root_grid
        sc.always_branches, sc.wavelength_range = bool(always), None
        return sc

    def node_table(self):
        """Numpy view of the node records (for tests)."""
        return np.ctypeslib.as_array(self.nodes)[: self.n_nodes] if self.n_nodes else np.zeros(0)


class _Builder:
    def __init__(self, accelerate=True):
        self.accelerate = accelerate
        self.nodes, self.materials, self.aux = [], [], []
        self.mat_index = {}
        self.leaves, self.limited, self.hooks = [], [], {}
        self.wavelength_range = None  # metres: where every fitted dispersion series of the scene is valid
        self.leaf_pose = []   # per leaf: (transform_matrix, origin, local box) for the fresh lab boxes (_trust_boxes)
        self.groups = []      # node indices of the groups, in node order
        self.max_children = 0
        self.always_branches = False

    # -- materials -----------------------------------------------------------------------
    def material(self, mat):
        if not hasattr(mat, "device_spec"):
            mat = Material("Constant", float(mat))
        spec = mat.device_spec()
        if spec is None:
            from .materials import callable_spec

            raise SceneError(f"Material {mat.name!r} is defined by a Python callable that has no device form: "
                             f"{callable_spec.why or 'not a function of the wavelength the compiler can sample'}")
        if spec[0] == "const":  # (the common case by far: a number, no arrays to flatten)
            key = ("const", float(spec[1]))
        else:
            key = (spec[0],) + tuple(np.ravel(np.concatenate([np.ravel(x) for x in spec[1:]])).tolist())
        if key not in self.mat_index:
            rec = abi.OtMaterial()
            if spec[0] == "const":
                rec.kind, rec.n = abi.MAT_CONST, spec[1]
            elif spec[0] == "cheb":  # n holds the aux offset of the series record [N, lo, hi, c...]
                from . import cheb

                rec.kind, rec.n = abi.MAT_CHEB, float(len(self.aux))
                self.aux.extend(cheb.record(spec[3], spec[1], spec[2]))
                lo, hi = self.wavelength_range or (spec[1], spec[2])
                self.wavelength_range = (max(lo, spec[1]), min(hi, spec[2]))
            else:
                rec.kind, rec.n = abi.MAT_SELLMEIER, 0.0
                rec.B[:], rec.C[:] = spec[1], spec[2]
            self.mat_index[key] = len(self.materials)
            self.materials.append(rec)
        return self.mat_index[key]

    # -- nodes -----------------------------------------------------------------------------
    def add(self, comp, in_group):
        if hasattr(comp, "components"):
            self.add_group(comp)
        else:
            self.add_leaf(comp, in_group)

    def add_group(self, grp):
        node = abi.OtNode()
        node.kind, node.flags = abi.NODE_GROUP, abi.NODE_CHECK_AABB
        node.aabb[:] = [float(x) for x in grp.bbox]
        node.leaf_id = node.aux = node.count_slot = -1
        node.max_interact_count = -1
        slot = len(self.nodes)
        self.nodes.append(node)
        for child in grp.components:
            self.add(child, in_group=True)
        node.end = len(self.nodes)
        self.groups.append(slot)
        return slot

    GRID_MIN_CHILDREN = 24

    def _maybe_grid(self, node, slot):
        """Large flat groups (MLA / MMA / DMD ...): a 2-D grid over the children's lab AABBs so the
        kernel visits only the children near the ray instead of all of them.  Pure acceleration:
        every child found through the grid still passes its own AABB test, and a child the grid
        misses could not have passed it.  Not built when a child is itself a group or count-limited
        (its gate must see every geometric hit).

        Superset argument.  A child is listed in the cells its box overlaps after SHRINKING the box by delta =
        margin / 4 on every side (a box thinner than that: the cell of its centre); the kernel looks up the cells
        the ray's footprint inside the group box overlaps after WIDENING it by 2 * margin (fp32: by 64 ulp of the
        group's largest coordinate, set at upload).  A child whose slab test passes has a box that the footprint
        reaches to within the test's 1e-12 / 1e-8 tolerances, so the widened footprint and the shrunken box share
        a point, and the cell of that point is both listed and looked up.  Shrinking instead of widening the boxes
        matters for lattices: their children end exactly on cell borders, and a widened box would be listed in all
        eight neighbouring cells as well.

        Lattices (equal boxes on a regular raster: MMA caps, MLA lenslets, DMD mirrors, component_group.py:228-304,
        367) get the raster itself as their grid — one member per cell, found in O(1) — and the device folds their
        members into one record plus a pose per member (optable_hip.hip find_runs).  Other groups get cells about
        half the size of a typical child, at most 64 x 64."""
        kids = list(range(slot + 1, node.end))
        if not self.accelerate or len(kids) < self.GRID_MIN_CHILDREN:
            return
        if not (node.flags & abi.NODE_BOX_TRUSTED) or not all(self.nodes[k].flags & abi.NODE_BOX_TRUSTED for k in kids):
            return  # a cached box that no longer holds its geometry (moved after a trace): plain AABB tests, no shortcuts
        if any(self.nodes[k].kind != abi.NODE_LEAF or self.nodes[k].max_interact_count >= 0 for k in kids):
            return
        gbox = np.array(node.aabb[:], dtype=float).reshape(3, 2)
        extent = gbox[:, 1] - gbox[:, 0]
        a0, a1 = sorted(np.argsort(extent)[-2:].tolist())
        if extent[a0] <= 0 or extent[a1] <= 0:
            return
        boxes = np.array([self.nodes[k].aabb[:] for k in kids], dtype=float).reshape(-1, 3, 2)
        if not np.all(np.isfinite(boxes)) or not np.all(np.isfinite(gbox)):
            return
        margin = 1e-7 + 1e-9 * float(extent.max())
        raster = _lattice_raster(boxes[:, [a0, a1], :], gbox[[a0, a1], :])
        if raster is not None:
            org, size, dims = raster
        else:
            typical = np.median(boxes[:, [a0, a1], 1] - boxes[:, [a0, a1], 0], axis=0)
            dims = [int(min(64, max(1, np.floor(extent[a] / max(typical[j] / 2, extent[a] / 64))))) for j, a in enumerate((a0, a1))]
            size = [extent[a0] / dims[0], extent[a1] / dims[1]]
            org = [gbox[a0, 0], gbox[a1, 0]]
        # cover the group box with a margin to spare on the high side (the low side starts at or before the box)
        for j, a in enumerate((a0, a1)):
            while org[j] + dims[j] * size[j] < gbox[a, 1] + 4 * margin:
                dims[j] += 1
        inv = [1.0 / size[0], 1.0 / size[1]]
        delta = margin / 4
        cells = [[] for _ in range(dims[0] * dims[1])]

        def span(lo, hi, o, i, n):
            lo, hi = lo + delta, hi - delta
            mid = 0.5 * (lo + hi)
            thin = hi < lo
            lo, hi = np.where(thin, mid, lo), np.where(thin, mid, hi)
            return (np.clip(np.floor((lo - o) * i), 0, n - 1).astype(int).tolist(),
                    np.clip(np.floor((hi - o) * i), 0, n - 1).astype(int).tolist())

        lo0s, hi0s = span(boxes[:, a0, 0], boxes[:, a0, 1], org[0], inv[0], dims[0])
        lo1s, hi1s = span(boxes[:, a1, 0], boxes[:, a1, 1], org[1], inv[1], dims[1])
        for k, lo0, hi0, lo1, hi1 in zip(kids, lo0s, hi0s, lo1s, hi1s):
            for c1 in range(lo1, hi1 + 1):
                for c0 in range(lo0, hi0 + 1):
                    cells[c1 * dims[0] + c0].append(k)
        starts, items = [0], []
        for lst in cells:
            items.extend(lst)
            starts.append(len(items))
        if len(items) > 32 * len(kids) + 64:   # pathological overlap: the grid would not help
            return
        node.aux = len(self.aux)
        node.flags |= abi.NODE_GRID
        self.aux.extend([float(a0), float(a1), float(dims[0]), float(dims[1]), org[0], org[1], inv[0], inv[1], 2 * margin])
        self.aux.extend(float(x) for x in starts)
        self.aux.extend(float(x) for x in items)

    def add_leaf(self, comp, in_group):
        from . import adapter  # objects of the reference package are recognised by duck typing there

        if not hasattr(comp, "surface") or not hasattr(comp, "transform_matrix"):
            raise SceneError(f"{type(comp).__name__} is not an optical component")
        surf = comp.surface
        try:
            low = adapter.lower_surface(surf)
            inter = adapter.lower_interaction(comp)
        except NotImplementedError as exc:
            raise SceneError(f"{type(comp).__name__}: {exc}") from exc
        node = abi.OtNode()
        node.kind, node.end = abi.NODE_LEAF, len(self.nodes) + 1
        node.flags = abi.NODE_CHECK_AABB if in_group else 0
        node.M[:] = np.asarray(comp.transform_matrix, dtype=float).ravel().tolist()
        node.origin[:] = np.asarray(comp.origin, dtype=float).tolist()
        if in_group or not low.planar:
            node.aabb[:] = [float(x) for x in comp.bbox]
        if not low.planar:
            node.lbox[:] = [float(x) for x in surf.get_bbox_local()]
        node.shape = low.kind
        for k, v in enumerate(low.params):
            node.p[k] = v
        if low.aux is not None:
            node.aux = len(self.aux)
            self.aux.extend(low.aux)
        else:
            node.aux = -1
        node.interaction = kind = inter["kind"]
        node.reflectivity = float(inter.get("reflectivity", 0.0))
        node.transmission = float(inter.get("transmission", 0.0))
        node.focal_length = float(inter.get("focal_length", 0.0))
        node.roc_kind, node.roc = inter.get("roc_kind", ROC_INF), float(inter.get("roc", np.inf))
        node.mat1 = self.material(inter["mat1"]) if kind == REFRACT else -1
        node.mat2 = self.material(inter["mat2"]) if kind == REFRACT else -1
        if comp.max_interact_count is None:
            node.max_interact_count, node.count_slot = -1, -1
        else:
            node.max_interact_count, node.count_slot = int(comp.max_interact_count), len(self.limited)
            self.limited.append(comp)
        node.leaf_id = len(self.leaves)
        if inter.get("host_hook"):
            self.hooks[node.leaf_id] = comp
        self.leaves.append(comp)
        self.leaf_pose.append((np.asarray(comp.transform_matrix, dtype=float), np.asarray(comp.origin, dtype=float),
                               np.asarray(surf.get_bbox_local(), dtype=float)))
        self.nodes.append(node)
        self.max_children = max(self.max_children, _fanout(kind, node.reflectivity, node.transmission))
        if kind in (MIRROR, REFRACT) and node.reflectivity > 0 and node.transmission > 0:
            self.always_branches = True  # every ordinary hit on this leaf emits two rays


LATTICE_MAX_CELLS = 128  # per axis


def _lattice_raster(boxes, gbox):
    """(origin, cell size, cells per axis) of the raster on which the children of a lattice group lie, or None.
    `boxes` [n, 2, 2] and `gbox` [2, 2] are the children's and the group's extents along the two grid axes.  A
    raster exists when at least 90 % of the children have the same extents (the others — an MMA's back plate —
    are simply listed in every cell they cover) and those members start on multiples of their size; the raster is
    continued in whole cells to the group's low edges."""
    width = boxes[:, :, 1] - boxes[:, :, 0]
    med = np.median(width, axis=0)
    if not np.all(med > 0):
        return None
    member = np.all(np.abs(width - med) <= 1e-9 * med, axis=1)
    if member.sum() < 0.9 * len(boxes):
        return None
    lo = boxes[member][:, :, 0]
    first = lo.min(axis=0)
    steps = (lo - first) / med
    if np.any(np.abs(steps - np.round(steps)) > 1e-6):
        return None
    below = np.maximum(np.ceil((first - gbox[:, 0]) / med - 1e-9), 0)
    org = first - below * med
    dims = np.maximum(np.ceil((np.maximum(gbox[:, 1], boxes[:, :, 1].max(axis=0)) - org) / med - 1e-9), 1)
    if np.any(dims > LATTICE_MAX_CELLS):
        return None
    return [float(org[0]), float(org[1])], [float(med[0]), float(med[1])], [int(dims[0]), int(dims[1])]


_CORNERS = np.array([[i, 2 + j, 4 + k] for k in (0, 1) for j in (0, 1) for i in (0, 1)])  # picks of (xmin,xmax,ymin,ymax,zmin,zmax)


def _trust_boxes(b):
    """Which cached boxes still hold their geometry.  A component caches its lab AABB on first use and never updates it
    (optical_component.py:62-67, component_group.py:29-38): after a move the box is stale, and the reference keeps using
    it as a pure pass / fail gate — so do the kernels.  But every SHORTCUT built on a box (skipping a node whose box
    starts behind the best hit, listing a component in grid cells by its box, stopping a grid walk early) assumes that
    hits lie inside the box.  A node gets NODE_BOX_TRUSTED when its cached box contains the box computed afresh from
    the current pose (leaves: the 8 corners of the local box; groups: all leaves below); the device prunes by trusted
    boxes only, and grids are built only over trusted nodes.  Returns the fresh leaf boxes [n_leaves, 6] (the
    top-level grid bins top-level leaves, which have no AABB gate at all, by these) and whether every box is trusted."""
    if not b.leaves:
        return np.zeros((0, 6)), True
    M = np.stack([p[0] for p in b.leaf_pose])
    org = np.stack([p[1] for p in b.leaf_pose])
    lb = np.stack([p[2] for p in b.leaf_pose])
    lab = np.einsum("nij,ncj->nci", M, lb[:, _CORNERS]) + org[:, None, :]          # [n, 8, 3]
    fresh = np.stack([lab[:, :, 0].min(1), lab[:, :, 0].max(1), lab[:, :, 1].min(1), lab[:, :, 1].max(1),
                      lab[:, :, 2].min(1), lab[:, :, 2].max(1)], axis=1)
    nodes = b.nodes
    leaf_nodes = [i for i, nd in enumerate(nodes) if nd.kind == abi.NODE_LEAF]
    tol = 1e-9 * (1.0 + np.abs(fresh).max(axis=1))
    # all leaves at once (a lattice scene has thousands): cached box against fresh box, flags set in one sweep
    lid = np.array([nodes[i].leaf_id for i in leaf_nodes], dtype=np.int64)
    gated = np.array([bool(nodes[i].flags & abi.NODE_CHECK_AABB) for i in leaf_nodes])
    cached = np.array([nodes[i].aabb[:] for i in leaf_nodes], dtype=float).reshape(-1, 6)
    f_leaf = fresh[lid]
    t_leaf = tol[lid][:, None]
    no_box = ~gated & ~cached.any(axis=1)  # no cached box in play: the fresh one is filled in by compile_scene
    holds = (cached[:, 0::2] <= f_leaf[:, 0::2] + t_leaf).all(axis=1) & (cached[:, 1::2] >= f_leaf[:, 1::2] - t_leaf).all(axis=1)
    trusted = no_box | holds
    everything = bool(trusted.all())
    for i, ok in zip(leaf_nodes, trusted.tolist()):
        if ok:
            nodes[i].flags |= abi.NODE_BOX_TRUSTED
    node_fresh = np.full((len(nodes), 6), np.nan)
    node_fresh[leaf_nodes] = f_leaf
    is_leaf = np.zeros(len(nodes), dtype=bool)
    is_leaf[leaf_nodes] = True
    for g in reversed(b.groups):  # a group's fresh box is the union of the leaves that lie below it
        nd = nodes[g]
        below = node_fresh[g + 1:nd.end][is_leaf[g + 1:nd.end]]
        if not len(below):
            nd.flags |= abi.NODE_BOX_TRUSTED
            continue
        f = np.array([below[:, 0].min(), below[:, 1].max(), below[:, 2].min(), below[:, 3].max(), below[:, 4].min(), below[:, 5].max()])
        c = np.array(nd.aabb[:])
        t = 1e-9 * (1.0 + np.abs(f).max())
        ok = bool(np.all(c[0::2] <= f[0::2] + t) and np.all(c[1::2] >= f[1::2] - t))
        if ok:
            nd.flags |= abi.NODE_BOX_TRUSTED
        everything &= ok
    return fresh, everything


def _fanout(kind, refl, trans):
    """Most children one hit can emit (optical_component.py:546-570, 678-715, 944-948)."""
    if kind == MIRROR:
        return int(refl > 0) + int(trans > 0)
    if kind == REFRACT:
        return 1 + int(refl > 0)  # transmitted-or-TIR, plus the partial reflection
    return 1 if kind == LENS else 0


def _flattenable_leaves(b, top):
    """Leaf node indices below `top` if the subtree can be replaced by them in the top-level grid, else None."""
    nd = b.nodes[top]
    if nd.kind == abi.NODE_LEAF:
        return [top]
    out = []
    j = top
    end = nd.end
    # depth-first list: the enclosing groups of node j are those opened before it whose `end` lies beyond it
    open_groups = []
    while j < end:
        node = b.nodes[j]
        while open_groups and b.nodes[open_groups[-1]].end <= j:
            open_groups.pop()
        box = np.array(node.aabb[:], dtype=float).reshape(3, 2)
        if not np.all(np.isfinite(box)):
            return None
        for g in open_groups:
            gbox = np.array(b.nodes[g].aabb[:], dtype=float).reshape(3, 2)
            if np.any(box[:, 0] < gbox[:, 0]) or np.any(box[:, 1] > gbox[:, 1]):
                return None  # a stale cached bbox: the group's own test is not implied, keep the subtree
        if node.kind == abi.NODE_GROUP:
            if node.flags & abi.NODE_GRID:
                return None
            if not (node.flags & abi.NODE_CHECK_AABB) and j != top:
                return None
            open_groups.append(j)
        else:
            if not (node.flags & abi.NODE_CHECK_AABB):
                return None  # every child of a group is AABB-tested (component_group.py:104-107)
            out.append(j)
        j += 1
    return out


ROOT_GRID_MIN_TOP = 12
ROOT_GRID_DIMS = None   # (g0, g1): the top-level grid's cells per axis, overriding the rule below (Engine.tune_root_grid, experiments)
ROOT_GRID_ASPECT = 1.0  # (experiment knob: > 1 = more, narrower cells along the first grid axis)
ROOT_GRID_CELLS_PER_COMPONENT = 1.0


def _root_grid(b, tops):
    """2-D grid over the top-level components (binned by lab AABB) for scenes with many of them.
    The kernel walks the cells a ray crosses in order and stops once its best hit lies inside the
    part of the ray already covered, instead of testing every component on every segment
    (optical_table.py:119-123 does the latter).  Acceleration only: a component found through the
    grid gets exactly the tests the linear pass applies, ties go to the lower index.  Skipped when
    a leaf is count-limited (its gate must see every geometric hit) or a box is not finite."""
    if len(tops) < ROOT_GRID_MIN_TOP or b.limited:
        return -1
    boxes = np.array([b.nodes[i].aabb[:] for i in tops], dtype=float).reshape(-1, 3, 2)
    if not np.all(np.isfinite(boxes)):
        return -1
    lo, hi = boxes[:, :, 0].min(axis=0), boxes[:, :, 1].max(axis=0)
    extent = hi - lo
    a0, a1 = sorted(np.argsort(extent)[-2:].tolist())
    if extent[a0] <= 0 or extent[a1] <= 0:
        return -1
    margin = 1e-7 + 1e-9 * float(extent.max())
    pad = 4 * margin
    org = [lo[a0] - pad, lo[a1] - pad]
    span = [extent[a0] + 2 * pad, extent[a1] + 2 * pad]
    # about one cell per component, shaped like the scene, at most 64 x 64.  Measured on cfg 3 (tools/
    # bench_configs.py): 1 cell per component is 1.45x faster than 4 — every extra cell a ray steps through
    # is a divergent DDA iteration for its whole wave, which costs more than the few extra candidate tests.
    cells_target = ROOT_GRID_CELLS_PER_COMPONENT * len(tops)
    g0 = int(np.clip(round(np.sqrt(cells_target * span[0] / span[1]) * ROOT_GRID_ASPECT), 1, 64))
    g1 = int(np.clip(round(np.sqrt(cells_target * span[1] / span[0]) / ROOT_GRID_ASPECT), 1, 64))
    if ROOT_GRID_DIMS is not None:
        g0, g1 = (int(np.clip(g, 1, 64)) for g in ROOT_GRID_DIMS)
    size = [span[0] / g0, span[1] / g1]
    inv = [1.0 / size[0], 1.0 / size[1]]
    cells = [[] for _ in range(g0 * g1)]
    # What goes into the cells: the LEAVES of a group instead of the group when every box on the way down
    # contains the boxes below it (always true unless a cached bbox went stale, optical_component.py:62-67) and
    # no group below has a grid of its own.  A ray that passes a leaf's AABB test then passes its ancestors' too
    # (bigger box: wider slab interval, laxer parallel-axis rule), so testing the leaf alone is the same test —
    # and all lanes of a wave run one flat loop over leaves instead of nested walks of different depth.
    entries = []
    for top in tops:
        leaves = _flattenable_leaves(b, top)
        entries.extend(leaves if leaves is not None else [top])
    entry_boxes = np.array([b.nodes[i].aabb[:] for i in entries], dtype=float).reshape(-1, 3, 2)
    if not np.all(np.isfinite(entry_boxes)):
        entries, entry_boxes = list(tops), boxes
    for idx, bx in zip(entries, entry_boxes):
        lo0 = int(np.clip(np.floor((bx[a0, 0] - margin - org[0]) * inv[0]), 0, g0 - 1))
        hi0 = int(np.clip(np.floor((bx[a0, 1] + margin - org[0]) * inv[0]), 0, g0 - 1))
        lo1 = int(np.clip(np.floor((bx[a1, 0] - margin - org[1]) * inv[1]), 0, g1 - 1))
        hi1 = int(np.clip(np.floor((bx[a1, 1] + margin - org[1]) * inv[1]), 0, g1 - 1))
        for c1 in range(lo1, hi1 + 1):
            for c0 in range(lo0, hi0 + 1):
                cells[c1 * g0 + c0].append(idx)
    starts, items = [0], []
    for lst in cells:
        items.extend(lst)  # ascending node index inside a cell
        starts.append(len(items))
    offset = len(b.aux)
    b.aux.extend([float(a0), float(a1), float(g0), float(g1), org[0], org[1], inv[0], inv[1], margin, size[0], size[1]])
    b.aux.extend(float(x) for x in starts)
    b.aux.extend(float(x) for x in items)
    return offset


def compile_scene(components, unit=1e-2, accelerate=True) -> CompiledScene:
    b = _Builder(accelerate)
    tops = []
    for comp in components:
        tops.append(len(b.nodes))
        b.add(comp, in_group=False)
    fresh, all_trusted = _trust_boxes(b)
    for i in tops:  # top-level leaves carry no AABB test in the reference; the grid still needs their boxes: the CURRENT ones
        nd = b.nodes[i]
        if nd.kind == abi.NODE_LEAF and not any(nd.aabb[:]):
            nd.aabb[:] = [float(x) for x in fresh[nd.leaf_id]]
    for g in b.groups:
        b._maybe_grid(b.nodes[g], g)
    root = _root_grid(b, tops) if accelerate and all_trusted else -1
    scene = CompiledScene(b.nodes, b.materials, b.aux, b.leaves, b.limited, b.max_children, unit, root, b.always_branches,
                          b.wavelength_range)
    scene.hooks = b.hooks
    return scene
