"""Route the ORIGINAL optable package's `OpticalTable.ray_tracing` through the MI355X engine.

    import optable, optable_amd
    optable_amd.install(optable)      # patches optable.OpticalTable.ray_tracing (and Monitor.record)
    ...                               # the user's script, rendering, GUI etc. stay untouched

Everything but the hot path remains the reference's own code.  The scene compiler needs a device
form for every shape / interaction / material; objects of this package provide it themselves
(`lower()`, `lower_interaction()`, `device_spec()`), objects of the reference are recognised here by
class name and attributes (duck typing — no reference code is imported or copied).  Anything that
cannot be recognised raises `SceneError`; nothing is silently evaluated on the host.
"""
import numpy as np

from . import shapes
from .components import MIRROR, REFRACT, LENS, BLOCK, ROC_INF, ROC_CONST, ROC_ASPHERE
from .shapes import Lowered


class AdapterError(NotImplementedError):
    pass


def _mro_names(obj):
    return [c.__name__ for c in type(obj).__mro__]


# ---------------------------------------------------------------------------------------------
# surfaces (reference optable/surfaces.py, recognised by class name + the attributes they carry)
def _polygon(surf):
    u, v = surf._basis
    rec = [float(len(surf._verts2d))] + list(surf._normal) + list(surf.vertices[0]) + list(u) + list(v)
    rec += [c for xy in surf._verts2d for c in xy]
    return Lowered(shapes.POLYGON2D if surf.planar else shapes.POLYGON3D, aux=rec, planar=bool(surf.planar))


def _asphere(surf):
    """The reference builds the sag as a closure; recover its parameters from the closure cells and
    CHECK them by evaluating both forms."""
    f = surf.f_asphere
    spec = getattr(f, "device_spec", None)
    if spec is None:
        cells = dict(zip(getattr(f.__code__, "co_freevars", ()), [c.cell_contents for c in (f.__closure__ or ())]))
        if {"R", "kappa", "a4", "a6", "a8"} <= set(cells):
            spec = (shapes.ASPHERE_PARAM, tuple(float(cells[k]) for k in ("R", "kappa", "a4", "a6", "a8")))
            mine = shapes.sag_parametric(*spec[1])
        elif {"EFL", "n"} <= set(cells):
            spec = (shapes.ASPHERE_EXACT, (float(cells["EFL"]), float(cells["n"])))
            mine = shapes.sag_exact(*spec[1])
        else:  # any other function of r: a verified Chebyshev series, or a refusal that says why (shapes.py)
            return shapes.lower_asphere_callable(float(surf.radius), f)
        probe = np.linspace(0.0, float(surf.radius), 7)
        if not np.allclose([f(r) for r in probe], [mine(r) for r in probe], rtol=1e-13, atol=1e-15):
            raise AdapterError("ASphere closure does not match the recognised sag formula")
    return Lowered(spec[0], [surf.radius] + list(spec[1]), planar=False)


_SURFACES = {
    "Circle": lambda s: Lowered(shapes.CIRCLE, [s.radius]),
    "Rectangle": lambda s: Lowered(shapes.RECT, [s.width / 2, s.height / 2]),
    "Polygon": _polygon,
    "Sphere": lambda s: Lowered(shapes.SPHERE, [s.radius, s.height], planar=False),
    "ASphere": _asphere,
    "Cylinder": lambda s: Lowered(shapes.CYLINDER, [s.radius, s.height / 2, s.theta_range[0], s.theta_range[1]], planar=False),
    "Point": lambda s: Lowered(shapes.POINT, planar=False),
}


def _closure_operands(surf):
    """(left, right) of a reference `Plane.union` / `Plane.subtract` result, or None.  Upstream builds those as
    a fresh Plane whose `within_boundary` is a closure over `self` and `other` (surfaces.py:100-136)."""
    fn = getattr(surf, "__dict__", {}).get("within_boundary")
    code = getattr(fn, "__code__", None)
    if code is None or not fn.__closure__ or set(code.co_freevars) != {"self", "other"}:
        return None
    cells = dict(zip(code.co_freevars, (c.cell_contents for c in fn.__closure__)))
    return cells["self"], cells["other"]


def _probe_points(box, n=9):
    _, _, y0, y1, z0, z1 = (float(v) for v in box)
    ys, zs = np.linspace(y0, y1, n), np.linspace(z0, z1, n)
    return [np.array([0.0, y, z]) for y in ys for z in zs]


def _boolean_postfix(surf, out):
    """Postfix program (shapes.BooleanPlane layout) of a closure-built aperture.  The operator is not
    readable from the closure, so it is MEASURED: at a point inside `other` a union is true and a subtraction
    false."""
    pair = _closure_operands(surf)
    if pair is None:
        low = lower_surface(surf)
        if low.kind not in (shapes.CIRCLE, shapes.RECT, shapes.POLYGON2D):
            raise AdapterError(f"boolean aperture operand {type(surf).__name__} has no device form")
        body = low.aux if low.kind == shapes.POLYGON2D else low.params
        out.extend([float(low.kind), float(len(body))] + list(body))
        return
    left, right = pair
    inside_right = next((P for P in _probe_points(right.get_bbox_local()) if right.within_boundary(P)), None)
    if inside_right is None:
        raise AdapterError("boolean aperture: no probe point inside the second operand")
    op = shapes.CSG_OR if surf.within_boundary(inside_right) else shapes.CSG_ANDNOT
    _boolean_postfix(left, out)
    _boolean_postfix(right, out)
    out.extend([float(op), 0.0])


def _boolean_plane(surf):
    prog = []
    _boolean_postfix(surf, prog)
    n_tokens = sum(1 for _ in shapes._walk_tokens(prog))
    low = Lowered(shapes.CSG, aux=[float(n_tokens)] + prog)
    # check the recovered program against the closure on a grid over the aperture's box
    def run(P):
        stack, i = [], 0
        while i < len(prog):
            kind, ln = int(prog[i]), int(prog[i + 1])
            body = prog[i + 2:i + 2 + ln]
            if kind == shapes.CSG_OR:
                b, a = stack.pop(), stack.pop(); stack.append(a or b)
            elif kind == shapes.CSG_ANDNOT:
                b, a = stack.pop(), stack.pop(); stack.append(a and not b)
            elif kind == shapes.CIRCLE:
                stack.append(float(np.linalg.norm(P)) <= body[0])
            elif kind == shapes.RECT:
                stack.append(abs(P[1]) <= body[0] and abs(P[2]) <= body[1])
            else:
                return None  # polygons are checked through their own lowering
            i += 2 + ln
        return stack[0]
    for P in _probe_points(surf.get_bbox_local(), 13):
        mine = run(P)
        if mine is not None and bool(mine) != bool(surf.within_boundary(P)):
            raise AdapterError("boolean aperture: recovered program disagrees with the closure")
    return low


def lower_surface(surf):
    if hasattr(surf, "lower"):
        return surf.lower()
    fn = _SURFACES.get(type(surf).__name__)
    if fn is None:
        if _closure_operands(surf) is not None:  # the closure-based Plane.union / subtract of the reference
            return _boolean_plane(surf)
        raise AdapterError(f"surface {type(surf).__name__} has no device form")
    return fn(surf)


# ---------------------------------------------------------------------------------------------
# materials (reference optable/material.py)
class _Spec:
    def __init__(self, spec, name):
        self._spec, self.name = spec, name

    def device_spec(self):
        return self._spec


def lower_material(mat):
    if hasattr(mat, "device_spec"):
        return mat
    if hasattr(mat, "Bs") and hasattr(mat, "Cs") and len(mat.Bs) == len(mat.Cs) <= 3:
        pad = 3 - len(mat.Bs)
        return _Spec(("sellmeier", list(mat.Bs) + [0.0] * pad, list(mat.Cs) + [-1.0] * pad), getattr(mat, "name", "?"))
    if hasattr(mat, "n_func"):
        from .materials import callable_spec

        return _Spec(callable_spec(mat.n_func, getattr(mat, "wavelength_range", None)), getattr(mat, "name", "?"))
    return _Spec(None, getattr(mat, "name", "?"))  # the scene compiler reports it


# ---------------------------------------------------------------------------------------------
# interactions (reference optable/optical_component.py)
def host_hook(comp):
    """True when `comp.interact_local` is the USER's: defined by a class outside this package and outside the reference
    package (optical_component.py:235-240 is the subclassing hook).  Such a leaf keeps its place in the device scene — the
    nearest-hit search, its boxes and its count gate are the kernels' — and its physics is the user's Python, called by
    `table.ray_tracing` on the rays the device found to hit it (table.py: _trace_hooked)."""
    if comp.__dict__.get("_builtin_physics"):  # table.interact_leaf_local: the built-in physics below a user's override
        return False
    for klass in type(comp).__mro__:
        if "interact_local" in klass.__dict__:
            root = (klass.__module__ or "").split(".")[0]
            return root not in ("optable_amd", "optable")
    return False


def lower_interaction(comp):
    if host_hook(comp):
        return dict(kind=BLOCK, host_hook=True)  # the device ends the ray at the hit; the host asks the user what it emits
    if hasattr(comp, "lower_interaction"):
        return comp.lower_interaction()
    names = _mro_names(comp)
    if "Monitor" in names:
        raise AdapterError("a Monitor inside `components` would re-emit its own ray forever (monitor.py:174-175)")
    if "Lens" in names:
        return dict(kind=LENS, transmission=comp.transmission, focal_length=comp.focal_length)
    if "BaseMirror" in names:
        return dict(kind=MIRROR, reflectivity=comp.reflectivity, transmission=comp.transmission)
    if "BaseRefraciveSurface" in names:
        roc = getattr(comp, "roc", np.inf)
        if callable(roc):
            if getattr(roc, "__self__", None) is not comp.surface:
                raise AdapterError("callable roc that is not the surface's own ASphere.roc has no device form")
            kind, val = ROC_ASPHERE, 0.0
        elif np.isinf(roc):
            kind, val = ROC_INF, np.inf
        else:
            kind, val = ROC_CONST, float(roc)
        return dict(kind=REFRACT, reflectivity=comp.reflectivity, transmission=comp.transmission,
                    mat1=lower_material(comp.__dict__["_n1"]), mat2=lower_material(comp.__dict__["_n2"]),
                    roc_kind=kind, roc=val)
    if "Block" in names or "PointObj" in names:
        return dict(kind=BLOCK)
    raise AdapterError(f"{type(comp).__name__} has no device interaction")


# ---------------------------------------------------------------------------------------------
def install(optable_module):
    """Patch `optable_module.OpticalTable.ray_tracing` (and `Monitor.record`) to run on the engine.
    Returns a function that undoes the patch."""

    from . import table as _table

    OT, Mon = optable_module.OpticalTable, optable_module.Monitor
    original = (OT.ray_tracing, Mon.record)

    def ray_tracing(self, rays, perfomance_limit=None):
        if isinstance(rays, optable_module.Ray):
            rays = [rays]
        cap = _table.MAX_TRACE_NUM
        if perfomance_limit is not None and "max_trace_num" in perfomance_limit:
            cap = int(perfomance_limit["max_trace_num"])
        if len(rays) and cap > 0:
            traced, capped = _table.OpticalTable._trace_objects(self, list(rays), cap)
            if capped:
                print(f"Ray tracing time exceeds the maximum tracing time after {cap} traces. ({capped} ray tree(s) truncated)")
            self.rays.extend(traced)
        return _table._clone_rays(self.rays)

    def record(self, rays):
        _table.record_monitor_hits(self, rays)

    def compile_(self):
        from .scene import compile_scene

        return compile_scene(self.components, getattr(self, "unit", 1e-2))

    OT.ray_tracing, OT.compile, Mon.record = ray_tracing, compile_, record

    def uninstall():
        OT.ray_tracing, Mon.record = original
        del OT.compile

    return uninstall
