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


def _recognise_user_surface(surf):
    """Device form of a USER-DEFINED surface (a `Surface` subclass with its own `f`, `normal`, `within_boundary`,
    `get_bbox_local`: surfaces.py:5-65), found by MEASURING the object, or an AdapterError that says what was measured.
    The reference treats every surface through these four methods (optical_component.py:151-233, 536-717), so a user
    surface that is numerically the same function as a shape the kernels know gets that shape's kernels and the reference's
    results.  Two families are recognised, both verified on sample points before anything is accepted:
      planar   f(P) = c * x, normal +x, an aperture that is the disc or the centred rectangle of its box  -> CIRCLE / RECT
      revolved f(P) = c * (x + F(r)), r = |(y, z)|, aperture r <= R with R the half-width of its box, normal (and `roc`, if the
               surface has one) equal to the reference ASphere's for that F (surfaces.py:339-423)          -> the verified
               Chebyshev series of F (shapes.lower_asphere_callable: the device form of ASphere(R, callable))
    Anything else — a saddle, an off-axis section, an aperture with a hole — is refused: an implicit function in Python has
    no device form, and evaluating it on the host would be a CPU path of the hot loop."""
    name = type(surf).__name__

    def refuse(why):
        return AdapterError(f"user-defined surface {name}: {why}; recognised are planar surfaces with a disc or centred-rectangle "
                            "aperture and surfaces of revolution x = -F(r) about the local x axis")

    try:
        box = np.array([float(v) for v in surf.get_bbox_local()])
    except Exception as exc:  # noqa: BLE001
        raise refuse(f"get_bbox_local() failed ({exc})") from exc
    if box.shape != (6,) or not np.all(np.isfinite(box)):
        raise refuse("its local box is not six finite numbers")
    x0, x1, y0, y1, z0, z1 = box
    P3 = lambda x, y, z: np.array([x, y, z], dtype=float)  # noqa: E731
    f = lambda P: float(surf.f(P))  # noqa: E731
    rng = np.random.default_rng(20240611)
    try:
        c = f(P3(1, 0, 0)) - f(P3(0, 0, 0))
        if not np.isfinite(c) or c == 0.0:
            raise refuse("f does not depend linearly on x")
        scale = max(abs(x0), abs(x1), abs(y0), abs(y1), abs(z0), abs(z1), 1e-300)
        if bool(getattr(surf, "planar", True)):
            hy, hz = 0.5 * (y1 - y0), 0.5 * (z1 - z0)
            if abs(y0 + y1) > 1e-12 * scale or abs(z0 + z1) > 1e-12 * scale or hy <= 0 or hz <= 0:
                raise refuse("its aperture box is not centred on the local origin")
            for _ in range(64):
                P = P3(rng.uniform(-scale, scale), rng.uniform(-1.2 * hy, 1.2 * hy), rng.uniform(-1.2 * hz, 1.2 * hz))
                if abs(f(P) - c * P[0]) > 1e-12 * abs(c) * scale:
                    raise refuse("planar, but f(P) is not c * x")
                if not np.allclose(np.asarray(surf.normal(P), dtype=float), [1.0, 0.0, 0.0], atol=1e-12):
                    raise refuse("planar, but its normal is not +x")
            pts = [P3(0.0, rng.uniform(-1.2 * hy, 1.2 * hy), rng.uniform(-1.2 * hz, 1.2 * hz)) for _ in range(400)]
            pts += [P3(0.0, y, z) for y in np.linspace(-1.1 * hy, 1.1 * hy, 47) for z in np.linspace(-1.1 * hz, 1.1 * hz, 47)]  # (a hole or a notch larger than ~5 % of the box is seen)
            pts += [P3(0.0, s * hy * (1 + e), 0.0) for s in (-1, 1) for e in (-1e-6, 1e-6)]
            pts += [P3(0.0, 0.0, s * hz * (1 + e)) for s in (-1, 1) for e in (-1e-6, 1e-6)]
            pts += [P3(0.0, s * hy * 0.9, t * hz * 0.9) for s in (-1, 1) for t in (-1, 1)]  # corners: inside the rectangle, outside the disc
            got = np.array([bool(surf.within_boundary(P)) for P in pts])
            yz = np.array([[P[1], P[2]] for P in pts])
            if abs(hy - hz) <= 1e-12 * scale and np.array_equal(got, np.hypot(yz[:, 0], yz[:, 1]) <= hy):
                return Lowered(shapes.CIRCLE, [hy])
            if np.array_equal(got, (np.abs(yz[:, 0]) <= hy) & (np.abs(yz[:, 1]) <= hz)):
                return Lowered(shapes.RECT, [hy, hz])
            raise refuse("planar, but its aperture is neither the disc nor the rectangle of its box")
        R = y1
        if R <= 0 or max(abs(y0 + R), abs(z0 + R), abs(z1 - R)) > 1e-12 * scale:
            raise refuse("its box is not the square -R..R in y and z that a surface of revolution about x has")
        F = lambda r: f(P3(0.0, r, 0.0)) / c  # noqa: E731   x = -F(r)
        for _ in range(96):
            r, th, x = rng.uniform(0, 1.2 * R), rng.uniform(0, 2 * np.pi), rng.uniform(-scale, scale)
            P = P3(x, r * np.cos(th), r * np.sin(th))
            want = c * (x + F(r))
            if abs(f(P) - want) > 1e-11 * max(abs(want), abs(c) * scale):
                raise refuse("f(P) is not c * (x + F(r)) with r the distance from the x axis")
        twin = shapes.ASphere(R, F)
        for k in range(600):
            r = R * (1 + 1e-6) if k == 0 else (R * (1 - 1e-6) if k == 1 else (rng.uniform(0, 1.3 * R) if k < 200 else 1.25 * R * ((k - 200) % 20 + 0.5) / 20))
            th = rng.uniform(0, 2 * np.pi) if k < 200 else 2 * np.pi * ((k - 200) // 20) / 20  # (then a polar grid: 20 radii x 20 angles)
            P = P3(-F(min(r, R)), r * np.cos(th), r * np.sin(th))
            if bool(surf.within_boundary(P)) != (r <= R):
                raise refuse("its aperture is not r <= R")
            if r <= R:
                if not np.allclose(np.asarray(surf.normal(P), dtype=float), twin.normal(P), rtol=0, atol=2e-7):
                    raise refuse("its normal differs from the normal of x = -F(r) (unit vector along (1, F' y / r, F' z / r))")
                if callable(getattr(surf, "roc", None)) and r > 1e-3 * R:
                    a, b = float(surf.roc(P)), float(twin.roc(P))
                    if not np.isclose(a, b, rtol=1e-5, atol=0):
                        raise refuse("its roc(P) differs from the radius of curvature of x = -F(r)")
        low = shapes.lower_asphere_callable(float(R), F)
    except AdapterError:
        raise
    except NotImplementedError as exc:  # the series fit's own refusal (a kink, a pole, F undefined out to the box corner)
        raise refuse(str(exc)) from exc
    except Exception as exc:  # noqa: BLE001 - the user's methods failed on a probe point
        raise refuse(f"probing it failed ({type(exc).__name__}: {exc})") from exc
    return low


def _user_overrides(surf):
    """True when a class OUTSIDE this package and the reference package defines one of the four methods a surface is known by
    (a subclass of Circle that cuts a hole into `within_boundary`, say): the built-in lowering would ignore it."""
    for klass in type(surf).__mro__:
        if _library_class(klass):
            return False
        if any(m in klass.__dict__ for m in ("f", "normal", "within_boundary", "get_bbox_local")):
            return True
    return False


def _surface_state(surf):
    """What a user surface's measurement depends on: its class and its instance attributes (numbers, strings, arrays by value;
    anything else by identity)."""
    items = []
    for k, v in sorted(vars(surf).items()):
        if k.startswith("_ot_"):
            continue
        if isinstance(v, (int, float, bool, str, type(None))):
            items.append((k, v))
        elif isinstance(v, np.ndarray):
            items.append((k, v.tobytes()))
        elif isinstance(v, (tuple, list)):
            items.append((k, repr(v)))
        else:
            items.append((k, id(v)))
    return (type(surf), tuple(items))


def _measured(surf):
    """`_recognise_user_surface`, remembered on the object while its state stands: the measurement is ~1,500 calls of the user's
    methods and a series fit (20-60 ms), and `table.ray_tracing` compiles the scene on every call."""
    state = _surface_state(surf)
    memo = surf.__dict__.get("_ot_lowered")
    if memo is None or memo[0] != state:
        memo = (state, _recognise_user_surface(surf))
        surf.__dict__["_ot_lowered"] = memo
    return memo[1]


def lower_surface(surf):
    if _user_overrides(surf):
        return _measured(surf)  # measured, never assumed from the base class (see _recognise_user_surface)
    if hasattr(surf, "lower"):
        try:
            return surf.lower()
        except NotImplementedError:
            if type(surf).lower is not shapes.Surface.lower:
                raise
            return _measured(surf)  # a user's subclass of this package's Surface: measured (see _recognise_user_surface)
    fn = _SURFACES.get(type(surf).__name__)
    if fn is None:
        if _closure_operands(surf) is not None:  # the closure-based Plane.union / subtract of the reference
            return _boolean_plane(surf)
        if all(callable(getattr(surf, m, None)) for m in ("f", "normal", "within_boundary", "get_bbox_local")):
            return _measured(surf)  # a user's subclass of the reference's Surface
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
def _library_class(klass):
    """A class of this package or of the reference package (not the synthetic user parts of optable_amd.workloads, which stand
    for a user's script)."""
    module = klass.__module__ or ""
    return module.split(".")[0] in ("optable_amd", "optable") and module != "optable_amd.workloads"


def host_hook(comp):
    """True when `comp.interact_local` is the USER's: defined by a class outside this package and outside the reference
    package (optical_component.py:235-240 is the subclassing hook).  Such a leaf keeps its place in the device scene — the
    nearest-hit search, its boxes and its count gate are the kernels' — and its physics is the user's Python, called by
    `table.ray_tracing` on the rays the device found to hit it (table.py: _trace_hooked)."""
    if comp.__dict__.get("_builtin_physics"):  # table.interact_leaf_local: the built-in physics below a user's override
        return False
    for klass in type(comp).__mro__:
        if "interact_local" in klass.__dict__:
            return not _library_class(klass)
    return False


def lower_interaction(comp):
    if host_hook(comp):
        # The host asks the user's method what a hit emits.  When the class ALSO has built-in physics underneath the override (a
        # Mirror subclass whose interact_local post-processes super()'s rays), the device computes those children with the hit, so
        # that `super().interact_local(ray)` inside the hook is answered from the generation that found the hit instead of a
        # launch of its own (table.py: _HOOK_MEMO); otherwise the leaf simply ends the ray.
        try:
            comp.__dict__["_builtin_physics"] = True
            inner = dict(lower_interaction(comp))
        except NotImplementedError:
            inner = dict(kind=BLOCK)
        finally:
            comp.__dict__.pop("_builtin_physics", None)
        inner["host_hook"] = True
        return inner
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
        max_time = _table.MAX_TRACE_TIME
        if perfomance_limit is not None and "max_trace_time" in perfomance_limit:
            max_time = float(perfomance_limit["max_trace_time"])
        if len(rays) and cap > 0 and max_time > 0:
            traced, capped = _table.OpticalTable._trace_objects(self, list(rays), cap, max_time)
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
