"""Parity scenes (SURVEY.md §8c G1-G13 and §8d cfg 2-5), written against a namespace `ns` so
the SAME builder runs on the reference (tools/make_golden.py, `ns = optable`) and on this
package (`ns = optable_amd`).  Each builder returns a dict:
    components, monitors, rays, limit (perfomance_limit or None)
Scene contents follow the reference's example scripts (cited per scene); synthetic configs
follow SURVEY.md §8(d) verbatim (seeds, draw order).
"""
import numpy as np

# the BASELINE configs (scene + ray generators) live in the package, where bench.py and the tools find them too
from optable_amd.workloads import (WL, W0, cfg2_rays, cfg2_components, cfg3_rays, cfg3_components,  # noqa: F401
                                   cfg4_rays, cfg4_components, cfg5_rays, cfg5_components)


def g01_gaussian_beam(ns):
    """examples/gaussian_beam.py:16-44 — mirror, three thin lenses, slab n=2, mirror."""
    wl, w0 = 780e-9, 10e-6
    R = ns.Ray
    rays = [R([-10, y, 0], [1, 0, 0], wavelength=wl, w0=w0) for y in (0, 2, 4, 6, 9)]
    rays.append(R([-10, 21, 0], [1, 0, 0], wavelength=wl, w0=w0).RotZ(-np.pi / 4))
    comps = [ns.Mirror([0, 0, 0]).RotZ(np.pi / 6), ns.Lens([0, 2, 0], radius=0.8, focal_length=5),
             ns.Lens([0, 4, 0], radius=0.8, focal_length=10), ns.Lens([0, 6.5, 0], radius=0.8, focal_length=10),
             ns.GlassSlab([0, 9, 0], n1=1, n2=2, thickness=5), ns.Mirror([0, 11, 0]).RotZ(-np.pi / 2)]
    return dict(components=comps, monitors=[], rays=rays, limit=None)


def g02_cfg2(ns, n=1000):
    o, d = cfg2_rays(n, 0)
    rays = [ns.Ray(o[i], d[i], wavelength=WL, w0=W0, id=i) for i in range(n)]
    return dict(components=cfg2_components(ns), monitors=[], rays=rays, limit={"max_trace_num": 5})


def g03_chromatic(ns):
    """examples/chromatic_aberration.py:16-40 — NBK7 slab, reflectivity 0.2 (branching, shared _id)."""
    r0 = [ns.Ray([-3, 2, 0], [np.cos(np.pi / 6), -np.sin(np.pi / 6), 0], wavelength=780e-7, w0=20e-4).Propagate(-2)]
    rays = ns.multiplex_rays_in_wavelength(r0, [780e-7, 560e-7, 400e-7])
    gs = ns.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=ns.Vacuum(), n2=ns.Glass_NBK7(), reflectivity=0.2)
    return dict(components=[gs], monitors=[], rays=rays, limit=None)


def g04_glass_slab(ns):
    """examples/glass_slab.py:26-43."""
    rays = [ns.Ray([-3, 2, 0], [np.cos(np.pi / 6), -np.sin(np.pi / 6), 0], wavelength=780e-7, w0=2e-4).Propagate(-2)]
    gs = ns.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=1, n2=1.5, reflectivity=0.2)
    return dict(components=[gs], monitors=[], rays=rays, limit=None)


def g05_cavity(ns):
    """examples/cavity_4mir.py:18-40 — R=0.9 ring cavity; branches until the 2000 cap."""
    L, D, R = 10 * 4 / 3, 4, 0.9
    dt1 = dt2 = 0.02
    comps = [ns.Mirror([0, 0, 0], radius=D, reflectivity=R).RotZ(-np.pi / 4),
             ns.Mirror([L, 0, 0], radius=D, reflectivity=R).RotZ(+np.pi / 4 + dt1),
             ns.Mirror([L, -L, 0], radius=D, reflectivity=R).RotZ(-np.pi / 4 + dt2),
             ns.Mirror([0, -L, 0], radius=D, reflectivity=R).RotZ(+np.pi / 4)]
    return dict(components=comps, monitors=[], rays=[ns.Ray([2, 0, 0], [1, 0, 0])], limit={"max_trace_num": 300})


def g06_mirror_pair(ns):
    """examples/mirror_pair.py:28-39 — 3-D rotated MirrorPair and two monitors."""
    r0 = ns.Ray([-10, 1, 0], [1, 0, 0])._RotAround([0, 1, 0], [0, 0, 0], 0.1)
    mp = ns.MirrorPair([3, 0, 0], 4, 4).RotX(0.3)
    mons = [ns.Monitor([1, 0, 0], 5, 5), ns.Monitor([-3, 0, 0], 5, 5)]
    return dict(components=[mp], monitors=mons, rays=[r0], limit=None)


def _fan(ns, N=3, D=0.6):
    """examples/calibrate_4f.py:187-192 — 7-ray fan with explicit ids."""
    return [ns.Ray([-10, i * D, 0], [1, 0, 0], wavelength=780e-7, w0=61e-4, id=int(i + N)).Propagate(-10)
            for i in np.arange(-N, N + 1)]


def g07_spherical_lenses(ns):
    """BiConvexLens + Doublet(N-SK2 / N-SF57) (examples/calibrate_4f.py:103-125)."""
    bi = ns.BiConvexLens([5, 0, 0], CT=0.6, R1=20.0, R2=-20.0, diameter=5.08, EFL=20.0, name="B0")
    dbl = ns.Doublet([26.4975, 0, 0], CT1=0.85, CT2=0.50, R1=12.1, R2=-11.87, R3=-42.24,
                     n12=ns.Glass_NSK2(), n23=ns.Glass_NSF57(), diameter=2.54 * 2, name="L0")
    mon = ns.Monitor([60, 0, 0], 10, 10)
    return dict(components=[bi, dbl], monitors=[mon], rays=_fan(ns), limit=None)


def g08_asphere(ns):
    """ASphericParametricLens LENS==9 (examples/calibrate_4f.py:149-180) + the 7-ray fan."""
    EFL, CT = 43.17, 0.8
    n = ns.Glass_UVFS()
    R = EFL * (n.n(780e-9) - 1)
    lens = ns.ASphericParametricLens([EFL, 0, 0], CT=CT, diameter=2.54 * 3, R=R, n=n, kappa=-1.03113,
                                     a4=-0.00223 * (1e-3 / 1e-2) ** 4, a6=0.006353 * (1e-3 / 1e-2) ** 6, name="L0")
    exact = ns.ASphericExactSphericalLens([70, 0, 0], EFL=20.0, CT=0.5, diameter=5.0, n=1.6)
    mon = ns.Monitor([95, 0, 0], 10, 10)
    return dict(components=[lens, exact], monitors=[mon], rays=_fan(ns, D=0.55), limit=None)


def g09_cfg5(ns, n=40):
    o, d = cfg5_rays(n, 3)
    rays = [ns.Ray(o[i], d[i], wavelength=WL, w0=W0, id=i) for i in range(n)]
    return dict(components=cfg5_components(ns), monitors=[], rays=rays, limit={"max_trace_num": 50})


def g10_cfg3(ns, n=400):
    o, d = cfg3_rays(n, 2)
    rays = [ns.Ray(o[i], d[i], wavelength=WL, w0=W0, id=i) for i in range(n)]
    return dict(components=cfg3_components(ns), monitors=[], rays=rays, limit={"max_trace_num": 20})


def g11_prism_refl(ns):
    """examples/prism_refl.py:25-70 — TriangularPrism, TIR, max_interact_count, explicit ids."""
    theta, L = 0.00956, 6
    n = ns.Glass_NBK7().n(780e-9)
    D = 3 - 210e-4
    rays = [ns.Ray([3, y + L / 2, 0], [-1, 0, 0], wavelength=780e-7, w0=61e-4, id=i).Propagate(-3).RotZ(theta)
            for i, y in enumerate(np.linspace(-D, D, 5))]
    ps = ns.TriangularPrism(origin=[0, 0, 0], width=L, height=L, n1=1, n2=n, alpha=np.pi / 4, beta=np.pi / 2,
                            reflectivity_1=1, reflectivity_3=0.01, max_interact_count_2=10, max_interact_count_3=10)
    mon = ns.Monitor([-3, 0, 0], width=L, height=L, name="Monitor 0")
    return dict(components=[ps], monitors=[mon], rays=rays, limit=None)


def g12_dove(ns):
    """examples/dove_prism.py:25-65 — planar and 3-D Polygon faces, TIR on the base."""
    L, D, Ng = 6.34, 1.515, 1.515
    dp = ns.DovePrism([0, 0, 0], L=L, D=D, Ng=Ng)
    z0 = dp.z0
    R = ns.Ray
    rays = [R([x, -50, z0], [0, 1, 0]) for x in np.linspace(-0.6, 0.6, 7)]
    rays += [R([3, y, z + 1], [-1, 0, -0.3]) for y in np.linspace(-1, 1, 3) for z in np.linspace(-1, 1, 3)]
    rays += [R([x, y, 3], [-0.3, 0, -1]) for x in np.linspace(-1, 1, 3) for y in np.linspace(-1, 1, 3)]
    rays += [R([x, 3, z], [0, -1, 0]) for x in np.linspace(-1, 1, 3) for z in np.linspace(-1, 1, 3)]
    dp = dp.RotX(0.02).RotZ(-0.01).TX(0.05).TZ(-0.03).RotYAroundLocal([0, 0, D / 2], theta=0.015)
    mon = ns.Monitor([0, 10, z0], width=2 * D, height=2 * D, name="Monitor 0").RotZ(np.pi / 2)
    return dict(components=[dp], monitors=[mon], rays=rays, limit=None)


def g13_count_shadow(ns):
    """a far mirror with max_interact_count=1 spends its count while
    shadowed by a nearer pass-through mirror, then becomes transparent."""
    near = ns.Mirror([2, 0, 0], radius=1, reflectivity=0.0, transmission=1.0)
    far = ns.Mirror([4, 0, 0], radius=1, max_interact_count=1)
    back = ns.Mirror([8, 0, 0], radius=1)
    rays = [ns.Ray([0, 0.1 * k, 0], [1, 0, 0], id=k) for k in range(3)]
    rays.append(ns.Ray([0, 0.05, 0], [1, 0, 0], id=0))  # shares counters with ray 0
    return dict(components=[near, far, back], monitors=[], rays=rays, limit={"max_trace_num": 12})


def g15_cfg4(ns, nbase=20, nwl=8):
    """Chromatic slab, non-branching (SURVEY.md §8d cfg 4, reflectivity=0)."""
    rng = np.random.default_rng(4)
    jit = rng.uniform(-0.3, 0.3, (nbase, 2))
    base = [ns.Ray([-3, 2 + jit[i, 0], jit[i, 1]], [np.cos(np.pi / 6), -np.sin(np.pi / 6), 0], wavelength=WL, w0=W0, id=i)
            for i in range(nbase)]
    rays = ns.multiplex_rays_in_wavelength(base, list(np.linspace(400e-7, 1100e-7, nwl)))
    gs = ns.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=ns.Vacuum(), n2=ns.Glass_NBK7(), reflectivity=0)
    return dict(components=[gs], monitors=[], rays=rays, limit=None)


def g16_misc(ns):
    """Shapes not covered above: CylMirror, Block with a hole (Plane.subtract), BeamSplitter,
    WedgePlate, CircleGlassSlab, MirrorCube, PlanoConvexLens, MLA, DMD, finite-length and dead rays."""
    comps = [
        ns.BeamSplitter([2, 0, 0], width=2, height=2, eta=0.3).RotZ(np.pi / 4),
        ns.Block([4, 0, 0], hole=ns.Circle(0.3), width=2, height=2),
        ns.WedgePlate([6, 0, 0], width=2, height=2, thickness=0.4, wedge_angle=0.05, n1=1.0, n2=1.45),
        ns.CircleGlassSlab([8, 0, 0], radius=1.0, thickness=0.3, n1=1.0, n2=1.7, reflectivity2=0.1),
        ns.PlanoConvexLens([10, 0, 0], EFL=15.0, CT=0.4, diameter=2.0, R=7.5),
        ns.MLA([13, 0, 0], N=(3, 3), pitch=0.4, focal_length=3.0, radius=0.2),
        ns.CylMirror([18, 0, 0], radius=2.0, height=3.0, theta_range=(np.pi / 2, np.pi)).RotZ(0.1),
        ns.MirrorCube([2, 6, 0], L=2.0).RotZ(np.pi / 2),
        ns.DMD([2, -6, 0], N=(3, 2), pitch=0.5, tilt_angle=np.pi / 5).RotZ(-np.pi / 2 + 0.2),
    ]
    R = ns.Ray
    rays = [R([0, y, z], [1, 0.002 * k, 0.001], wavelength=WL, w0=W0, id=k)
            for k, (y, z) in enumerate([(0, 0), (0.1, 0.05), (0.25, -0.1), (0.5, 0.2), (-0.45, 0.1), (0.05, 0.29)])]
    rays.append(R([0, 0.2, 0], [1, 0, 0], length=1.5, id=10))          # too short to reach anything
    rays.append(R([0, 0.2, 0], [1, 0, 0], alive=False, id=11))         # dead on input
    rays.append(R([0, 0, 0], [0.1, 1, 0.02], wavelength=WL, w0=W0, id=12))   # up into the corner cube
    rays.append(R([0, 0, 0], [0.12, -1, 0.0], wavelength=WL, w0=W0, id=13))  # down onto the DMD
    return dict(components=comps, monitors=[ns.Monitor([30, 0, 0], 20, 20)], rays=rays, limit={"max_trace_num": 200})


def g18_fifo_gate(ns):
    """Two siblings of one tree reach a max_interact_count=1 mirror in the SAME generation: the
    beam-splitter's reflected branch (queued first, optical_component.py:546-570) must win the count
    and the transmitted branch must find the mirror transparent (optical_component.py:140-149)."""
    a = -np.pi / 4
    comps = [ns.BeamSplitter([2, 0, 0], width=2, height=2, eta=0.5).RotZ(a),
             ns.Mirror([2, 3, 0], radius=1).RotZ(a), ns.Mirror([5, 0, 0], radius=1).RotZ(a),
             ns.Mirror([5, 3, 0], radius=1, max_interact_count=1).RotZ(5 * np.pi / 4),
             ns.Mirror([5, 6, 0], radius=1).RotZ(-np.pi / 2), ns.Mirror([9, 3, 0], radius=1).RotZ(np.pi)]
    rays = [ns.Ray([0, 0, 0.05 * k], [1, 0, 0], wavelength=WL, w0=W0, id=k) for k in range(3)]
    return dict(components=comps, monitors=[], rays=rays, limit={"max_trace_num": 30})


def g19_units_and_disorder(ns):
    """MMADisordered with tilted caps (component_group.py:307-364), MirrorPrism, a dispersive slab hit by
    rays that carry their OWN length unit (mm instead of the default cm: n(lambda*unit), base.py:31), rays
    without q and without wavelength."""
    pts = [[0, 0.25 * (k % 3 - 1), 0.25 * (k // 3 - 1)] for k in range(9)]
    tilt = [[-1.0, 0.02 * (k % 3 - 1), -0.015 * (k // 3 - 1)] for k in range(9)]
    comps = [ns.GlassSlab([3, 0, 0], width=3, height=3, thickness=0.6, n1=ns.Vacuum(), n2=ns.Glass_NSF11()).RotZ(0.2),
             ns.MMADisordered([9, 0, 0], PList=pts, pitch=0.25, roc=6.0, n=1.5, thickness=0.2, nList=tilt, reflectivity=1, transmission=0),
             ns.MirrorPrism([-4, 0, 0], width=3, height=3, angle=np.pi / 2).RotZ(np.pi)]
    R = ns.Ray
    rays = [R([0, 0.05 * k - 0.2, 0.03 * k - 0.1], [1, 0.004 * k, -0.002 * k], wavelength=600e-7 + 40e-7 * k, w0=50e-4, id=k) for k in range(6)]
    rays += [R([0, 0.1, 0.0], [1, 0, 0.01], wavelength=6000e-7, w0=50e-3, id=10, unit=1e-3),   # millimetres: same 600 nm
             R([0, -0.1, 0.05], [1, 0.01, 0], id=11),                                           # no wavelength, no q
             R([0, 0.15, -0.05], [1, -0.01, 0.004], wavelength=450e-7, id=12)]                  # wavelength, no q
    return dict(components=comps, monitors=[ns.Monitor([1.5, 0, 0], 4, 4)], rays=rays, limit={"max_trace_num": 40})


def abcd_4f(ns):
    """4f relay of two bi-convex lenses between two monitors (the system calibrate_symmetric_4f builds,
    optical_table.py:328-340); used for calculate_abcd_matrix parity (optical_table.py:211-297)."""
    F1, F2 = 19.7, 20.4
    mk = lambda x: ns.BiConvexLens([x, 0, 0], CT=0.6, R1=20.0, R2=-20.0, diameter=5.08, EFL=20.0)
    l0, l1 = mk(F1), mk(F1 + 2 * F2).RotZ(np.pi)
    mon0 = ns.Monitor(origin=[0, 0, 0], width=5, height=5)
    mon1 = ns.Monitor(origin=[2 * F1 + 2 * F2, 0, 0], width=5, height=5)
    rays = [ns.Ray([-10, i * 0.3, 0], [1, 0, 0], wavelength=780e-7, w0=61e-4, id=int(i + 3)).Propagate(-10)
            for i in np.arange(-3, 4)]
    return dict(components=[l0, l1], monitors=[mon0, mon1], rays=rays, limit=None)


def calibrate_case(ns):
    """Inputs of calibrate_symmetric_4f (optical_table.py:299-422): one bi-convex lens, the 7-ray fan with ids."""
    lens = ns.BiConvexLens([0, 0, 0], CT=0.6, R1=20.0, R2=-20.0, diameter=5.08, EFL=20.0)
    rays = [ns.Ray([-10, i * 0.3, 0], [1, 0, 0], wavelength=780e-7, w0=61e-4, id=int(i + 3)).Propagate(-10)
            for i in np.arange(-3, 4)]
    return dict(lens=lens, rays=rays, F10=19.7, F20=20.4)


def g21_ties(ns):
    """Exact ties: two mirrors in the same plane (the first in list order must win: strict `t < t_min`,
    optical_table.py:119-123) and a group whose two children coincide (np.argmin keeps the first,
    component_group.py:118-120)."""
    near_a = ns.Mirror([3, 0, 0], radius=1.0, reflectivity=0.25)
    near_b = ns.Mirror([3, 0, 0], radius=1.0, reflectivity=0.75)
    pair = ns.ComponentGroup([6, 0, 0])
    pair.add_component(ns.Lens([6, 0, 0], focal_length=4.0, radius=1.0))
    pair.add_component(ns.Mirror([6, 0, 0], radius=1.0))
    back = ns.Mirror([-2, 0, 0], radius=2.0).RotZ(np.pi)
    rays = [ns.Ray([0, y, 0], [1, 0, 0], wavelength=WL, w0=W0) for y in (-0.3, 0.0, 0.2)]
    rays += [ns.Ray([4.5, y, 0], [1, 0.01, 0], wavelength=WL, w0=W0) for y in (-0.1, 0.25)]   # start beyond the mirrors
    return dict(components=[back, near_a, near_b, pair], monitors=[], rays=rays, limit={"max_trace_num": 6})


def g24_callables(ns):
    """User functions in the scene (component_group.py:1014-1055 ASphericLens(f_asphere_1 / _2 = callable),
    material.py:4-21 Material(name, n = callable)): a lens with two sags that are none of the recognised closed
    forms, made of a glass whose dispersion is a Python function, a tilted slab of a second such glass, a mirror
    that sends the light back through both; three wavelengths."""
    sag_front = lambda r: 0.5 * (np.cosh(0.3 * r) - 1.0) + 1e-3 * r**4          # noqa: E731
    sag_back = lambda r: -0.02 * r**2 / (1.0 + 0.1 * r**2)                        # noqa: E731
    flint = ns.Material("cauchy flint", n=lambda wl_m: 1.6 + 8e-15 / wl_m**2)
    crown = ns.Material("two-term crown", n=lambda wl_m: np.sqrt(2.2 + 6e-15 / wl_m**2 - 1e9 * wl_m**2))
    lens = ns.ASphericLens([6, 0, 0], CT=0.7, f_asphere_1=sag_front, f_asphere_2=sag_back, diameter=2.6, n=flint)
    slab = ns.GlassSlab([12, 0.1, 0], width=3, height=3, thickness=0.6, n1=ns.Vacuum(), n2=crown).RotZ(0.3)
    mirror = ns.Mirror([16, 0, 0], radius=2.0).RotZ(np.pi + 0.04)
    rays = []
    for wl in (450e-7, 633e-7, 850e-7):
        for y in (-0.9, -0.45, 0.0, 0.3, 0.75):
            for z in (-0.4, 0.2):
                rays.append(ns.Ray([0, y, z], [1, 0.01 * y, -0.02 * z], wavelength=wl, w0=W0))
    return dict(components=[lens, slab, mirror], monitors=[], rays=rays, limit={"max_trace_num": 24})


def g25_user_components(ns):
    """User-defined components (optical_component.py:235-240: `interact_local` is the subclassing hook): a transmission
    grating written against the public API (three orders, the hit point asked from `intersect_point_local`), a mirror
    subclass that post-processes `super().interact_local()` and lets weak rays die, an absorber that emits nothing — next
    to built-in parts that branch on the device (a slab with reflectivity) and a count-limited mirror; one grating sits
    inside a group (lab AABB gate).  The classes are built here from `ns`, so the reference and this package run the
    very same user code."""

    class Grating(ns.OpticalComponent):
        def __init__(self, origin, radius=1.0, pitch=2e-4, **kwargs):
            super().__init__(origin, **kwargs)
            self.surface = ns.Circle(radius)
            self.pitch = pitch

        def get_bbox_local(self):
            return self.surface.get_bbox_local()

        def interact_local(self, ray):
            P, t = self.intersect_point_local(ray)
            out = []
            for order, share in ((-1, 0.25), (0, 0.5), (1, 0.25)):
                d = np.array(ray.direction, dtype=float)
                d[1] += order * ray.wavelength / self.pitch
                s = 1.0 - d[1] ** 2 - d[2] ** 2
                if s <= 0:
                    continue  # evanescent order
                d[0] = np.sign(d[0]) * np.sqrt(s)
                out.append(ray.copy(origin=P, direction=d, intensity=ray.intensity * share,
                                    qo=None if ray.qo is None else ray.q_at_z(t), _pathlength=ray.pathlength(float(t))))
            return out

    class LossyMirror(ns.Mirror):
        def interact_local(self, ray):
            rays = super().interact_local(ray)
            for r in rays:
                r.intensity *= 0.6
                if r.intensity < 0.05:
                    r.alive = False  # archived at once, never traced (optical_table.py:126-130)
            return rays

    class Absorber(ns.OpticalComponent):
        def __init__(self, origin, **kwargs):
            super().__init__(origin, **kwargs)
            self.surface = ns.Rectangle(3, 3)

        def get_bbox_local(self):
            return self.surface.get_bbox_local()

        def interact_local(self, ray):
            return []

    grating = Grating([4, 0, 0], radius=1.2).RotZ(0.1)
    slab = ns.GlassSlab([8, 0, 0], width=4, height=4, thickness=0.5, n1=1, n2=1.5, reflectivity=0.1).RotZ(0.2)
    group = ns.ComponentGroup([12, 0, 0])
    group.add_component(Grating([12, 0.6, 0], radius=0.5, pitch=3e-4))
    group.add_component(ns.Lens([12, -0.6, 0], focal_length=5.0, radius=0.5))
    lossy = LossyMirror([16, 0, 0], radius=3.0).RotZ(np.pi + 0.02)
    gate = ns.Mirror([-2, 0, 0], radius=3.0, max_interact_count=1)
    absorber = Absorber([6, 5, 0]).RotZ(-np.pi / 2)
    rays = [ns.Ray([0, y, z], [1, dy, 0], wavelength=wl, w0=W0)
            for wl, y, z, dy in ((633e-7, 0.0, 0.0, 0.0), (633e-7, 0.4, 0.1, 0.01), (450e-7, -0.5, -0.2, 0.02), (850e-7, 0.7, 0.0, -0.03))]
    rays.append(ns.Ray([0, 2.5, 0], [1, 0.45, 0], wavelength=633e-7))  # past the grating, into the absorber; no q
    return dict(components=[grating, slab, group, lossy, gate, absorber], monitors=[], rays=rays, limit={"max_trace_num": 150})


def user_surface_classes(ns):
    """User-defined surfaces (surfaces.py:5-65: f / normal / within_boundary / get_bbox_local are the whole contract) and
    components carrying them, built from `ns` so that the reference and this package run the very same user code."""

    class Paraboloid(ns.Surface):
        """x = -r^2 / (4 focal): concave towards +x."""

        def __init__(self, focal, radius):
            super().__init__()
            self.planar = False
            self.focal, self.radius = focal, radius

        def f(self, P):
            return P[0] + (P[1] ** 2 + P[2] ** 2) / (4 * self.focal)

        def normal(self, P):
            n = np.array([1.0, P[1] / (2 * self.focal), P[2] / (2 * self.focal)])
            return n / np.linalg.norm(n)

        def within_boundary(self, P):
            return P[1] ** 2 + P[2] ** 2 <= self.radius**2

        def get_bbox_local(self):
            sag, R = self.radius**2 / (4 * self.focal), self.radius
            return (min(-sag, 0.0), max(-sag, 0.0), -R, R, -R, R)

    class Hyperboloid(Paraboloid):
        """x = -b (sqrt(1 + r^2 / a^2) - 1), scaled by 2.5: f need not have unit slope in x."""

        def __init__(self, a, b, radius):
            ns.Surface.__init__(self)
            self.planar = False
            self.a, self.b, self.radius = a, b, radius

        def _F(self, r2):
            return self.b * (np.sqrt(1 + r2 / self.a**2) - 1)

        def f(self, P):
            return 2.5 * (P[0] + self._F(P[1] ** 2 + P[2] ** 2))

        def normal(self, P):
            k = self.b / (self.a**2 * np.sqrt(1 + (P[1] ** 2 + P[2] ** 2) / self.a**2))
            n = np.array([1.0, k * P[1], k * P[2]])
            return n / np.linalg.norm(n)

        def get_bbox_local(self):
            sag, R = self._F(self.radius**2), self.radius
            return (-sag, 0.0, -R, R, -R, R)

    class Slit(ns.Plane):
        def __init__(self, width, height):
            super().__init__()
            self.width, self.height = width, height

        def within_boundary(self, P):
            return abs(P[1]) <= self.width / 2 and abs(P[2]) <= self.height / 2

        def get_bbox_local(self):
            return (0, 0, -self.width / 2, self.width / 2, -self.height / 2, self.height / 2)

    class Saddle(Paraboloid):  # not a surface of revolution: must be refused
        def f(self, P):
            return P[0] + (P[1] ** 2 - P[2] ** 2) / (4 * self.focal)

    class CurvedMirror(ns.BaseMirror):
        def __init__(self, origin, surface, **kwargs):
            super().__init__(origin, **kwargs)
            self.surface = surface

    class CurvedInterface(ns.BaseRefraciveSurface):
        def __init__(self, origin, surface, **kwargs):
            super().__init__(origin, **kwargs)
            self.surface = surface

    return dict(Paraboloid=Paraboloid, Hyperboloid=Hyperboloid, Slit=Slit, Saddle=Saddle, CurvedMirror=CurvedMirror,
                CurvedInterface=CurvedInterface)


def g26_user_surfaces(ns):
    """A parabolic mirror, a hyperbolic glass interface (front of a thick lens whose back is a built-in plane face), and a
    rectangular slit aperture on a partially transmitting mirror — all three surfaces user classes.  Rays parallel to the
    axis, tilted ones and one that misses the slit."""
    U = user_surface_classes(ns)
    front = U["CurvedInterface"]([6, 0, 0], U["Hyperboloid"](3.0, 1.2, 1.5), n1=1.0, n2=1.5).RotZ(np.pi)
    back = ns.CircleRefractive([6.9, 0, 0], radius=1.5, n1=1.0, n2=1.5)
    parabola = U["CurvedMirror"]([14, 0, 0], U["Paraboloid"](4.0, 2.0)).RotZ(np.pi + 0.01)
    slit = U["CurvedMirror"]([2, 0, 0], U["Slit"](1.2, 0.8), reflectivity=0.3, transmission=0.7).RotZ(0.05)
    rays = [ns.Ray([0, y, z], [1, dy, dz], wavelength=wl, w0=W0)
            for wl, y, z, dy, dz in ((633e-7, 0.0, 0.0, 0.0, 0.0), (633e-7, 0.3, 0.1, 0.0, 0.0), (633e-7, -0.45, -0.2, 0.01, 0.0),
                                     (450e-7, 0.2, 0.3, -0.02, 0.01), (850e-7, -0.1, -0.35, 0.015, -0.01), (633e-7, 0.9, 0.0, 0.0, 0.0))]
    return dict(components=[slit, front, back, parabola], monitors=[], rays=rays, limit={"max_trace_num": 40})


SCENES = {
    "g24_callables": g24_callables, "g26_user_surfaces": g26_user_surfaces,
    "g01_gaussian_beam": g01_gaussian_beam, "g02_cfg2": g02_cfg2, "g03_chromatic": g03_chromatic,
    "g04_glass_slab": g04_glass_slab, "g05_cavity": g05_cavity, "g06_mirror_pair": g06_mirror_pair,
    "g07_spherical_lenses": g07_spherical_lenses, "g08_asphere": g08_asphere, "g09_cfg5": g09_cfg5,
    "g10_cfg3": g10_cfg3, "g11_prism_refl": g11_prism_refl, "g12_dove": g12_dove,
    "g13_count_shadow": g13_count_shadow, "g15_cfg4": g15_cfg4, "g16_misc": g16_misc, "g18_fifo_gate": g18_fifo_gate,
    "g19_units_and_disorder": g19_units_and_disorder, "g21_ties": g21_ties,
}


HOOKED_SCENES = {"g25_user_components": g25_user_components}  # (the C oracle has no callbacks: reference fixture vs device only)


def interact_cases(ns):
    """Single-call API of the hot path: `component.interact(ray)` and, for leaves,
    `intersect_point_local(ray_to_local_coordinates(ray))`.  Returns [(name, component, [rays])]."""
    R = ns.Ray
    wl, w0 = 780e-7, 50e-4
    fan = lambda x0: [R([x0, y, z], [1, dy, 0.01], wavelength=wl, w0=w0) for y, z, dy in
                      ((0.0, 0.0, 0.0), (0.3, -0.1, 0.02), (-0.4, 0.2, -0.05), (2.5, 0.0, 0.0))]  # the last one misses
    dead = R([-3, 0, 0], [1, 0, 0], wavelength=wl, w0=w0, alive=False)
    inside = [R([0.0, 0.1, 0.0], [np.cos(a), np.sin(a), 0.0], wavelength=wl, w0=w0) for a in (0.2, 0.75, 1.2)]
    for r in inside:
        r._n = 1.5   # travelling in glass towards the interface: the steep one is totally reflected
    return [
        ("mirror", ns.Mirror([2, 0, 0], radius=1.0).RotZ(0.3), fan(-3) + [dead]),
        ("partial_mirror", ns.Mirror([2, 0, 0], radius=1.0, reflectivity=0.7, transmission=0.3).RotZ(-0.2), fan(-3)),
        ("lens", ns.Lens([2, 0, 0], focal_length=6.0, radius=1.0).RotY(0.1), fan(-3)),
        ("block", ns.Block([2, 0, 0], width=2, height=2), fan(-3)),
        ("interface", ns.SquareRefractive([1, 0, 0], width=4, height=4, n1=1.0, n2=1.5, reflectivity=0.1).RotZ(np.pi), inside),
        ("sphere", ns.SphereRefractive([2, 0, 0], radius=6.0, height=0.5, n1=1.5, n2=1.0).RotZ(np.pi), fan(-8)),
        ("slab_group", ns.GlassSlab([2, 0, 0], width=2, height=2, thickness=0.5, n1=1.0, n2=ns.Glass_NBK7()).RotZ(0.2), fan(-3)),
        ("asphere_group", ns.ASphericParametricLens([2, 0, 0], CT=0.6, diameter=2.4, n=1.5, R=8.0, kappa=-1, a4=1e-4), fan(-3)),
        ("prism_limited", ns.TriangularPrism([2, 0, 0], width=1.5, height=2, n1=1, n2=1.5), fan(-3)),
    ]
