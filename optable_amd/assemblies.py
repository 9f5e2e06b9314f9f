"""Rigid assemblies of leaf components (reference: optable/component_group.py).

A group is scene-compiler input: the compiler walks `.components` depth first and emits one
`ot_node` per group (lab AABB + index one past its last descendant) followed by its children,
so the kernel's skip-list traversal reproduces `ComponentGroup.interact`'s AABB prune and
first-minimum selection (component_group.py:93-122).  The factories below only place leaves;
their constructor signatures match the reference so existing scripts keep working.
"""
from typing import Callable, Union

import numpy as np

from .geometry import pivot_origin
from .components import (OpticalComponent, BaseRefraciveSurface, SquareMirror, SquareRefractive,
                         CircleRefractive, SphereRefractive, Lens)
from .shapes import Polygon, ASphere, sag_parametric, sag_exact
from .slab import solve_normal_to_normal_rotation

_Z = [0, 0, 1]


class ComponentGroup(OpticalComponent):
    def __init__(self, origin, **kwargs):
        super().__init__(origin, **kwargs)
        self._bboxes = []
        self.components = []
        self.monitors = []
        self.rays = []
        self.refpoints = []

    def __repr__(self):
        return f"ComponentGroup(origin={self.origin}, transform_matrix={self.transform_matrix})"

    # -- boxes (cached like the leaves') ----------------------------------------------------
    @property
    def bboxes(self):
        if not self._bboxes:
            self.get_bboxes()
        return self._bboxes

    def get_bboxes(self):
        self._bboxes = [c.bbox for c in self.components]
        return self._bboxes

    def get_bbox(self) -> tuple:
        return self.surface.merge_bboxs(self.get_bboxes())

    # -- rigid motion of the whole assembly (component_group.py:49-85) ---------------------
    def _members(self):
        return [*self.rays, *self.components, *self.monitors, *self.refpoints]

    def _RotAroundLocal(self, axis, localpoint, theta):
        lp = np.array(localpoint)
        pivot = self.origin + lp
        rot = self.R(axis, theta)
        self.transform_matrix = rot @ self.transform_matrix
        self.origin = pivot_origin(self.origin, rot, lp)
        for member in self._members():
            member._RotAroundLocal(axis, pivot - member.origin, theta)
        return self

    def _RotAroundCenter(self, axis, theta):
        return self._RotAroundLocal(axis, [0, 0, 0], theta)

    def _Translate(self, movement):
        self.origin += np.array(movement)
        for member in self._members():
            member._Translate(movement)
        return self

    # -- membership ---------------------------------------------------------------------
    def add_rays(self, rays):
        self.rays.extend(rays)

    def add_component(self, component):
        self.components.append(component)
        if hasattr(component, "rays"):
            self.rays.extend(component.rays)

    def add_components(self, components):
        for c in components:
            self.add_component(c)

    def add_monitor(self, monitor):
        self.monitors.append(monitor)

    def add_monitors(self, monitors):
        self.monitors.extend(monitors)

    def add_refpoint(self, point):
        self.refpoints.append(point)


def _at(origin, dx=0.0, dy=0.0, dz=0.0):
    return origin + np.array([dx, dy, dz])


class GlassSlab(ComponentGroup):
    """Two parallel rectangular faces `thickness` apart, front face at the origin (:148-185)."""

    def __init__(self, origin, width=1.0, height=1.0, thickness=1.0, n1=1.0, n2=1.5,
                 reflectivity=0, transmission=1, **kwargs):
        super().__init__(origin, **kwargs)
        for shift, (outer, inner) in ((0, (n1, n2)), (-thickness, (n2, n1))):
            self.add_component(SquareRefractive(_at(origin, shift), width, height, outer, inner,
                                                reflectivity=reflectivity, transmission=transmission, **kwargs))


class CircleGlassSlab(ComponentGroup):
    def __init__(self, origin, radius=1.0, thickness=1.0, n1=1.0, n2=1.5, reflectivity1=0,
                 transmission1=1, reflectivity2=0, transmission2=1, **kwargs):
        super().__init__(origin, **kwargs)
        self.radius = radius
        faces = ((0, n1, n2, reflectivity1, transmission1), (-thickness, n2, n1, reflectivity2, transmission2))
        for shift, na, nb, refl, trans in faces:
            self.add_component(CircleRefractive(_at(origin, shift), radius, na, nb,
                                                reflectivity=refl, transmission=trans, **kwargs))


def _lattice(N, pitch_y, pitch_z):
    """(i, j, y, z) of an ny x nz grid centred on the origin, z-major (component_group.py:239-243)."""
    ny, nz = N
    for i in range(nz):
        for j in range(ny):
            yield i, j, (j - (ny - 1) / 2) * pitch_y, (i - (nz - 1) / 2) * pitch_z


class MLA(ComponentGroup):
    """Micro-lens array of thin lenses (:228-246)."""

    def __init__(self, origin, N, pitch, focal_length, radius, focal_drift=0, **kwargs):
        super().__init__(origin)
        self.pitch, self.focal_length, self.radius = pitch, focal_length, radius
        if isinstance(N, int):
            N = (N, 1)
        for _, _, y, z in _lattice(N, pitch, pitch):
            f = focal_length * (1 + focal_drift * np.random.randn())
            self.add_component(Lens(origin=_at(self.origin, 0, y, z), focal_length=f, radius=radius, **kwargs))


def _cap_height(roc, aperture):
    return roc - np.sqrt(roc**2 - (aperture / 2) ** 2)


class MMA(ComponentGroup):
    """Micro-mirror array: spherical caps on the front of a glass plate plus the flat back
    face (:249-304)."""

    def __init__(self, origin, N, pitch, roc, n, thickness, roc_drift=0, **kwargs):
        super().__init__(origin, **kwargs)
        self.pitch = pitch
        if not isinstance(N, tuple):
            N = (N, 1)
        ny, nz = N
        py, pz = pitch if isinstance(pitch, tuple) else (pitch, pitch)
        if isinstance(roc, tuple):
            roc = np.array(roc)
            assert roc.shape == (nz, ny), f"roc shape {roc.shape} does not match ({nz}, {ny})"
        else:
            roc = np.ones((nz, ny)) * roc
        shear_y, shear_z = kwargs.get("shifty_z", 0), kwargs.get("shiftz_y", 0)
        off_y, off_z = kwargs.get("mma_shifty", 0), kwargs.get("mma_shiftz", 0)
        for i, j, y0, z0 in _lattice((ny, nz), py, pz):
            y = y0 + i * shear_y + off_y
            z = z0 + j * shear_z + off_z
            r = roc[i, j] * (1 + roc_drift * np.random.randn())
            self.add_component(SphereRefractive(origin=_at(self.origin, -r, y, z), radius=r,
                                                height=_cap_height(r, self.pitch), n1=n, n2=1.0, **kwargs))
        self.add_component(SquareRefractive(
            origin=_at(self.origin, thickness), width=kwargs.get("mma_width", ny * py),
            height=kwargs.get("mma_height", nz * pz), n1=1, n2=n,
            reflectivity=kwargs.get("back_reflectivity", 0), transmission=kwargs.get("back_transmission", 1)))


class MMADisordered(ComponentGroup):
    """MMA with caps at arbitrary points and optional per-cap normals (:307-364)."""

    def __init__(self, origin, PList, pitch, roc, n, thickness, roc_drift=0, nList=None, **kwargs):
        super().__init__(origin, **kwargs)
        self.pitch = pitch
        PList = np.array(PList)
        assert PList.shape[1] == 3, "PList must be a list of 3D points"
        if nList is not None:
            nList = np.array(nList)
            assert nList.shape[0] == PList.shape[0], "nList must match PList length"
        if isinstance(roc, (int, float)):
            roc = np.ones(PList.shape[0]) * roc
        else:
            roc = np.array(roc)
            assert roc.shape[0] == PList.shape[0], "roc must match PList length"
        for k, apex in enumerate(PList):
            r = roc[k] * (1 + roc_drift * np.random.randn())
            cap = SphereRefractive(origin=np.array(apex) + self.origin + [-r, 0, 0], radius=r,
                                   height=_cap_height(r, self.pitch), n1=n, n2=1.0, **kwargs)
            if nList is not None:
                axis, theta = solve_normal_to_normal_rotation(cap.normal, -nList[k])
                cap._RotAroundLocal(axis, [r, 0, 0], theta)
            self.add_component(cap)
        span = PList.max(axis=0) - PList.min(axis=0)
        self.add_component(SquareRefractive(origin=_at(self.origin, thickness), width=span[1] * 1.1,
                                            height=span[2] * 1.1, n1=1, n2=n, reflectivity=0, transmission=1))


class DMD(ComponentGroup):
    """Grid of tilted square mirrors (:367-391)."""

    def __init__(self, origin, N, pitch, tilt_angle=np.pi / 4, **kwargs):
        super().__init__(origin)
        self.pitch, self.tilt_angle = pitch, tilt_angle
        if isinstance(N, int):
            N = (N, 1)
        for _, _, y, z in _lattice(N, pitch, pitch):
            self.add_component(SquareMirror(origin=_at(self.origin, 0, y, z), width=pitch, height=pitch,
                                            reflectivity=1.0, **kwargs).RotZ(tilt_angle))


class WedgePlate(ComponentGroup):
    def __init__(self, origin, width=1.0, height=1.0, thickness=1.0, wedge_angle=0.0, n1=1.0, n2=1.5,
                 reflectivity=0, transmission=1, **kwargs):
        super().__init__(origin, **kwargs)
        for sign, (na, nb) in ((+1, (n1, n2)), (-1, (n2, n1))):
            face = SquareRefractive(_at(origin, sign * thickness / 2), width, height, na, nb,
                                    reflectivity=reflectivity, transmission=transmission, **kwargs)
            self.add_component(face.RotZ(sign * wedge_angle / 2))


def _roof_pair(make_face, origin, width, angle):
    """Two faces of length `width` meeting at `origin` with included angle `angle`, symmetric
    about the x axis; each is built centred at y = +-width/2 and swung about the apex
    (component_group.py:475-497, 557-583, 728-747)."""
    swing = (np.pi - angle) / 2
    for sign in (+1, -1):
        yield make_face(_at(origin, 0, sign * width / 2), sign)._RotAroundLocal(_Z, [0, -sign * width / 2, 0], sign * swing)


class MirrorPair(ComponentGroup):
    def __init__(self, origin, width=1.0, height=1.0, angle: float = np.pi / 2, reflectivity_1=1,
                 transmission_1=0, reflectivity_2=1, transmission_2=0, **kwargs):
        super().__init__(origin, **kwargs)
        self.name = kwargs.get("name", self.__class__.__name__)
        coeff = {+1: (reflectivity_1, transmission_1, 1), -1: (reflectivity_2, transmission_2, 2)}

        def face(o, sign):
            refl, trans, idx = coeff[sign]
            return SquareMirror(o, width=width, height=height, reflectivity=refl, transmission=trans,
                                **{**kwargs, "name": f"{self.name} mirror {idx}"})

        for m in _roof_pair(face, origin, width, angle):
            self.add_component(m)


class Prism(ComponentGroup):
    """Isosceles prism, apex at the origin pointing +x, hypotenuse behind it (:500-597)."""

    def __init__(self, origin, width=1.0, height=1.0, n1=1.0, n2=1.5, angle: float = np.pi / 2,
                 reflectivity_leg=0, transmission_leg=1, reflectivity_hyp=0, transmission_hyp=1, **kwargs):
        super().__init__(origin, **kwargs)
        self.name = kwargs.get("name", self.__class__.__name__)

        def leg(o, sign):
            return SquareRefractive(o, width, height, n1, n2, reflectivity=reflectivity_leg,
                                    transmission=transmission_leg,
                                    **{**kwargs, "name": f"{self.name} leg {1 if sign > 0 else 2}"})

        for f in _roof_pair(leg, origin, width, angle):
            self.add_component(f)
        self.add_component(SquareRefractive(
            _at(origin, -width * np.cos(angle / 2)), width * 2 * np.sin(angle / 2), height, n2, n1,
            reflectivity=reflectivity_hyp, transmission=transmission_hyp,
            **{**kwargs, "name": f"{self.name} hypotenuse"}))


class TriangularPrism(ComponentGroup):
    """Entrance face 1 on the y axis from the origin up to `width`; faces 2 and 3 leave its
    top and bottom edges at interior angles alpha and beta (:600-712)."""

    def __init__(self, origin, width=1.0, height=1.0, n1=1.0, n2=1.5, alpha=np.pi / 4, beta=np.pi / 2,
                 reflectivity_1=0, reflectivity_2=0, reflectivity_3=0, transmission_1=1, transmission_2=1,
                 transmission_3=1, max_interact_count_2=5, max_interact_count_3=5, **kwargs):
        super().__init__(origin, **kwargs)
        self.name = kwargs.get("name", self.__class__.__name__)
        apex = np.sin(np.pi - alpha - beta)
        len2 = width * np.sin(beta) / apex
        len3 = width * np.sin(alpha) / apex
        self.add_component(SquareRefractive(origin=_at(origin, 0, width / 2), width=width, height=height,
                                            n1=n1, n2=n2, reflectivity=reflectivity_1,
                                            transmission=transmission_1, **kwargs))
        self.add_component(SquareRefractive(origin=_at(origin, 0, width - len2 / 2), width=len2, height=height,
                                            n1=n2, n2=n1, reflectivity=reflectivity_2, transmission=transmission_2,
                                            max_interact_count=max_interact_count_2, **kwargs)
                           ._RotAroundLocal(_Z, [0, len2 / 2, 0], -alpha))
        self.add_component(SquareRefractive(origin=_at(origin, 0, len3 / 2), width=len3, height=height,
                                            n1=n2, n2=n1, reflectivity=reflectivity_3, transmission=transmission_3,
                                            max_interact_count=max_interact_count_3, **kwargs)
                           ._RotAroundLocal(_Z, [0, -len3 / 2, 0], beta))


class MirrorPrism(ComponentGroup):
    def __init__(self, origin, width=1.0, height=1.0, angle: float = np.pi / 2, reflectivity=1.0,
                 transmission=0.0, **kwargs):
        super().__init__(origin, **kwargs)

        def face(o, sign):
            return SquareMirror(o, width, height, reflectivity=reflectivity, transmission=transmission, **kwargs)

        for m in _roof_pair(face, origin, width, angle):
            self.add_component(m)


class MirrorCube(ComponentGroup):
    """Corner cube whose diagonal lies along x (:750-783)."""

    def __init__(self, origin, L=1.0, reflectivity=1.0, **kwargs):
        super().__init__(origin, **kwargs)
        h = L / 2
        faces = [
            SquareMirror(_at(self.origin, 0.0, h, h), width=L, height=L, reflectivity=reflectivity, **kwargs),
            SquareMirror(_at(self.origin, h, 0.0, h), width=L, height=L, reflectivity=reflectivity, **kwargs).RotZ(np.pi / 2),
            SquareMirror(_at(self.origin, h, h, 0.0), width=L, height=L, reflectivity=reflectivity, **kwargs).RotY(-np.pi / 2),
        ]
        self.add_components(faces)
        self._RotAroundLocal([0, 0, 1], [0, 0, 0], -np.pi / 4)
        self._RotAroundLocal([0, 1, 0], [0, 0, 0], np.arccos(np.sqrt(2 / 3)))


class DovePrism(ComponentGroup):
    """Dove prism of base length L, height D, index Ng; faces are absolute-positioned
    (the `origin` argument only labels the group, as upstream :786-850)."""

    def __init__(self, origin, L, D, Ng, **kwargs):
        super().__init__(origin, **kwargs)
        self.L, self.D, self.Ng = L, D, Ng
        top = L - 2 * D
        trapezoid = Polygon(np.array([[-L / 2, 0], [L / 2, 0], [top / 2, D], [-top / 2, D]]))

        def polygon_face(o, na, nb, poly):
            face = BaseRefraciveSurface(origin=o, n1=na, n2=nb)
            face.surface = poly
            return face

        def slanted(sign):
            return Polygon(np.array([[-D / 2, sign * L / 2, 0], [D / 2, sign * L / 2, 0],
                                     [D / 2, sign * top / 2, D], [-D / 2, sign * top / 2, D]]))

        side_l = polygon_face([-D / 2, 0, 0], Ng, 1, trapezoid)
        side_r = polygon_face([D / 2, 0, 0], 1, Ng, trapezoid)
        roof = SquareRefractive(origin=[0, 0, D], width=top, height=D, n1=Ng, n2=1).RotY(np.pi / 2)
        base = SquareRefractive(origin=[0, 0, 0], width=L, height=D, n1=1, n2=Ng).RotY(np.pi / 2)
        face_in = polygon_face([0, 0, 0], Ng, 1, slanted(-1))
        face_out = polygon_face([0, 0, 0], 1, Ng, slanted(+1))
        self.add_components([side_l, side_r, roof, base, face_in, face_out])

    @property
    def z0(self):
        """Height at which a ray passes straight through (:846-850)."""
        theta = np.arcsin((np.sqrt(2) / 2) / self.Ng)
        return (self.L / 2) / (1 + np.tan(np.pi / 4 + theta))


def _spherical_face(vertex_x, R, diameter, n_before, n_after, origin, kwargs):
    """One spherical refracting face whose vertex sits at x = vertex_x on the axis, as seen by a
    beam travelling +x: R > 0 bulges towards the incoming beam.  The cap's centre of curvature
    is at vertex_x + R either way; a convex-to-the-beam cap is built facing +x and turned by pi
    (component_group.py:901-935, 950-1007)."""
    centre = np.array([vertex_x + R, 0, 0]) + origin
    r = abs(R)
    h = _cap_height(r, diameter)
    if R > 0:
        return SphereRefractive(origin=centre, radius=r, height=h, n1=n_before, n2=n_after, **kwargs).RotZ(np.pi)
    return SphereRefractive(origin=centre, radius=r, height=h, n1=n_after, n2=n_before, **kwargs)


class PlanoConvexLens(ComponentGroup):
    def __init__(self, origin, EFL, CT, diameter, R, **kwargs):
        super().__init__(origin, **kwargs)
        n = 1 + R / EFL
        principal = CT / n  # principal plane measured from the flat face
        self.add_component(SphereRefractive(origin=_at(self.origin, R - (CT - principal)), radius=R,
                                            height=_cap_height(R, diameter), n1=1.0, n2=n, **kwargs).RotZ(np.pi))
        self.add_component(CircleRefractive(origin=_at(self.origin, principal), radius=diameter / 2,
                                            n1=n, n2=1.0, **kwargs).RotZ(np.pi))


class BiConvexLens(ComponentGroup):
    def __init__(self, origin, CT, R1, R2, diameter, EFL=None, n=None, **kwargs):
        super().__init__(origin, **kwargs)
        if n is None:
            assert EFL is not None, "Either n or EFL must be provided"
            n = 1 + (1 / EFL) / (1 / R1 - 1 / R2)
            for _ in range(3):  # thick-lens refinement of the lensmaker estimate
                n = 1 + (1 / EFL) / (1 / R1 - 1 / R2 + (n - 1) * CT / (n * R1 * R2))
        else:
            assert isinstance(n, (int, float)), "n must be a number"
        self.add_component(_spherical_face(0, R1, diameter, 1.0, n, self.origin, kwargs))
        self.add_component(_spherical_face(CT, R2, diameter, n, 1.0, self.origin, kwargs))


class Doublet(ComponentGroup):
    def __init__(self, origin, CT1, CT2, R1, R2, R3, diameter, n12, n23, **kwargs):
        super().__init__(origin, **kwargs)
        self.add_component(_spherical_face(0, R1, diameter, 1.0, n12, self.origin, kwargs))
        self.add_component(_spherical_face(CT1, R2, diameter, n12, n23, self.origin, kwargs))
        self.add_component(_spherical_face(CT1 + CT2, R3, diameter, n23, 1.0, self.origin, kwargs))


class ASphericLens(ComponentGroup):
    """Aspheric front face x = -F1(r) (after a pi turn: bulging towards -x) and an aspheric or
    flat back face CT behind it (:1014-1055)."""

    def __init__(self, origin, CT, f_asphere_1: Callable, f_asphere_2: Union[Callable, None], diameter, n, **kwargs):
        super().__init__(origin, **kwargs)
        self.f_asphere_1, self.f_asphere_2 = f_asphere_1, f_asphere_2
        half = diameter / 2
        self.add_component(BaseRefraciveSurface(origin=self.origin, n1=1, n2=n,
                                                surface=ASphere(half, f_asphere_1), **kwargs).RotZ(np.pi))
        back_at = _at(self.origin, CT)
        if f_asphere_2 is not None:
            back = BaseRefraciveSurface(origin=back_at, n1=n, n2=1.0, surface=ASphere(half, f_asphere_2), **kwargs)
        else:
            back = CircleRefractive(origin=back_at, radius=half, n1=n, n2=1.0, **kwargs)
        self.add_component(back.RotZ(np.pi))


class ASphericExactSphericalLens(ASphericLens):
    def __init__(self, origin, EFL, CT, diameter, n, **kwargs):
        super().__init__(origin, CT, sag_exact(EFL, n), None, diameter, n, **kwargs)


class ASphericParametricLens(ASphericLens):
    def __init__(self, origin, CT, diameter, n, R, kappa, a4=0, a6=0, a8=0, **kwargs):
        super().__init__(origin, CT, sag_parametric(R, kappa, a4, a6, a8), None, diameter, n, **kwargs)
