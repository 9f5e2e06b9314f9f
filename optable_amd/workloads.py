"""The BASELINE configurations as synthetic workloads (scene, ray generator, seeds and draw order).

One place for what `bench.py`, the tools and the tests all trace.  The builders take the class namespace `ns`
as an argument (`optable_amd` here; `tools/make_golden.py` passes the reference package to produce the fixtures
from the very same recipe) and return plain numpy arrays, so nothing in this module touches the device.

    cfg 2  1e6 rays from the focus of Lens([5,0,0], f=5, r=1) -> MirrorPair([10,0,0], 4, 4); 5-segment cap; fp64
    cfg 3  1e7 rays through 8x4 mixed components (Mirror / Lens / GlassSlab / Prism, 56 leaves); cap 20; fp32
    cfg 4  1e7 rays x 64 wavelengths through an N-BK7 slab (dispersion), 3 segments each; fp64; 2 and 4 GPUs
    cfg 5  1e8 rays, SquareMirror + aspheric lens + 16x16 micro-mirror array (260 leaves); cap 50; fp32; 8 GPUs
"""
import numpy as np

WL, W0 = 780e-7, 61e-4  # every synthetic ray carries a Gaussian q


def cfg2_rays(n, seed=0):
    """Point source at the lens focus, cone half-angle 0.15*sqrt(u) (draw order: all u, then all phi)."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(0, 1, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    theta = 0.15 * np.sqrt(u)
    d = np.stack([np.cos(theta), np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi)], axis=1)
    return np.zeros((n, 3)), d


def cfg2_components(ns):
    return [ns.Lens([5, 0, 0], focal_length=5, radius=1.0), ns.MirrorPair([10, 0, 0], 4, 4)]


def cfg3_components(ns, slab_reflectivity=0):
    """`slab_reflectivity=0.1` is a heavy BRANCHING variant (every slab face splits the ray: trees of up to 20 segments
    through 56 leaves, generation kernels) used to measure that path; the BASELINE config has 0."""
    rng = np.random.default_rng(1)
    comps = []
    for ix in range(8):
        for iy in range(4):
            origin = [4 * (ix + 1), 3 * (iy - 1.5), 0]
            kind = ["Mirror", "Lens", "GlassSlab", "Prism"][(ix + iy) % 4]
            a = rng.uniform(-np.pi, np.pi)
            if kind == "Mirror":
                comps.append(ns.Mirror(origin, radius=1).RotZ(a))
            elif kind == "Lens":
                f = rng.uniform(4, 12)
                comps.append(ns.Lens(origin, focal_length=f, radius=1).RotZ(0.2 * a))
            elif kind == "GlassSlab":
                comps.append(ns.GlassSlab(origin, width=2, height=2, thickness=0.5, n1=1, n2=1.5, reflectivity=slab_reflectivity).RotZ(0.3 * a))
            else:
                comps.append(ns.Prism(origin, width=1.5, height=2, n1=1, n2=1.5).RotZ(a))
    return comps


def cfg3_rays(n, seed=2):
    rng = np.random.default_rng(seed)
    draws = rng.uniform(0, 1, (n, 4))
    y0 = -5.5 + 11.0 * draws[:, 0]
    z0 = -0.5 + 1.0 * draws[:, 1]
    dy = -0.05 + 0.10 * draws[:, 2]
    dz = -0.02 + 0.04 * draws[:, 3]
    o = np.stack([np.zeros(n), y0, z0], axis=1)
    d = np.stack([np.ones(n), dy, dz], axis=1)
    return o, d / np.linalg.norm(d, axis=1, keepdims=True)


CFG4_WAVELENGTHS = 64


def cfg4_components(ns, reflectivity=0):
    """`reflectivity=0.2` is the branching variant (every hit splits: ray trees, generation kernels)."""
    return [ns.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=ns.Vacuum(), n2=ns.Glass_NBK7(), reflectivity=reflectivity)]


def cfg4_rays(n_base, seed=4, n_wavelengths=CFG4_WAVELENGTHS):
    """`n_base` jittered copies of Ray([-3,2,0], 30 degrees down) x `n_wavelengths` wavelengths in the order
    `multiplex_rays_in_wavelength` produces (wavelength-major, ray.py:441-444).  Returns (origins, directions,
    wavelengths) of n_base * n_wavelengths ray-wavelength pairs."""
    rng = np.random.default_rng(seed)
    jit = rng.uniform(-0.3, 0.3, (n_base, 2))
    o = np.stack([np.full(n_base, -3.0), 2 + jit[:, 0], jit[:, 1]], 1)
    d = np.tile([np.cos(np.pi / 6), -np.sin(np.pi / 6), 0.0], (n_base, 1))
    wl = np.repeat(np.linspace(400e-7, 1100e-7, n_wavelengths), n_base)
    return np.tile(o, (n_wavelengths, 1)), np.tile(d, (n_wavelengths, 1)), wl


def cfg5_components(ns, N=(16, 16)):
    return [ns.SquareMirror([-1, 0, 0], 6, 6),
            ns.ASphericParametricLens([5, 0, 0], CT=0.8, diameter=5, n=1.5, R=10, kappa=-1, a4=1e-5),
            ns.MMA(origin=[15, 0, 0], N=N, pitch=0.2, roc=28, n=1.5, thickness=0.1, reflectivity=1, transmission=0)]


def cfg5_rays(n, seed=3):
    rng = np.random.default_rng(seed)
    yz = rng.uniform(-1.4, 1.4, (n, 2))  # y then z per ray
    o = np.concatenate([np.zeros((n, 1)), yz], axis=1)
    d = np.tile([1.0, 0.0, 0.0], (n, 1))
    return o, d


def gaussian_beam_scene(ns):
    """BASELINE configs[0], examples/gaussian_beam.py:16-44: mirror, three thin lenses, slab n = 2, mirror; 6 rays -> 13 segments.
    Returns (components, rays): the reference's own use — a handful of Ray objects through `table.ray_tracing`."""
    wl, w0 = 780e-9, 10e-6
    rays = [ns.Ray([-10, y, 0], [1, 0, 0], wavelength=wl, w0=w0) for y in (0, 2, 4, 6, 9)]
    rays.append(ns.Ray([-10, 21, 0], [1, 0, 0], wavelength=wl, w0=w0).RotZ(-np.pi / 4))
    comps = [ns.Mirror([0, 0, 0]).RotZ(np.pi / 6), ns.Lens([0, 2, 0], radius=0.8, focal_length=5),
             ns.Lens([0, 4, 0], radius=0.8, focal_length=10), ns.Lens([0, 6.5, 0], radius=0.8, focal_length=10),
             ns.GlassSlab([0, 9, 0], n1=1, n2=2, thickness=5), ns.Mirror([0, 11, 0]).RotZ(-np.pi / 2)]
    return comps, rays


def chromatic_scene(ns):
    """examples/chromatic_aberration.py:16-40: an N-BK7 slab with reflectivity 0.2 (every hit branches), three wavelengths of
    one ray (shared _id) -> 59 segments.  Returns (components, rays)."""
    r0 = [ns.Ray([-3, 2, 0], [np.cos(np.pi / 6), -np.sin(np.pi / 6), 0], wavelength=780e-7, w0=20e-4).Propagate(-2)]
    rays = ns.multiplex_rays_in_wavelength(r0, [780e-7, 560e-7, 400e-7])
    slab = ns.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=ns.Vacuum(), n2=ns.Glass_NBK7(), reflectivity=0.2)
    return [slab], rays


def user_parts_scene(ns):
    """A scene with USER-DEFINED parts, written against the public API as they would be for the reference (optical_component.py:
    235-240, surfaces.py:5-65): a three-order transmission grating (its `interact_local` is Python), a thin lens, a slab whose
    faces split, and a parabolic mirror whose surface is a user class.  Three rays, 40 traces per tree.  Returns (components, rays)."""

    class Grating(ns.OpticalComponent):
        def __init__(self, origin, radius, pitch, **kwargs):
            super().__init__(origin, **kwargs)
            self.surface, self.pitch = ns.Circle(radius), pitch

        def get_bbox_local(self):
            return self.surface.get_bbox_local()

        def interact_local(self, ray):
            P, t = self.intersect_point_local(ray)
            out = []
            for order, share in ((-1, 0.25), (0, 0.5), (1, 0.25)):
                d = np.array(ray.direction, dtype=float)
                d[1] += order * ray.wavelength / self.pitch
                d[0] = np.sign(d[0]) * np.sqrt(max(1.0 - d[1] ** 2 - d[2] ** 2, 0.0))
                out.append(ray.copy(origin=P, direction=d, intensity=ray.intensity * share,
                                    qo=None if ray.qo is None else ray.q_at_z(t), _pathlength=ray.pathlength(float(t))))
            return out

    class Paraboloid(ns.Surface):
        def __init__(self, focal, radius):
            super().__init__()
            self.planar, self.focal, self.radius = False, focal, radius

        def f(self, P):
            return P[0] + (P[1] ** 2 + P[2] ** 2) / (4 * self.focal)

        def normal(self, P):
            n = np.array([1.0, P[1] / (2 * self.focal), P[2] / (2 * self.focal)])
            return n / np.linalg.norm(n)

        def within_boundary(self, P):
            return P[1] ** 2 + P[2] ** 2 <= self.radius**2

        def get_bbox_local(self):
            R, sag = self.radius, self.radius**2 / (4 * self.focal)
            return (-sag, 0.0, -R, R, -R, R)

    class ParabolicMirror(ns.BaseMirror):
        def __init__(self, origin, focal, radius, **kwargs):
            super().__init__(origin, **kwargs)
            self.surface = Paraboloid(focal, radius)

    comps = [Grating([3, 0, 0], radius=1.0, pitch=4e-4), ns.Lens([6, 0, 0], focal_length=6.0, radius=1.5),
             ns.GlassSlab([9, 0, 0], width=3, height=3, thickness=0.4, n1=1, n2=1.5, reflectivity=0.1).RotZ(0.1),
             ParabolicMirror([12, 0, 0], focal=3.0, radius=2.0).RotZ(np.pi)]
    rays = [ns.Ray([0, y, 0], [1, 0, 0], wavelength=633e-7, w0=50e-4) for y in (-0.3, 0.0, 0.3)]
    return comps, rays


class Workload:
    """One BASELINE config: what to build, what to trace, at which total size and precision."""

    def __init__(self, name, components, rays, max_segments, precision, total_rays, scaling, label):
        self.name, self.components, self._rays = name, components, rays
        self.max_segments, self.precision = max_segments, precision
        self.total_rays, self.scaling, self.label = total_rays, scaling, label

    def rays_per_rank(self, world, override=None):
        """Rays ONE rank traces.  Weak scaling (cfg 2, cfg 3): the config's size on every GPU.  Strong scaling
        (cfg 4, cfg 5: the config names the total, sharded over the GPUs): total / world, contiguous shards
        (optable_amd.dist.shard_range)."""
        if override:
            return int(override)
        return self.total_rays if self.scaling == "weak" else -(-self.total_rays // world)

    def rays(self, n, rank=0):
        """(origins, directions, wavelengths) of this rank's shard: `n` rays from the config's generator with the
        config's seed offset by the rank (independent shards of one synthetic distribution)."""
        return self._rays(n, rank)


def _with_wl(gen, base_seed):
    return lambda n, rank: gen(n, base_seed + rank) + (WL,)


def _cfg4_shard(n, rank):
    o, d, wl = cfg4_rays(max(n // CFG4_WAVELENGTHS, 1), 4 + rank)
    return o, d, wl


def baseline_workloads(ns):
    return {
        "cfg2": Workload("cfg2", lambda: cfg2_components(ns), _with_wl(cfg2_rays, 0), 5, "f64", 1_000_000, "weak",
                         "cfg2: 1e6 point-source rays -> Lens + MirrorPair (S=3 leaves), 5-segment cap, fp64"),
        "cfg3": Workload("cfg3", lambda: cfg3_components(ns), _with_wl(cfg3_rays, 2), 20, "f32", 10_000_000, "weak",
                         "cfg3: 1e7 rays, 32 mixed components (S=56 leaves), 20-segment cap, fp32"),
        "cfg4": Workload("cfg4", lambda: cfg4_components(ns), _cfg4_shard, 3, "f64", 640_000_000, "strong",
                         "cfg4: 1e7 rays x 64 wavelengths = 6.4e8 ray-wavelength pairs through an N-BK7 slab, fp64, "
                         "sharded over the GPUs"),
        "cfg5": Workload("cfg5", lambda: cfg5_components(ns), _with_wl(cfg5_rays, 3), 50, "f32", 100_000_000, "strong",
                         "cfg5: 1e8 rays, asphere + MMA 16x16 (S=260 leaves), 50-segment cap, fp32, sharded over the GPUs"),
    }
