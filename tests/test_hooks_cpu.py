"""CPU: recognition of user-defined components (subclasses overriding `interact_local`, optical_component.py:235-240) by the
scene compiler — a hooked leaf keeps its place, pose, boxes and count gate in the device tables and is lowered as a surface
that ends the ray (the host asks the user's method what the hit emits: tests/test_gpu_hooks.py)."""
import numpy as np

import helpers
import optable_amd as oa
from optable_amd import abi, adapter
from optable_amd.components import BLOCK


def test_only_user_overrides_are_hooks():
    class Tweaked(oa.Mirror):  # new constructor, the class's own physics: not a hook
        def __init__(self, origin):
            super().__init__(origin, radius=2.0)

    class Custom(oa.Mirror):
        def interact_local(self, ray):
            return super().interact_local(ray)

    assert not adapter.host_hook(oa.Mirror([0, 0, 0]))
    assert not adapter.host_hook(Tweaked([0, 0, 0]))
    assert adapter.host_hook(Custom([0, 0, 0]))
    probe = Custom([0, 0, 0])
    probe._builtin_physics = True  # what table.interact_leaf_local sets on the copy it traces for super().interact_local()
    assert not adapter.host_hook(probe)
    assert adapter.lower_interaction(probe)["kind"] == oa.components.MIRROR


def test_hooked_leaves_keep_their_place_in_the_scene():
    table, sc = helpers.build("g25_user_components")
    scene = table.compile()
    assert sorted(scene.hooks) == [0, 3, 5, 7]  # grating, the group's grating, LossyMirror, Absorber (leaf order = depth first)
    nodes = [n for n in scene.nodes[: scene.n_nodes] if n.kind == abi.NODE_LEAF]
    for leaf_id, comp in scene.hooks.items():
        node = nodes[leaf_id]
        # a hooked leaf ends the ray on the device — unless its class has physics of its own under the override (LossyMirror is a
        # Mirror): then the device computes those children too, for the hook's `super().interact_local(ray)` (table._HOOK_MEMO)
        assert node.leaf_id == leaf_id and node.interaction == (oa.components.MIRROR if type(comp).__name__ == "LossyMirror" else BLOCK)
        np.testing.assert_allclose(np.array(node.origin[:]), comp.origin)
    assert scene.max_children == 2  # the slab's faces split on the device
    assert len(scene.limited) == 1


def test_user_surfaces_are_recognised_by_measurement_or_refused():
    """User `Surface` subclasses (surfaces.py:5-65): a surface of revolution x = -F(r) gets the verified series of its F (the
    device form of ASphere(R, callable)), a planar surface with the rectangle of its box as aperture the rectangle; a saddle, an
    aperture with a hole and a wrong normal are refused with what was measured."""
    import pytest
    import scenes
    from optable_amd import shapes
    from optable_amd.scene import SceneError, compile_scene

    table, sc = helpers.build("g26_user_surfaces")
    scene = table.compile()
    nodes = [n for n in scene.nodes[: scene.n_nodes] if n.kind == abi.NODE_LEAF]
    assert [n.shape for n in nodes] == [shapes.RECT, shapes.ASPHERE_CHEB, shapes.CIRCLE, shapes.ASPHERE_CHEB]
    assert not scene.hooks  # surfaces are device forms, not callbacks
    np.testing.assert_allclose(nodes[0].p[:2], [0.6, 0.4])
    np.testing.assert_allclose(nodes[1].lbox[:], sc["components"][1].surface.get_bbox_local())  # the USER's box bounds the root scan
    U = scenes.user_surface_classes(oa)
    with pytest.raises(SceneError, match="not c \\* \\(x \\+ F\\(r\\)\\)"):
        compile_scene([U["CurvedMirror"]([0, 0, 0], U["Saddle"](4.0, 2.0))])

    class Annulus(oa.Circle):
        def within_boundary(self, P):
            return 0.5 <= np.hypot(P[1], P[2]) <= self.radius

    with pytest.raises(SceneError, match="neither the disc nor the rectangle"):
        compile_scene([U["CurvedMirror"]([0, 0, 0], Annulus(1.0))])

    class Flipped(U["Paraboloid"]):
        def normal(self, P):
            return -super().normal(P)

    with pytest.raises(SceneError, match="normal differs"):
        compile_scene([U["CurvedMirror"]([0, 0, 0], Flipped(4.0, 2.0))])

    class Disc(oa.Plane):  # a user's own circle
        def within_boundary(self, P):
            return P[1] ** 2 + P[2] ** 2 <= 0.49

        def get_bbox_local(self):
            return (0, 0, -0.7, 0.7, -0.7, 0.7)

    leaf = [n for n in compile_scene([U["CurvedMirror"]([0, 0, 0], Disc())]).nodes[:1]][0]
    assert leaf.shape == shapes.CIRCLE and leaf.p[0] == pytest.approx(0.7)


def test_a_user_surface_is_measured_once_per_state():
    """The measurement (1,500 calls of the user's methods + a series fit) is remembered on the surface while its attributes
    stand — `table.ray_tracing` compiles on every call — and repeated when one changes."""
    import scenes
    from optable_amd.scene import compile_scene

    U = scenes.user_surface_classes(oa)
    surf = U["Paraboloid"](4.0, 2.0)
    calls = {"n": 0}
    inner = surf.f

    def counting_f(P):
        calls["n"] += 1
        return inner(P)

    type(surf).f = lambda self, P: counting_f(P)  # (a class attribute: instance state is what the memo is keyed on)
    mirror = U["CurvedMirror"]([0, 0, 0], surf)
    a = compile_scene([mirror])
    first = calls["n"]
    assert first > 100
    b = compile_scene([mirror])
    assert calls["n"] == first
    np.testing.assert_array_equal(np.array(a.aux[: a.n_aux]), np.array(b.aux[: b.n_aux]))
    surf.radius = 1.5
    c = compile_scene([mirror])
    assert calls["n"] > first
    assert c.nodes[0].p[0] == 1.5
