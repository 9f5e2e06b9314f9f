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
        assert node.leaf_id == leaf_id and node.interaction == BLOCK
        np.testing.assert_allclose(np.array(node.origin[:]), comp.origin)
    assert scene.max_children == 2  # the slab's faces split on the device
    assert len(scene.limited) == 1
