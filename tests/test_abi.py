"""CPU: the C-ABI library loads and exports every symbol include/optable_hip.h declares; the
product fails loudly without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from optable_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "optable_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ot_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(abi.SYMBOLS)


def test_library_exports_every_symbol():
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = abi.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.ot_abi_version() == abi.ABI_VERSION


def test_struct_sizes_match_header():
    # 36 doubles + 12 int32 / 7 doubles + 2 int32, no padding surprises
    assert C.sizeof(abi.OtNode) == 36 * 8 + 12 * 4
    assert C.sizeof(abi.OtMaterial) == 7 * 8 + 8
    assert C.sizeof(abi.OtRays) == 15 * 8
    assert C.sizeof(abi.OtSegments) == 14 * 8
    assert C.sizeof(abi.OtMonitor) == 14 * 8


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import optable_amd as oa

    table = oa.OpticalTable()
    table.add_components([oa.Mirror([1, 0, 0])])
    with pytest.raises((abi.EngineUnavailable, RuntimeError)):
        table.ray_tracing([oa.Ray([0, 0, 0], [1, 0, 0])])
    lib = abi.load()
    ctx = C.c_void_p()
    rc = lib.ot_ctx_create(0, None, C.byref(ctx))
    assert rc < 0 and lib.ot_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "optable_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|oracle/|libot_oracle", src, flags=re.M), f



def test_frame_transforms_are_host_logic():
    """ray_to_local_coordinates / ray_to_lab_coordinates (optical_component.py:106-124): plain rotations of a
    single Ray object, round trip exact to rounding; no GPU involved."""
    import numpy as np
    import optable_amd as oa

    comp = oa.Mirror([2.0, -1.0, 0.5], radius=1.0).RotZ(0.7).RotY(-0.3)
    ray = oa.Ray([0.3, 0.2, -0.1], [1.0, 0.2, -0.05], wavelength=780e-7, w0=50e-4)
    local = comp.ray_to_local_coordinates(ray)
    M = comp.transform_matrix
    np.testing.assert_allclose(local.origin, M.T @ (ray.origin - comp.origin), atol=1e-15)
    np.testing.assert_allclose(local.direction, M.T @ ray.direction, atol=1e-15)
    back = comp.ray_to_lab_coordinates(local)
    np.testing.assert_allclose(back.origin, ray.origin, atol=1e-14)
    np.testing.assert_allclose(back.direction, ray.direction, atol=1e-14)
    assert back._id == ray._id and back.qo == ray.qo and local is not ray


def test_out_of_scope_methods_say_so():
    """Rendering and component-metadata methods of the reference are not rebuilt; calling one raises a
    NotImplementedError that names the way out (keep the original package + install())."""
    import pytest
    import optable_amd as oa

    table = oa.OpticalTable()
    for obj, name in ((table, "render"), (table, "gather_components"), (table, "export_components_csv"),
                      (oa.Mirror([0, 0, 0]), "render"), (oa.Ray([0, 0, 0], [1, 0, 0]), "render"), (oa.Monitor([0, 0, 0], 1, 1), "render")):
        with pytest.raises(NotImplementedError, match="install"):
            getattr(obj, name)()


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure and the reference cannot travel: nothing under optable_amd/ (Python or
    HIP sources), nor the build recipe, may import, link or mention either as a code path."""
    import os
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "optable_amd")
    offenders = []
    for folder, _, files in os.walk(root):
        for name in files:
            if not name.endswith((".py", ".h", ".hip", "Makefile")) and name != "Makefile":
                continue
            text = open(os.path.join(folder, name), errors="ignore").read()
            if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "ot_oracle" in text or "libot_oracle" in text:
                offenders.append(name + ": oracle")
            if re.search(r"sys\.path\.(insert|append)\([^)]*reference", text) or re.search(r"^\s*import\s+optable\s*$", text, re.M):
                offenders.append(name + ": reference import")
    assert not offenders, offenders


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """Without the built .so there is nothing to run: load() names the missing file and the build command."""
    monkeypatch.setattr(abi, "LIB_PATH", str(tmp_path / "liboptable_hip.so"))
    monkeypatch.setattr(abi, "_lib", None)
    with pytest.raises(abi.EngineUnavailable, match="no CPU fallback"):
        abi.load()


def test_library_binds_to_the_hip_runtime_that_is_already_in_the_process():
    """One HIP runtime per process: the library names libamdhip64.so.7 by soname, abi.load() brings torch's
    copy in first, so that copy is the one the library's symbols resolve to — and it is the only one mapped."""
    import subprocess

    from optable_amd import abi

    path, version = abi.runtime_info()
    copies = abi.hip_runtimes_in_process()
    assert len(copies) == 1 and os.path.realpath(path) in copies, (path, copies)
    assert version > 0
    needed = subprocess.check_output(["readelf", "-d", abi.LIB_PATH], text=True)
    assert "Shared library: [libamdhip64.so.7]" in needed  # by soname: no absolute path baked in
