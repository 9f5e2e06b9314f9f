"""The C-ABI from a client with no Python and no torch in it: examples/abi_client.c is compiled with gcc (C11) against
include/optable_hip.h, linked to liboptable_hip.so and run; device memory comes from the HIP runtime."""
import os
import shutil
import subprocess

import pytest

from optable_amd import abi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_traces_through_the_abi(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = tmp_path / "abi_client"
    libdir = os.path.dirname(abi.LIB_PATH)
    subprocess.check_call([gcc, "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "abi_client.c"),
                           "-L", libdir, "-loptable_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-lm", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
    assert "NULL field" in out.stdout
    assert "lane-per-tree kernel: yes" in out.stdout and "ray trees: 3 rays per tree" in out.stdout  # ot_trace_trees_plan / ot_trace_trees_f64 from C
