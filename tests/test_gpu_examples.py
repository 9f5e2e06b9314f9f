"""The example scripts run as a user would run them."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,needle", [("gaussian_beam.py", "6 rays -> 13 segments"), ("million_rays.py", "5000000 segments"),
                                            ("user_components.py", "hooks: ['Grating']")])
def test_example_runs(script, needle):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert needle in out.stdout, out.stdout
