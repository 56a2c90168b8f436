"""The scripts under examples/ run to the end on the GPU box (each in its own interpreter)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,says", [("hello_world.py", "first recorded history"), ("lsc.py", "Optical Efficiency"), ("mesh_gem.py", None)])
def test_example_runs(script, says):
    done = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], capture_output=True, text=True, cwd=ROOT,
                          timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    if says:
        assert says in done.stdout, done.stdout[-1500:]
