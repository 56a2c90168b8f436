"""The reference's OWN unit-test files, run against this package (build container only: the files are read where they lie,
under /root/reference/tests, and nothing is written there).

`tests/ref_compat_plugin.py` makes `pvtrace_amd` answer to the name `pvtrace` (`pvtrace_amd.compat.install()`) before pytest
imports a test module, so `from pvtrace.geometry.utils import ...`, `from pvtrace.material.utils import ...` in the
reference's tests resolve to the product's modules.  Run here: every test file of the reference that needs neither a
third-party package the image lacks (its geometry / node / scene test files import `anytree` themselves, its mesh tests
`trimesh`) nor a GPU (`test_engine.py`, `test_scene.py`'s simulate tests and `test_3D_flux_comparison.py` are mirrored call for
call in tests/test_gpu_engine_api.py instead)."""
import os
import re
import subprocess
import sys

import pytest

REF_TESTS = "/root/reference/tests"
FILES = {   # file -> number of tests it holds
    "test_geometry_utils.py": 9,       # close_to_zero, floats_close, magnitude, norm, angle_between (3), smallest_angle_between, ray_z_cylinder
    "test_material.py": 2,
    "test_distibution.py": 2,          # Distribution.sample end points, hist=True step sampling
    "test_frensel_reflection.py": 3,   # fresnel_reflectivity, specular_reflection
    "test_frensel_refraction.py": 2,   # fresnel_refraction
    "test_transformable.py": 1,
}


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference tree not present (build container only)")
def test_the_references_own_unit_tests_pass_against_this_package():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "pytest", "-p", "tests.ref_compat_plugin", "-p", "no:cacheprovider", "--import-mode=importlib",
           "--rootdir=/tmp", "-q"] + [os.path.join(REF_TESTS, f) for f in FILES]
    done = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True)
    tail = done.stdout[-1500:] + done.stderr[-500:]
    assert done.returncode == 0, tail
    assert re.search(rf"\b{sum(FILES.values())} passed\b", done.stdout), tail
    assert not os.path.exists(os.path.join(REF_TESTS, "__pycache__")) and not os.path.exists(os.path.join(REF_TESTS, ".pytest_cache"))
