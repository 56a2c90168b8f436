"""Build the reference's own native kernel as a parity referee (THIS container only).

TEST INFRASTRUCTURE — never imported by the product (pvtrace_amd/).

Cythonizes /root/reference/pvtrace/engine/_kernel.pyx *where it lies* (the
generated C and all build products go to a temporary directory) with the flags
the reference's own build recipe uses (pvtrace/engine/build.py:19-34:
``-O3 -fopenmp``) and drops only the resulting extension module into
``oracle/_ref/``.  That directory is listed in .gitignore AND .gpurunignore:
the reference is a Python package and may not travel in any form, so the
binary is used only here, to (a) prove the C restatement in oracle/pvt_oracle.c
bit-identical to the reference kernel and (b) generate the golden fixtures
under tests/golden/ (see tests/golden/make_golden.py).

No reference source is copied into the repository.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

REF_PYX = "/root/reference/pvtrace/engine/_kernel.pyx"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")


def ref_available():
    return os.path.exists(REF_PYX)


def built_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return os.path.join(OUT_DIR, "_kernel" + suffix)


def build(force=False):
    """Returns the path of the built reference kernel, or None if the
    reference tree is not present (e.g. on the GPU box)."""
    if not ref_available():
        return None
    out = built_path()
    if os.path.exists(out) and not force:
        if os.path.getmtime(out) >= os.path.getmtime(REF_PYX):
            return out
    import numpy as np

    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="pvt_ref_") as tmp:
        c_file = os.path.join(tmp, "_kernel.c")
        subprocess.check_call(
            [sys.executable, "-m", "cython", "-3", REF_PYX, "-o", c_file]
        )
        include = sysconfig.get_paths()["include"]
        cmd = [
            "gcc", "-shared", "-fPIC", "-O3", "-fopenmp", "-w",
            "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
            "-I", include, "-I", np.get_include(),
            c_file, "-o", os.path.join(tmp, "_kernel.so"), "-lm",
        ]
        subprocess.check_call(cmd)
        shutil.copyfile(os.path.join(tmp, "_kernel.so"), out)
    return out


def load():
    """Import the reference kernel module (builds it if needed)."""
    path = build()
    if path is None:
        raise ImportError("reference tree not present; oracle/_ref unavailable")
    import importlib.util

    spec = importlib.util.spec_from_file_location("_kernel", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
