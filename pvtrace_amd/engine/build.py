"""`python -m pvtrace_amd.engine.build` -- the counterpart of the reference's `python -m pvtrace.engine.build`
(pvtrace/engine/build.py: compiles its Cython kernel in place): compiles the HIP library for gfx950 in place
(`pvtrace_amd/csrc/libpvtrace_hip.so`) with the project's flags.  The recipe itself lives in `__graft_entry__.build()` at
the repository root, next to the package; hipcc cross-compiles, so no GPU is needed to build."""
import importlib.util
import os
import sys


def main(force=False):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    recipe = os.path.join(root, "__graft_entry__.py")
    if not os.path.exists(recipe):
        raise SystemExit(f"build recipe not found next to the package ({recipe}); this is an in-tree build")
    spec = importlib.util.spec_from_file_location("__graft_entry__", recipe)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    module.build(force=force)
    print("built", module.LIB)


if __name__ == "__main__":
    main(force="--force" in sys.argv[1:])
