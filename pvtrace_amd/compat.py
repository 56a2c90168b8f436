"""Module paths of the reference, for code written against it.

The reference is a package of packages (`pvtrace.material.surface`, `pvtrace.geometry.utils`, `pvtrace.light.light` ...);
this package keeps the same names in fewer files (`pvtrace_amd.material`, `.geometry`, `.light`, `.scene`).  `LAYOUT` maps
every module path of the reference that belongs to the traced path and its callers onto the module here that holds its names,
and the aliases are registered so that both spellings import:

    from pvtrace_amd.material.surface import FresnelSurfaceDelegate      # always (registered when pvtrace_amd is imported)

    import pvtrace_amd.compat; pvtrace_amd.compat.install()              # opt-in: the package answers to `pvtrace` too
    from pvtrace import *                                                # ... and a script written for the reference runs
    from pvtrace.geometry.utils import flip, angle_between               #     on the engine unchanged

What is not there is not aliased: `pvtrace.scene.renderer` (meshcat), `pvtrace.cli.main` and the rest of the command-line tool
(`pvtrace.cli.parse.parse`, the scene-spec reader, is), `pvtrace.studio` raise ImportError as any missing module does."""
import importlib
import sys

# reference module path (below the package) -> module here (below pvtrace_amd)
LAYOUT = {
    "algorithm": "algorithm", "algorithm.photon_tracer": "algorithm.photon_tracer",
    "cli": "spec", "cli.parse": "spec",   # (only the scene-spec parser, `parse(filename) -> Scene`; the command-line tool itself is not here)
    "common": "common", "common.errors": "common",
    "data": "data", "data.lumogen_f_red_305": "data.lumogen_f_red_305", "data.fluro_red": "data.fluro_red",
    "device": "device", "device.lsc": "device.lsc",
    "engine": "engine", "engine.api": "engine.api", "engine.compiler": "engine.compiler", "engine.emit": "engine.emit",
    "engine.recorder": "engine.recorder", "engine.tally": "engine.tally", "engine.build": "engine.build",
    "geometry": "geometry", "geometry.geometry": "geometry", "geometry.box": "geometry", "geometry.sphere": "geometry",
    "geometry.cylinder": "geometry", "geometry.mesh": "geometry", "geometry.utils": "geometry",
    "geometry.transformable": "geometry", "geometry.transformations": "geometry", "geometry.intersection": "scene",
    "light": "light", "light.light": "light", "light.ray": "light", "light.event": "light",
    "material": "material", "material.material": "material", "material.component": "material",
    "material.distribution": "material", "material.surface": "material", "material.utils": "material",
    "scene": "scene", "scene.node": "scene", "scene.scene": "scene",
}


def _register(prefix):
    for theirs, ours in LAYOUT.items():
        sys.modules.setdefault(f"{prefix}.{theirs}", importlib.import_module(f"pvtrace_amd.{ours}"))


def install(name="pvtrace", force=False):
    """Make this package importable as `name` with the reference's module layout.  Refuses when another package of that
    name is already imported or importable (the reference itself, say) unless `force`."""
    import importlib.util

    import pvtrace_amd

    present = sys.modules.get(name)
    if present is not None and present is not pvtrace_amd and not force:
        raise ImportError(f"a different package is already imported as {name!r}")
    if present is None and not force:
        try:
            found = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            found = None
        if found is not None:
            raise ImportError(f"{name!r} is importable from {found.origin}; pass force=True to shadow it")
    sys.modules[name] = pvtrace_amd
    _register(name)
    return pvtrace_amd
