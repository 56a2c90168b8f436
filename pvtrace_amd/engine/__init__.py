"""MI355X photon-tracing engine behind pvtrace's `engine` API.

`simulate(scene, num_rays, ...)` flattens the scene graph to SoA tables,
uploads them once, and runs the per-photon loop as a HIP kernel on gfx950
(see DESIGN.md).  Public names match the reference's pvtrace/engine/__init__.py.
"""
from pvtrace_amd.engine.compiler import CompiledScene, UnsupportedSceneError, compile_scene
from pvtrace_amd.engine.recorder import Heatmap, Histogram, Recorder
from pvtrace_amd.engine.tally import tally_histories
from pvtrace_amd.engine.instrument import auto_recorders, instrument, recorders_from_spec
from pvtrace_amd.engine.native import EngineUnavailableError
from pvtrace_amd.engine.pipeline import BundlePipeline, trace_stream
from pvtrace_amd.engine.api import (
    EngineResult,
    RecorderResult,
    is_available,
    simulate,
    simulate_stream,
    Session,
    release_resident_scenes,
)

__all__ = [
    "CompiledScene", "UnsupportedSceneError", "compile_scene", "Recorder", "Histogram",
    "Heatmap", "EngineResult", "RecorderResult", "EngineUnavailableError", "is_available",
    "simulate", "simulate_stream", "Session", "tally_histories", "BundlePipeline", "trace_stream", "auto_recorders", "instrument", "recorders_from_spec", "release_resident_scenes",
]
