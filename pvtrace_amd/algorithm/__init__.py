"""`pvtrace.algorithm` of the reference, on the engine: `photon_tracer.follow` / `step_forward` for callers that trace
ray by ray."""
from pvtrace_amd.algorithm import photon_tracer  # noqa: F401
