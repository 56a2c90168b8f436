"""pvtrace_amd — pvtrace's Scene/Node/Geometry/Material API over an MI355X-native
photon-tracing engine (HIP, gfx950).  Only the engine hot path is implemented
natively; see DESIGN.md for scope."""
__version__ = "0.1.0"

from pvtrace_amd.common import AppError, GeometryError, TraceError
from pvtrace_amd.data import fluro_red, lumogen_f_red_305
from pvtrace_amd.geometry import Box, Cylinder, Mesh, Sphere, Transformable
from pvtrace_amd.light import (
    Event, Light, Ray, circular_mask, cube_mask, rectangular_mask,
)
from pvtrace_amd.material import (
    Absorber, CoatedSurfaceDelegate, Coating, Distribution, FresnelSurfaceDelegate,
    Luminophore, Material, NullSurfaceDelegate, Reactor, Scatterer, Surface,
    SurfaceDelegate, cone, henyey_greenstein, isotropic, lambertian,
)
from pvtrace_amd.scene import Node, Scene
from pvtrace_amd import engine
from pvtrace_amd.device.lsc import LSC
from pvtrace_amd import spec
from pvtrace_amd.algorithm import photon_tracer
from pvtrace_amd import compat as _compat

_compat._register("pvtrace_amd")   # `pvtrace_amd.material.surface`, `pvtrace_amd.geometry.utils` ...: the reference's module paths (compat.py)
