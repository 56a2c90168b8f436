"""Light sources, rays and events.

API mirror of the reference's pvtrace/light/light.py:48-262, ray.py:16-95 and
event.py:4-16.  A `Light` is three delegates (wavelength, position, direction)
sampled in the light node's frame.  The engine recognises the built-in
delegates below — including ``functools.partial(cone, theta)`` and friends,
which the reference's vectorised emitter misses (pvtrace/engine/emit.py:62-89)
— and samples them either with seeded numpy on the host or directly on the
device; anything else falls back to one Python call per ray.
"""
from dataclasses import dataclass, replace
from enum import Enum
from typing import Optional

import numpy as np

SPEED_OF_LIGHT_CM_PER_S = 2.99792458e10


class Event(Enum):
    """What happened to a ray (codes shared with the device kernel)."""

    GENERATE = 0
    REFLECT = 1
    TRANSMIT = 2
    ABSORB = 3
    NONRADIATIVE = 4
    SCATTER = 5
    EMIT = 6
    EXIT = 7
    REACT = 8
    KILL = 9


@dataclass(frozen=True)
class Ray:
    """Position (cm), unit direction, wavelength (nm), path length travelled
    (cm), elapsed time (s) and the name of the light/component that emitted it."""

    position: tuple
    direction: tuple
    wavelength: Optional[float]
    travelled: float = 0.0
    duration: float = 0.0
    source: Optional[str] = None

    def __repr__(self):
        fmt = lambda v: "(" + ", ".join("{:.2f}".format(c) for c in v) + ")"
        return "Ray(pos={}, dir={}, nm={:.2f})".format(
            fmt(self.position), fmt(self.direction), self.wavelength
        )

    def propagate(self, distance, refractive_index):
        pos = np.asarray(self.position, dtype=np.float64)
        step = np.asarray(self.direction, dtype=np.float64) * distance
        return replace(
            self,
            position=tuple((pos + step).tolist()),
            travelled=self.travelled + distance,
            duration=self.duration
            + distance * refractive_index / SPEED_OF_LIGHT_CM_PER_S,
        )

    def representation(self, from_node, to_node):
        return replace(
            self,
            position=from_node.point_to_node(self.position, to_node),
            direction=from_node.vector_to_node(self.direction, to_node),
        )


# -- delegate callables ---------------------------------------------------

def default_wavelength():
    return 555.0


def default_position():
    return (0.0, 0.0, 0.0)


def default_direction():
    return (0.0, 0.0, 1.0)


def rectangular_mask(X, Y):
    return (np.random.uniform(-X, X), np.random.uniform(-Y, Y), 0.0)


def circular_mask(radius):
    angle = np.random.uniform(0, 2.0 * np.pi)
    r = np.sqrt(np.random.uniform()) * radius
    return (r * np.cos(angle), r * np.sin(angle), 0.0)


def cube_mask(X, Y, Z):
    return (
        np.random.uniform(-X, X),
        np.random.uniform(-Y, Y),
        np.random.uniform(-Z, Z),
    )


class DefaultWavelength(object):
    def __call__(self):
        return default_wavelength()


class DefaultPosition(object):
    def __call__(self):
        return default_position()


class DefaultDirection(object):
    def __call__(self):
        return default_direction()


class ConstantWavelengthMask(object):
    def __init__(self, nanometers):
        self.nanometers = float(nanometers)

    def __call__(self):
        return self.nanometers


class SpectrumWavelengthMask(object):
    """Wavelengths drawn from a `Distribution` by inverse-CDF sampling."""

    def __init__(self, distribution):
        self.distribution = distribution

    def __call__(self):
        return self.distribution.sample(np.random.uniform(0, 1))


class RectangularMask(object):
    def __init__(self, x, y):
        self.x, self.y = float(x), float(y)

    def __call__(self):
        return rectangular_mask(self.x, self.y)


class CircularMask(object):
    def __init__(self, radius):
        self.radius = radius

    def __call__(self):
        return circular_mask(self.radius)


class CubeMask(object):
    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z

    def __call__(self):
        return cube_mask(self.x, self.y, self.z)


class Light(object):
    """Emits rays along +z of its node unless delegates say otherwise."""

    def __init__(self, wavelength=None, position=None, direction=None, name="Light"):
        self.wavelength = default_wavelength if wavelength is None else wavelength
        self.position = default_position if position is None else position
        self.direction = default_direction if direction is None else direction
        self.name = name

    def emit(self, num_rays=None):
        if num_rays is None or num_rays == 0:
            return
        for _ in range(num_rays):
            yield Ray(
                wavelength=self.wavelength(),
                position=tuple(self.position()),
                direction=tuple(np.asarray(self.direction()).tolist()),
                source=self.name,
            )
