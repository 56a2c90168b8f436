"""Materials: refractive index, surface delegates and volume components.

Constructor-level mirror of the reference's material package
(pvtrace/material/material.py:10-63, component.py:33-440, distribution.py:8-193,
surface.py:13-272, utils.py:8-186).  These objects only *describe* a material:
the flattener (`pvtrace_amd.engine.compiler`) lowers them to SoA tables and the
HIP kernel does all per-photon sampling.  The scalar helper functions
(`fresnel_reflectivity`, `cone`, ...) are kept because user scenes pass them as
delegates (e.g. ``functools.partial(cone, theta)``) and tests use them as known
answers.
"""
import abc
import math
from dataclasses import replace

import numpy as np

Q_E = 1.60217662e-19  # C
KB_EV = 1.380649e-23 / Q_E  # eV / K   (reference component.py:25-26)


# ----------------------------------------------------------------------
# Optics helpers (reference material/utils.py:8-45)

def fresnel_reflectivity(angle, n1, n2):
    """Unpolarised Fresnel reflectivity; 1.0 beyond the critical angle."""
    if n2 < n1 and angle > math.asin(n2 / n1):
        return 1.0
    c, s = math.cos(angle), math.sin(angle)
    k = math.sqrt(1.0 - (n1 / n2 * s) ** 2)
    rs = ((n1 * c - n2 * k) / (n1 * c + n2 * k)) ** 2
    rp = ((n1 * k - n2 * c) / (n1 * k + n2 * c)) ** 2
    return 0.5 * (rs + rp)


def specular_reflection(direction, normal):
    d = np.asarray(direction, dtype=np.float64)
    n = np.asarray(normal, dtype=np.float64)
    if np.dot(n, d) < 0.0:
        n = -n
    return d - 2.0 * np.dot(n, d) * n


def fresnel_refraction(direction, normal, n1, n2):
    """Snell refraction, vector form; `normal` must point along the ray."""
    d = np.asarray(direction, dtype=np.float64)
    nrm = np.asarray(normal, dtype=np.float64)
    n = n1 / n2
    dd = float(np.dot(d, nrm))
    c = math.sqrt(1.0 - n * n * (1.0 - dd * dd))
    sign = -1.0 if dd < 0.0 else 1.0
    return n * d + sign * (c - sign * n * dd) * nrm


def gaussian(x, c1, c2, c3):
    return c1 * np.exp(-(((c2 - x) / c3) ** 2))


def bandgap(x, cutoff, alpha):
    return (1 - np.heaviside(x - cutoff, 0.5)) * alpha


def simple_convert_spectum(spec):
    """(n, 2) spectrum with its x column turned from nanometres into electron-volts or back (E [eV] = hc/q 10^9 / nm is
    its own inverse); the y column untouched.  (The reference's helper of this -- misspelt -- name, material/utils.py:59-69.)"""
    out = np.array(spec)
    out[:, 0] = (6.62607015e-34 * 299792458.0 / 1.60217662e-19 * 1e9) / out[:, 0]
    return out


def thermodynamic_emission(abs_spec, T=300, mu=0.5):
    """Emission line shape in detailed balance with an absorption spectrum (generalised Planck
    law): `abs_spec` is an (n, 2) array of (nm, absorptance); returns (nm, emission) normalised to
    a peak of 1.  `T` in kelvin, `mu` the chemical potential in eV.  (Counterpart of the
    reference's pvtrace/material/utils.py:72-85, used to build test spectra.)"""
    planck, light, charge, boltzmann = 6.62607015e-34, 299792458.0, 1.60217662e-19, 1.38064852e-23
    spec = np.asarray(abs_spec, dtype=np.float64)
    ev = planck * light / charge * 1e9 / spec[:, 0]                  # photon energy of each row
    occupation = np.expm1((ev - mu) / (boltzmann / charge * T))
    emission = spec[:, 1] * ev * ev / occupation
    return np.column_stack((spec[:, 0], emission / np.max(emission)))


def spherical_to_cart(theta, phi, r=1):
    cart = np.column_stack((r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)))
    return cart[0, :] if cart.size == 3 else cart


# ----------------------------------------------------------------------
# Phase functions / angular distributions (reference material/utils.py:104-186).
# The engine recognises these by identity / type and samples them on the
# device; calling them directly draws from numpy's global generator -- with numpy's
# own arccos / arcsin / sin / cos, as the reference's do: under one numpy seed the
# two packages return the same directions to the last bit (tests/golden/object_methods.npz).

def isotropic():
    g1, g2 = np.random.uniform(0, 1, 2)
    return spherical_to_cart(np.arccos(2 * g2 - 1), 2 * np.pi * g1)


def henyey_greenstein(g=0.0):
    p = np.random.uniform(0, 1)
    if abs(g) < 2.220446049250313e-13:
        return isotropic()
    s = 2 * p - 1
    mu = 1 / (2 * g) * (1 + g ** 2 - ((1 - g ** 2) / (1 + g * s)) ** 2)
    phi = 2 * np.pi * np.random.uniform()
    return spherical_to_cart(np.arccos(mu), phi)


def cone(theta_max):
    if np.isclose(theta_max, 0.0) or theta_max > np.pi / 2:
        raise ValueError("Expected 0 < theta_max <= pi/2")
    p1, p2 = np.random.uniform(0, 1, 2)
    return spherical_to_cart(np.arcsin(np.sqrt(p1) * np.sin(theta_max)), 2 * np.pi * p2)


def lambertian():
    p1, p2 = np.random.uniform(0, 1, 2)
    return spherical_to_cart(np.arcsin(np.sqrt(p1)), 2 * np.pi * p2)


class HenyeyGreenstein(object):
    def __init__(self, g):
        self.g = float(g)

    def __call__(self):
        return henyey_greenstein(self.g)


class Cone(object):
    def __init__(self, theta_max):
        self.theta_max = float(theta_max)

    def __call__(self):
        return cone(self.theta_max)


# ----------------------------------------------------------------------
# Spectral distribution (reference material/distribution.py:8-193)

class Distribution(object):
    """A sampled spectrum y(x) with its cumulative distribution.

    With ``hist=False`` the CDF is the trapezoid integral normalised to 1 with a
    leading 0 (`_cdf` has the same length as `_x`); this is what the device
    kernel inverts for emission-wavelength sampling.  A constant is represented
    by ``x=None`` and a float `y`.
    """

    def __init__(self, x, y, hist=False):
        self.hist = hist
        if x is None and isinstance(y, float):
            self._x, self._y = None, y
            return
        x = np.asarray(x)
        y = np.asarray(y)
        if not np.all(np.diff(x) > 0):
            raise ValueError("x must be sorted and ascending.")
        if not np.isfinite(y).any():
            raise ValueError("All values of y must be finite.")
        if np.any(y < 0.0):
            raise ValueError(
                "Distributions are like histograms all counts must be positive."
            )
        self._x_range = (np.min(x), np.max(x))
        self._x, self._y = x, y
        if hist:
            cdf = np.cumsum(y, dtype=float)
            cdf *= 1.0 / cdf[-1]
            self._cdf = cdf
            self._edges = np.insert(x, x.size, 2 * x[-1] - x[-2])
        else:
            cdf = np.cumsum((y[:-1] + y[1:]) * 0.5)
            cdf = cdf / np.max(cdf)
            self._cdf = np.hstack([0.0, cdf])

    def _check(self, v, lo, hi, what):
        arr = np.asarray(v)
        if np.any(arr < lo) or np.any(arr > hi):
            raise ValueError(what, {"value": v, "range": (lo, hi)})

    def __call__(self, x):
        if self._x is None:
            if isinstance(x, (list, tuple, np.ndarray)):
                return np.zeros(len(x)) + self._y
            return self._y
        self._check(x, *self._x_range, "x is outside data range.")
        if self.hist:
            return self._y[np.searchsorted(self._edges[:-1], x)]
        return np.interp(x, self._x, self._y, left=np.nan, right=np.nan)

    def lookup(self, x):
        """CDF value at x."""
        self._check(x, *self._x_range, "x is outside data range.")
        if self.hist:
            return self._cdf[np.searchsorted(self._edges[:-1], x)]
        prob = np.interp(x, self._x, self._cdf, left=np.nan, right=np.nan)
        return prob.tolist() if np.size(prob) == 1 else prob

    def sample(self, p):
        """Inverse CDF."""
        self._check(p, 0.0, 1.0, "p is outside valid range.")
        if self.hist:
            idx = np.searchsorted(self._cdf, p)
            try:
                return self._x[idx]
            except IndexError:
                return self._x[-1]
        xval = np.interp(p, self._cdf, self._x, left=np.nan, right=np.nan)
        return xval.tolist() if np.size(xval) == 1 else xval

    @classmethod
    def from_functions(cls, x, callables, hist=False):
        x = np.array(x)
        if x.ndim != 1:
            raise ValueError("Requires a 1D array.")
        y = np.zeros(len(x))
        for f in callables:
            part = f(x)
            part[np.where(~np.isfinite(part))] = 0.0
            y += part
        return cls(x=x, y=y, hist=hist)


# ----------------------------------------------------------------------
# Surfaces (reference material/surface.py:13-272)

class SurfaceDelegate(abc.ABC):
    """Per-interaction surface behaviour.  Arbitrary Python delegates cannot
    run on the device; the flattener accepts the built-in delegates below and
    declarative `CoatedSurfaceDelegate` objects, and rejects everything else
    with `UnsupportedSceneError` (as the reference compiler does,
    pvtrace/engine/compiler.py:237-247)."""

    @abc.abstractmethod
    def reflectivity(self, surface, ray, geometry, container, adjacent):
        pass

    @abc.abstractmethod
    def reflected_direction(self, surface, ray, geometry, container, adjacent):
        pass

    @abc.abstractmethod
    def transmitted_direction(self, surface, ray, geometry, container, adjacent):
        pass


def _flipped_normal(geometry, ray):
    normal = np.asarray(geometry.normal(ray.position), dtype=np.float64)
    if np.dot(normal, ray.direction) < 0.0:
        normal = -normal
    return normal


class FresnelSurfaceDelegate(SurfaceDelegate):
    """Fresnel reflection / Snell refraction from the two refractive indices."""

    def reflectivity(self, surface, ray, geometry, container, adjacent):
        n1 = container.geometry.material.refractive_index
        n2 = adjacent.geometry.material.refractive_index
        normal = _flipped_normal(geometry, ray)
        cosang = float(np.clip(np.dot(normal, ray.direction), -1.0, 1.0))
        return float(fresnel_reflectivity(math.acos(cosang), n1, n2))

    def reflected_direction(self, surface, ray, geometry, container, adjacent):
        normal = geometry.normal(ray.position)
        return tuple(specular_reflection(ray.direction, normal).tolist())

    def transmitted_direction(self, surface, ray, geometry, container, adjacent):
        n1 = container.geometry.material.refractive_index
        n2 = adjacent.geometry.material.refractive_index
        normal = _flipped_normal(geometry, ray)
        return tuple(fresnel_refraction(ray.direction, normal, n1, n2).tolist())


class NullSurfaceDelegate(SurfaceDelegate):
    """Transmits everything without refraction (useful for counting)."""

    def reflectivity(self, surface, ray, geometry, container, adjacent):
        return 0.0

    def reflected_direction(self, surface, ray, geometry, container, adjacent):
        raise NotImplementedError("This surface delegate does not reflect.")

    def transmitted_direction(self, surface, ray, geometry, container, adjacent):
        return ray.direction


class Coating(object):
    """Declarative override of the optics on part of a node's surface.

    The reference expresses coatings as Python subclasses of
    `FresnelSurfaceDelegate` that inspect the hit normal / position per ray
    (e.g. pvtrace/device/lsc.py:22-86, examples/006 Coatings.ipynb cell 3);
    callbacks cannot run on the GPU, so the same behaviours are written as data:

    facet : outward face normal in the node's LOCAL frame the coating covers
        (matched like ``np.allclose``: |n_i - facet_i| <= 1e-8 + 1e-5 |facet_i|).
    region : optional ((xlo, xhi), (ylo, yhi), (zlo, zhi)) open intervals in the
        local frame restricting where on that face it applies (None = unbounded).
    reflectivity : probability of reflection in [0, 1], or None to keep Fresnel.
    reflection : "specular" or "lambertian" (cosine-weighted about the outward
        facet normal, in the local frame).
    transmission : "fresnel" (Snell refraction) or "matched" (index-matched:
        direction unchanged, e.g. a perfectly coupled solar cell).
    """

    REFLECTION_MODES = {"specular": 0, "lambertian": 1}
    TRANSMISSION_MODES = {"fresnel": 0, "matched": 1}

    def __init__(
        self,
        facet,
        reflectivity=None,
        region=None,
        reflection="specular",
        transmission="fresnel",
    ):
        self.facet = tuple(float(v) for v in facet)
        if len(self.facet) != 3:
            raise ValueError("facet must be a 3-vector")
        if reflectivity is not None and not 0.0 <= float(reflectivity) <= 1.0:
            raise ValueError("reflectivity must be in [0, 1] or None")
        self.reflectivity = None if reflectivity is None else float(reflectivity)
        if reflection not in self.REFLECTION_MODES:
            raise ValueError(f"reflection must be one of {sorted(self.REFLECTION_MODES)}")
        if transmission not in self.TRANSMISSION_MODES:
            raise ValueError(
                f"transmission must be one of {sorted(self.TRANSMISSION_MODES)}"
            )
        self.reflection = reflection
        self.transmission = transmission
        bounds = []
        for axis in range(3):
            pair = None if region is None else region[axis]
            lo, hi = (None, None) if pair is None else pair
            bounds.append(
                (-math.inf if lo is None else float(lo), math.inf if hi is None else float(hi))
            )
        self.region = tuple(bounds)

    def covers(self, normal, position):
        for a in range(3):
            if abs(normal[a] - self.facet[a]) > 1e-8 + 1e-5 * abs(self.facet[a]):
                return False
            lo, hi = self.region[a]
            if not (lo < position[a] < hi):
                return False
        return True


class CoatedSurfaceDelegate(FresnelSurfaceDelegate):
    """Fresnel surface with an ordered list of `Coating` overrides; the first
    coating covering the hit point wins, uncovered points are plain Fresnel."""

    def __init__(self, coatings=None):
        super(CoatedSurfaceDelegate, self).__init__()
        self._coatings = [] if coatings is None else list(coatings)

    @property
    def coatings(self):
        return list(self._coatings)

    def _match(self, ray, geometry):
        normal = geometry.normal(ray.position)
        for coating in self.coatings:
            if coating.covers(normal, ray.position):
                return coating
        return None

    def reflectivity(self, surface, ray, geometry, container, adjacent):
        fresnel = super(CoatedSurfaceDelegate, self).reflectivity(
            surface, ray, geometry, container, adjacent
        )
        coating = self._match(ray, geometry)
        if coating is None or coating.reflectivity is None:
            return fresnel
        if fresnel == 1.0 and coating.transmission != "matched":
            return 1.0   # beyond the critical angle no refracted ray exists: stays totally reflected
        return coating.reflectivity

    def transmitted_direction(self, surface, ray, geometry, container, adjacent):
        coating = self._match(ray, geometry)
        if coating is not None and coating.transmission == "matched":
            return tuple(ray.direction)
        return super(CoatedSurfaceDelegate, self).transmitted_direction(
            surface, ray, geometry, container, adjacent
        )


class BaseSurface(abc.ABC):
    """What a material's surface answers (reference material/surface.py:180-203): its delegate, and the three verbs."""

    @property
    @abc.abstractmethod
    def delegate(self):
        """An object that implements `SurfaceDelegate`."""

    @abc.abstractmethod
    def is_reflected(self, ray, geometry, container, adjacent):
        """True when the ray is reflected."""

    @abc.abstractmethod
    def reflect(self, ray, geometry, container, adjacent):
        """The reflected ray."""

    @abc.abstractmethod
    def transmit(self, ray, geometry, container, adjacent):
        """The transmitted ray."""


class Surface(BaseSurface):
    def __init__(self, delegate=None):
        super(Surface, self).__init__()
        self._delegate = FresnelSurfaceDelegate() if delegate is None else delegate

    @property
    def delegate(self):
        return self._delegate

    # The per-interaction verbs of the reference's Python tracer (surface.py:224-272), for code that steps rays itself:
    # one uniform draw from numpy's global generator decides, the delegate supplies reflectivity and directions.  (The
    # engine does not call these: it lowers the delegate to tables, `engine/compiler.py`.)
    def is_reflected(self, ray, geometry, container, adjacent):
        r = self.delegate.reflectivity(self, ray, geometry, container, adjacent)
        if not isinstance(r, (int, float)):
            raise ValueError("Reflectivity must be a number.")
        if r == 0.0:
            return False   # (no draw: keeps a seeded sequence in step with the reference's)
        return bool(np.random.uniform() < r)

    def _turned(self, ray, which, *where):
        direction = getattr(self.delegate, which)(self, ray, *where)
        if not isinstance(direction, tuple):
            raise ValueError(f"Delegate method `{which}` should return a tuple.")
        if len(direction) != 3:
            raise ValueError(f"Delegate method `{which}` should return a tuple of length 3.")
        return replace(ray, direction=direction)

    def reflect(self, ray, geometry, container, adjacent):
        return self._turned(ray, "reflected_direction", geometry, container, adjacent)

    def transmit(self, ray, geometry, container, adjacent):
        return self._turned(ray, "transmitted_direction", geometry, container, adjacent)


# ----------------------------------------------------------------------
# Volume components (reference material/component.py:33-440)

def _spectrum(value, x, hist, what):
    """A constant, an (n, 2) table or a list of callables sampled on `x` -> `Distribution` (the three spellings the
    reference's components accept for a coefficient or an emission spectrum, component.py:64-90, :327-350)."""
    if isinstance(value, np.ndarray):
        return Distribution(x=value[:, 0], y=value[:, 1], hist=hist)
    if isinstance(value, (list, tuple)):
        if x is None:
            raise ValueError(f"{what} given as callables needs `x`, the wavelengths to sample them on.")
        return Distribution.from_functions(x, value, hist=hist)
    if isinstance(value, float):
        return Distribution(x=None, y=value, hist=hist)
    raise ValueError(f"{what}: expected a number, an (n, 2) array or a list of callables, got {type(value).__name__}.")


class Component:
    def __init__(self, name="Component"):
        self.name = name

    def is_radiative(self, ray):
        return False

    def nonradiative_absorb(self, ray):
        return ray


class Scatterer(Component):
    """Scattering centre with attenuation coefficient (cm^-1), constant or spectral."""

    def __init__(self, coefficient, x=None, quantum_yield=1.0, tau_rad=None, tau_nr=None, phase_function=None,
                 hist=False, name="Scatterer"):
        super().__init__(name=name)
        if coefficient is None:
            raise ValueError("A component needs an attenuation coefficient.")
        if isinstance(coefficient, (int, np.integer, np.floating)) and not isinstance(coefficient, bool):
            coefficient = float(coefficient)
        self._coefficient = coefficient
        self._abs_dist = _spectrum(coefficient, x, hist, "coefficient")
        # two lifetimes fix the yield; otherwise it is given (reference component.py:92-104)
        both = tau_rad is not None and tau_nr is not None
        qy = tau_nr / (tau_nr + tau_rad) if both else (float("nan") if quantum_yield is None else quantum_yield)
        if not np.isfinite(qy):
            raise ValueError("Give `quantum_yield`, or both `tau_rad` and `tau_nr`.")
        self.quantum_yield, self.tau_rad, self.tau_nr = qy, tau_rad, tau_nr
        self.phase_function = phase_function or isotropic

    def coefficient(self, wavelength):
        return self._abs_dist(wavelength)

    # What happens to an absorbed ray, one numpy draw per decision in the reference's order (component.py:168-196):
    def is_radiative(self, ray):
        return bool(np.random.uniform() < self.quantum_yield)

    def nonradiative_absorb(self, ray):
        """The ray as it ends: with `tau_nr` set its clock runs on by an exponentially distributed delay."""
        if self.tau_nr:
            return replace(ray, duration=ray.duration - np.log(1 - np.random.uniform()) * self.tau_nr)
        return ray

    def emit(self, ray, **kwargs):
        """Scattered: a new direction from the phase function (in the frame the ray is given in), the scatterer as source."""
        return replace(ray, direction=self.phase_function(), source=self.name)


class Absorber(Scatterer):
    """Non-radiative absorber (quantum yield 0)."""

    def __init__(self, coefficient, x=None, tau_nr=None, name="Absorber", hist=False):
        super().__init__(coefficient, x=x, quantum_yield=0.0, tau_rad=0.0, tau_nr=tau_nr, hist=hist, name=name)

    def is_radiative(self, ray):
        return False   # (and no draw, as in the reference, component.py:236-239)


class Reactor(Absorber):
    """Absorber whose absorptions are tallied as photochemical reactions."""

    def __init__(self, coefficient, x=None, name="Reactor", hist=False):
        super().__init__(coefficient, x=x, hist=hist, name=name)


class Luminophore(Scatterer):
    """Absorbs and re-emits with a new wavelength drawn from `emission`."""

    def __init__(self, coefficient, emission=None, x=None, hist=False, quantum_yield=1.0, tau_rad=None, tau_nr=None,
                 phase_function=None, name="Luminophore"):
        super().__init__(coefficient, x=x, quantum_yield=quantum_yield, tau_rad=tau_rad, tau_nr=tau_nr,
                         phase_function=phase_function, hist=hist, name=name)
        self._emission = emission
        if emission is None:   # the reference's default line: a Gaussian at 600 nm, 40 nm wide (component.py:330-335)
            emission = [lambda v: gaussian(v, 1.0, 600.0, 40.0)]
        elif isinstance(emission, float):
            raise ValueError("emission: expected an (n, 2) array or a list of callables.")
        self._ems_dist = _spectrum(emission, x, hist, "emission")

    def emit(self, ray, method="kT", T=300.0, **kwargs):
        """Re-emitted (reference component.py:381-440; draws in its order: phase function, wavelength, delay): a new
        direction, a wavelength from the emission spectrum above the point `method` allows -- "kT": from 3/2 kT (at `T` K)
        above the absorbed photon's energy, "redshift": from the absorbed wavelength, "full": the whole spectrum -- and,
        with `tau_rad` set, an exponentially distributed emission delay.  Like the reference's it raises when "kT" lands
        outside the spectrum's range; the engine clamps there (differences between the two tracers)."""
        direction = self.phase_function()
        nm = ray.wavelength
        if method == "kT":
            nm = 1240.0 / (1240.0 / nm + 3 / 2 * KB_EV * T)
            start = self._ems_dist.lookup(nm)
        elif method == "redshift":
            start = self._ems_dist.lookup(nm)
        elif method == "full":
            start = 0.0
        else:
            raise ValueError(f"emit method {method!r}: use 'kT', 'redshift' or 'full'")
        wavelength = self._ems_dist.sample(np.random.uniform(start, 1.0))
        delay = -np.log(1 - np.random.uniform()) * self.tau_rad if self.tau_rad else 0.0
        return replace(ray, direction=direction, wavelength=wavelength, source=self.name, duration=ray.duration + delay)


class Material(object):
    def __init__(self, refractive_index, surface=None, components=None):
        self.refractive_index = refractive_index
        self.surface = Surface() if surface is None else surface
        self.components = [] if components is None else components

    def total_attenutation_coefficient(self, wavelength):
        return float(np.sum([c.coefficient(wavelength) for c in self.components]))

    # The volume decisions of the reference's Python tracer (material.py:22-63), one numpy draw each:
    def penetration_depth(self, wavelength):
        """How far a photon of this wavelength gets before something absorbs it: exponential with the summed
        coefficient; inf for a clear medium, 0 for an opaque one."""
        alpha = self.total_attenutation_coefficient(wavelength)
        if np.isclose(alpha, 0.0):
            return float("inf")
        if not np.isfinite(alpha):
            return 0.0
        return -np.log(1 - np.random.uniform()) / alpha

    def is_absorbed(self, ray, full_distance):
        """(absorbed before `full_distance`?, the sampled depth)"""
        depth = self.penetration_depth(ray.wavelength)
        return (depth < full_distance, depth)

    def component(self, wavelength):
        """Which component took the photon: each in proportion to its coefficient at this wavelength."""
        weights = np.array([c.coefficient(wavelength) for c in self.components])
        if np.any(weights < 0.0):
            raise ValueError("Must be positive.")
        steps = np.cumsum(weights)
        ladder = np.hstack([0, steps / max(steps)])
        at = np.interp(np.random.uniform(), ladder, list(range(len(self.components) + 1)))
        return self.components[int(np.floor(at))]
