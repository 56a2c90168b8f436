"""Geometry primitives and rigid poses for the scene API.

Mirrors the constructor surface of the reference's geometry package
(pvtrace/geometry/box.py:22-45, sphere.py:9-20, cylinder.py:10-23,
transformable.py:12-96).  Only analytic primitives the device tracer supports
exist here; every shape is centred on its node's origin, cylinders run along z.

The per-ray queries (`intersections`, `normal`) are numpy restatements of the
device functions in csrc/pvt_trace.hip and follow the *kernel* semantics
(reference pvtrace/engine/_kernel.pyx:245-400), i.e. Box is analytic f64 rather
than a triangle mesh.  They exist for host-side plumbing and unit tests; the
hot path never calls them.
"""
import math

import numpy as np

from pvtrace_amd.common import GeometryError

# Distance tolerance (reference pvtrace/geometry/utils.py:12, _kernel.pyx:29)
EPS_ZERO = 2.220446049250313e-13


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / math.sqrt(float(np.dot(v, v)))


def translation_matrix(vector):
    m = np.identity(4)
    m[:3, 3] = np.asarray(vector, dtype=np.float64)[:3]
    return m


def rotation_matrix(angle, axis, point=None):
    """Homogeneous rotation by `angle` about `axis` through `point`
    (Rodrigues form; same convention as the reference's vendored
    transformations.rotation_matrix, pvtrace/geometry/transformations.py:303-348)."""
    s, c = math.sin(angle), math.cos(angle)
    d = _unit(np.asarray(axis, dtype=np.float64)[:3])
    rot = np.diag([c, c, c]) + np.outer(d, d) * (1.0 - c)
    d = d * s
    rot = rot + np.array(
        [[0.0, -d[2], d[1]], [d[2], 0.0, -d[0]], [-d[1], d[0], 0.0]]
    )
    m = np.identity(4)
    m[:3, :3] = rot
    if point is not None:
        p = np.asarray(point, dtype=np.float64)[:3]
        m[:3, 3] = p - rot @ p
    return m


class Transformable(object):
    """A coordinate frame with a 4x4 `pose` relative to its parent.

    `translate` moves the frame in the parent's coordinates; `rotate` is a
    body rotation about the frame's current location (the location is kept).
    """

    def __init__(self, location=None):
        super(Transformable, self).__init__()
        loc = np.zeros(3) if location is None else np.array(location, dtype=np.float64)
        self._location = loc
        self._pose = translation_matrix(loc)

    @classmethod
    def from_pose(cls, pose):
        pose = np.asarray(pose, dtype=np.float64)
        if pose.shape != (4, 4):
            raise ValueError("Must be a 4x4 transform matrix")
        obj = cls()
        obj.pose = pose
        return obj

    @property
    def pose(self):
        return self._pose

    @pose.setter
    def pose(self, value):
        value = np.array(value, dtype=np.float64)
        self._location = value[:3, 3].copy()
        self._pose = value

    @property
    def location(self):
        return self._location

    @location.setter
    def location(self, value):
        self._location = np.array(value, dtype=np.float64)
        self._pose[:3, 3] = self._location

    def translate(self, vector):
        vector = np.asarray(vector, dtype=np.float64)
        self._location = self._location + vector
        self._pose = translation_matrix(vector) @ self._pose
        return self

    def rotate(self, angle, axis):
        self._pose = rotation_matrix(angle, axis, point=self._location) @ self._pose
        return self


class Geometry(object):
    """Base class: a shape with a material."""

    def __init__(self, material=None):
        self._material = material

    @property
    def material(self):
        return self._material

    @material.setter
    def material(self, value):
        self._material = value

    # -- host restatements of the device queries ------------------------
    def intersections(self, origin, direction):
        """Forward intersection points (distance > EPS_ZERO), nearest first."""
        o = np.asarray(origin, dtype=np.float64)
        d = np.asarray(direction, dtype=np.float64)
        ts = sorted(self._ray_distances(o, d))
        return tuple(tuple((o + t * d).tolist()) for t in ts)

    def _ray_distances(self, o, d):
        raise NotImplementedError

    def normal(self, point):
        raise NotImplementedError


class Box(Geometry):
    """Axis-aligned box of side lengths `size`, centred on the origin."""

    def __init__(self, size, material=None):
        super(Box, self).__init__(material=material)
        self._size = np.array(size, dtype=np.float64)
        if self._size.shape != (3,):
            raise ValueError("Box size must be (length, width, height).")

    @property
    def size(self):
        return self._size

    def _ray_distances(self, o, d):
        tmin, tmax = -math.inf, math.inf
        for a in range(3):
            lo, hi = -0.5 * self._size[a], 0.5 * self._size[a]
            if abs(d[a]) < 1e-300:
                if o[a] < lo or o[a] > hi:
                    return []
            else:
                inv = 1.0 / d[a]
                t1, t2 = (lo - o[a]) * inv, (hi - o[a]) * inv
                if t1 > t2:
                    t1, t2 = t2, t1
                tmin, tmax = max(tmin, t1), min(tmax, t2)
        if tmax < tmin:
            return []
        return [t for t in (tmin, tmax) if t > EPS_ZERO]

    def contains(self, point):
        p = np.abs(np.asarray(point, dtype=np.float64))
        return bool(np.all(0.5 * self._size - (p + EPS_ZERO) > 0.0))

    def is_on_surface(self, point):
        p = np.abs(np.asarray(point, dtype=np.float64))
        half = 0.5 * self._size
        inside = np.all(p <= half + EPS_ZERO)
        return bool(inside and np.any(np.abs(p - half) < EPS_ZERO))

    def normal(self, point):
        p = np.asarray(point, dtype=np.float64)
        best, out = math.inf, (0.0, 0.0, 0.0)
        for a in range(3):
            for sign in (-1.0, 1.0):
                dist = abs(p[a] - sign * 0.5 * self._size[a])
                if dist < best:
                    best = dist
                    n = [0.0, 0.0, 0.0]
                    n[a] = sign
                    out = tuple(n)
        return out


class Sphere(Geometry):
    """Sphere of `radius` centred on the origin."""

    def __init__(self, radius, material=None):
        super(Sphere, self).__init__(material=material)
        self.radius = radius

    def _ray_distances(self, o, d):
        a = float(np.dot(d, d))
        b = 2.0 * float(np.dot(d, o))
        c = float(np.dot(o, o)) - self.radius * self.radius
        disc = b * b - 4.0 * a * c
        if disc < 0.0:
            return []
        sq = math.sqrt(disc)
        ts = [(-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)]
        return [t for t in ts if t > EPS_ZERO]

    def contains(self, point):
        r = math.sqrt(float(np.sum(np.asarray(point, dtype=np.float64) ** 2)))
        return self.radius - (r + EPS_ZERO) > 0.0

    def is_on_surface(self, point):
        r = math.sqrt(float(np.sum(np.asarray(point, dtype=np.float64) ** 2)))
        return abs(r - self.radius) < EPS_ZERO

    def normal(self, point):
        p = np.asarray(point, dtype=np.float64)
        return tuple((p / math.sqrt(float(np.dot(p, p)))).tolist())


class Cylinder(Geometry):
    """Capped cylinder along z, centred on the origin."""

    def __init__(self, length, radius, material=None):
        super(Cylinder, self).__init__(material=material)
        self.length = length
        self.radius = radius

    def _ray_distances(self, o, d):
        half, rad = 0.5 * self.length, self.radius
        cand = []
        a = d[0] * d[0] + d[1] * d[1]
        if a > 1e-300:
            b = 2.0 * (o[0] * d[0] + o[1] * d[1])
            c = o[0] * o[0] + o[1] * o[1] - rad * rad
            disc = b * b - 4.0 * a * c
            if disc >= 0.0:
                sq = math.sqrt(disc)
                for t in ((-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)):
                    z = o[2] + t * d[2]
                    if -half < z < half:
                        cand.append(t)
        if abs(d[2]) > 1e-300:
            for cap in (-half, half):
                t = (cap - o[2]) / d[2]
                x, y = o[0] + t * d[0], o[1] + t * d[1]
                if x * x + y * y <= rad * rad:
                    cand.append(t)
        return [t for t in cand if t > EPS_ZERO]

    def contains(self, point):
        p = np.asarray(point, dtype=np.float64)
        r = math.hypot(p[0], p[1])
        return bool(
            self.radius - (r + EPS_ZERO) > 0.0
            and 0.5 * self.length - (abs(p[2]) + EPS_ZERO) > 0.0
        )

    def normal(self, point):
        p = np.asarray(point, dtype=np.float64)
        half = 0.5 * self.length
        tol = 1e-8 + 1e-5 * abs(half)
        if abs(p[2] + half) <= tol:
            return (0.0, 0.0, -1.0)
        if abs(p[2] - half) <= tol:
            return (0.0, 0.0, 1.0)
        r = math.sqrt(p[0] * p[0] + p[1] * p[1])
        if r == 0.0:
            raise GeometryError("Point on the cylinder axis has no radial normal.")
        return (p[0] / r, p[1] / r, 0.0)
