"""Geometry primitives and rigid poses for the scene API.

Mirrors the constructor surface of the reference's geometry package
(pvtrace/geometry/box.py:22-45, sphere.py:9-20, cylinder.py:10-23,
transformable.py:12-96).  Only analytic primitives the device tracer supports
exist here; every shape is centred on its node's origin, cylinders run along z.

The per-ray queries (`intersections`, `normal`) are numpy restatements of the
device functions in csrc/pvt_trace_kernel.h and follow the *kernel* semantics
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


# ----------------------------------------------------------------------
# Small vector and tolerance helpers (the names of reference geometry/utils.py:364-440): what hand-written surface
# delegates are written with -- `flip`, `angle_between` and friends; `EPS_ZERO` is the one tolerance everything shares.

def close_to_zero(value):
    """Every component within EPS_ZERO of zero (strictly)."""
    return bool(np.all(np.absolute(value) < EPS_ZERO))


def floats_close(a, b):
    return close_to_zero(a - b)


def distance_between(point1, point2):
    return float(np.linalg.norm(np.subtract(point1, point2)))


def points_equal(point1, point2):
    return close_to_zero(distance_between(point1, point2))


def allinrange(x, x_range):
    """No value of `x` (a number or an array) lies outside the closed interval `x_range`."""
    values = np.atleast_1d(np.asarray(x))
    return not bool(np.any((values < x_range[0]) | (values > x_range[1])))


def flip(vector):
    return -np.array(vector)


def magnitude(vector):
    v = np.array(vector)
    return np.sqrt(np.dot(v, v))


def norm(vector):
    return np.array(vector) / np.linalg.norm(vector)


def angle_between(normal, vector):
    """Angle in radians between two UNIT vectors; exactly 0 and pi for (numerically) parallel and anti-parallel ones."""
    a, b = np.array(normal), np.array(vector)
    if np.allclose(a, b):
        return 0.0
    if np.allclose(-a, b):
        return np.pi
    return np.arccos(np.dot(a, b))


def smallest_angle_between(normal, vector):
    rads = angle_between(normal, vector)
    return np.arctan2(np.sin(rads), np.cos(rads))


def intersection_point_is_ahead(ray_position, ray_direction, intersection_point):
    """A point ON the ray's line lies ahead of its origin by more than EPS_ZERO."""
    return bool(np.dot(ray_direction, intersection_point) - np.dot(ray_direction, ray_position) > EPS_ZERO)


def ray_z_cylinder(length, radius, ray_origin, ray_direction):
    """(points, distances) of a ray with a capped cylinder along z, centred on the origin, nearest first -- the return form
    of the reference's helper of this name (geometry/utils.py:131-362); the crossings are `Cylinder`'s (the kernel's
    arithmetic, distance > EPS_ZERO), `([], [])` for a miss."""
    o, d = np.asarray(ray_origin, dtype=np.float64), np.asarray(ray_direction, dtype=np.float64)
    distances = sorted(Cylinder(length, radius)._ray_distances(o, d))
    if not distances:
        return ([], [])
    return tuple(tuple((o + t * d).tolist()) for t in distances), tuple(float(t) for t in distances)


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

    def is_on_surface(self, point):
        raise NotImplementedError

    def contains(self, point):
        raise NotImplementedError

    def normal(self, surface_point):
        raise NotImplementedError

    def is_entering(self, surface_point, direction):
        """A ray at `surface_point` travelling along `direction` goes INTO the shape
        (reference geometry protocol, e.g. sphere.py:78-86)."""
        return bool(np.dot(self.normal(surface_point), np.asarray(direction, dtype=np.float64)) < 0.0)


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

    def normal(self, surface_point):
        point = surface_point
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

    def normal(self, surface_point):
        point = surface_point
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

    def is_on_surface(self, point):
        p = np.asarray(point, dtype=np.float64)
        r, half = math.hypot(p[0], p[1]), 0.5 * self.length
        on_cap = abs(abs(p[2]) - half) < EPS_ZERO and r <= self.radius + EPS_ZERO
        on_side = abs(r - self.radius) < EPS_ZERO and abs(p[2]) <= half + EPS_ZERO
        return bool(on_cap or on_side)

    def normal(self, surface_point):
        point = surface_point
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


class Mesh(Geometry):
    """Closed triangle mesh (reference pvtrace/geometry/mesh.py:11-24: `Mesh(trimesh, material)`).

    `trimesh` is anything with `.vertices` (V,3) and `.faces` (F,3) -- a `trimesh.Trimesh`
    works, so does a `(vertices, faces)` pair.  As in the reference the vertices are re-centred
    on the centre of mass (mesh.py:17).  The surface must be closed and consistently wound
    (every edge shared by two faces, once per direction); an inward winding is flipped.

    Ray queries follow the device semantics: every crossing with distance > EPS_ZERO counts,
    the normal of a crossing is the face normal (mesh.py:63-86), and the watertight
    ray/triangle test of the kernel (see `pvtrace_amd.mesh.ray_triangle_distances`) decides
    which face owns a ray through an edge or a vertex.
    """

    def __init__(self, trimesh=None, material=None, *, recenter=True):
        super(Mesh, self).__init__(material=material)
        from pvtrace_amd import mesh as M

        if hasattr(trimesh, "vertices") and hasattr(trimesh, "faces"):
            vertices, faces = trimesh.vertices, trimesh.faces
        else:
            vertices, faces = trimesh
        vertices = np.array(vertices, dtype=np.float64).reshape(-1, 3)
        faces = np.array(faces, dtype=np.int32).reshape(-1, 3)
        if faces.size == 0 or faces.min() < 0 or faces.max() >= len(vertices):
            raise GeometryError("Mesh faces must index into the vertex array.")
        if not M.is_watertight(faces):
            raise GeometryError("Mesh must be a closed, consistently wound surface.")
        if M.signed_volume(vertices, faces) < 0.0:
            faces = faces[:, ::-1].copy()
        if recenter:
            vertices = vertices - M.center_of_mass(vertices, faces)
        self.vertices = vertices
        self.faces = faces
        self.face_normals = M.face_normals(vertices, faces)

    # -- constructors ------------------------------------------------------
    @classmethod
    def icosphere(cls, subdivisions=3, radius=1.0, material=None):
        from pvtrace_amd import mesh as M

        return cls(M.icosphere(subdivisions, radius), material=material)

    @classmethod
    def box(cls, size, material=None):
        from pvtrace_amd import mesh as M

        return cls(M.box_mesh(size), material=material)

    @classmethod
    def from_file(cls, path, material=None):
        from pvtrace_amd import mesh as M

        return cls(M.load_stl(path), material=material)

    # -- queries -------------------------------------------------------------
    def _crossings(self, origin, direction):
        from pvtrace_amd import mesh as M

        return M.ray_triangle_distances(self.vertices, self.faces, origin, direction)

    def _ray_distances(self, o, d):
        return list(self._crossings(o, d)[0])

    def contains(self, point):
        """Strictly inside: odd number of crossings along +z and not on the surface."""
        if self.is_on_surface(point):
            return False
        ts, _ = self._crossings(np.asarray(point, dtype=np.float64), np.array([0.0, 0.0, 1.0]))
        return bool(len(ts) % 2 == 1)

    def _closest_face(self, point):
        """(distance, face) of the closest point of the surface."""
        p = np.asarray(point, dtype=np.float64)
        tri = self.vertices[self.faces]
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        # closest point on each triangle (Ericson, Real-Time Collision Detection 5.1.5), vectorised
        ab, ac, ap = b - a, c - a, p - a
        d1, d2 = np.einsum("ij,ij->i", ab, ap), np.einsum("ij,ij->i", ac, ap)
        bp = p - b
        d3, d4 = np.einsum("ij,ij->i", ab, bp), np.einsum("ij,ij->i", ac, bp)
        cp = p - c
        d5, d6 = np.einsum("ij,ij->i", ab, cp), np.einsum("ij,ij->i", ac, cp)
        vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
        with np.errstate(divide="ignore", invalid="ignore"):
            denom = va + vb + vc
            v, w = vb / denom, vc / denom
            q = a + ab * v[:, None] + ac * w[:, None]                       # interior
            sel = (vc <= 0) & (d1 >= 0) & (d3 <= 0)                            # edge ab
            q = np.where(sel[:, None], a + ab * (d1 / (d1 - d3))[:, None], q)
            sel = (vb <= 0) & (d2 >= 0) & (d6 <= 0)                            # edge ac
            q = np.where(sel[:, None], a + ac * (d2 / (d2 - d6))[:, None], q)
            sel = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)              # edge bc
            q = np.where(sel[:, None], b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None], q)
        q = np.where(((d1 <= 0) & (d2 <= 0))[:, None], a, q)
        q = np.where(((d3 >= 0) & (d4 <= d3))[:, None], b, q)
        q = np.where(((d6 >= 0) & (d5 <= d6))[:, None], c, q)
        dist = np.sqrt(np.einsum("ij,ij->i", q - p, q - p))
        k = int(np.argmin(dist))
        return float(dist[k]), k

    def is_on_surface(self, point):
        return bool(self._closest_face(point)[0] < EPS_ZERO)

    def normal(self, surface_point):
        point = surface_point
        dist, face = self._closest_face(point)
        if not dist < EPS_ZERO:
            raise GeometryError("Point is not on surface.")
        return tuple(self.face_normals[face].tolist())

    def is_entering(self, surface_point, direction):
        point = surface_point
        return bool(np.dot(self.normal(point), np.asarray(direction, dtype=np.float64)) < 0.0)
