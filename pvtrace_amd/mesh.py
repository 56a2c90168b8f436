"""Triangle-mesh helpers for `geometry.Mesh`: primitives, STL reading, topology checks and the
host restatement of the device ray/triangle test.

The reference wraps a `trimesh.Trimesh` (pvtrace/geometry/mesh.py:11-24) and its scene spec
loads mesh files through `trimesh.exchange.load` (pvtrace/cli/parse.py:130-138).  trimesh is
not a dependency here: a mesh is just `(vertices (V,3) f64, faces (F,3) i32)`, and anything
exposing `.vertices` / `.faces` (a Trimesh included) is accepted by `geometry.Mesh`.
"""
import struct

import numpy as np

EPS_ZERO = 2.220446049250313e-13


# ---------------------------------------------------------------------------
# primitives

def box_mesh(size):
    """12-triangle box centred on the origin, outward winding."""
    sx, sy, sz = (0.5 * float(s) for s in size)
    v = np.array([[x, y, z] for x in (-sx, sx) for y in (-sy, sy) for z in (-sz, sz)], dtype=np.float64)
    # vertex index = 4*ix + 2*iy + iz
    quads = [
        (0, 1, 3, 2),  # -x
        (4, 6, 7, 5),  # +x
        (0, 4, 5, 1),  # -y
        (2, 3, 7, 6),  # +y
        (0, 2, 6, 4),  # -z
        (1, 5, 7, 3),  # +z
    ]
    faces = []
    for a, b, c, d in quads:
        faces += [(a, b, c), (a, c, d)]
    return v, np.array(faces, dtype=np.int32)


def icosphere(subdivisions=3, radius=1.0):
    """Unit icosahedron subdivided `subdivisions` times and pushed onto the sphere (the shape
    `trimesh.creation.icosphere` makes; the reference's mesh tests use it, tests/test_mesh.py:11).
    20 * 4**subdivisions faces; after one subdivision (0, 0, +-radius) are vertices."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
         (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
         (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = [tuple(x) for x in f]
    for _ in range(int(subdivisions)):
        cache = {}

        def midpoint(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nxt = []
        for a, b, c in faces:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            nxt += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nxt
    return np.array(verts, dtype=np.float64) * float(radius), np.array(faces, dtype=np.int32)


# ---------------------------------------------------------------------------
# IO

def merge_vertices(triangles, digits=12):
    """(F,3,3) triangle soup -> (vertices, faces) with coincident corners merged (rounded to
    `digits` significant decimals of the bounding size, as STL stores no connectivity)."""
    tri = np.asarray(triangles, dtype=np.float64).reshape(-1, 3)
    scale = float(np.max(np.abs(tri))) if tri.size else 1.0
    key = np.round(tri / (scale or 1.0), digits)
    _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)                 # keep first-appearance order: deterministic
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    vertices = tri[first[order]]
    faces = rank[np.asarray(inverse).reshape(-1)].reshape(-1, 3).astype(np.int32)
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return vertices, faces[keep]


def load_stl(path):
    """Read a binary or ASCII STL file -> (vertices, faces)."""
    with open(path, "rb") as fp:
        raw = fp.read()
    if len(raw) >= 84:
        (count,) = struct.unpack_from("<I", raw, 80)
        if len(raw) == 84 + 50 * count:                       # binary layout is exact
            rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                                offset=84, count=count)
            return merge_vertices(rec["v"].astype(np.float64))
    text = raw.decode("ascii", errors="replace").split()
    tri = [float(text[i + k]) for i, tok in enumerate(text) if tok == "vertex" for k in (1, 2, 3)]
    if not tri or len(tri) % 9:
        raise ValueError(f"{path}: not a readable STL file")
    return merge_vertices(np.array(tri).reshape(-1, 3, 3))


def save_stl(path, vertices, faces):
    """Write a binary STL (test fixtures, round trips)."""
    v = np.asarray(vertices, dtype=np.float64)[np.asarray(faces)]
    n = face_normals(vertices, faces)
    rec = np.zeros(len(v), dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["n"], rec["v"] = n, v
    with open(path, "wb") as fp:
        fp.write(b"pvtrace_amd".ljust(80, b"\0"))
        fp.write(struct.pack("<I", len(v)))
        fp.write(rec.tobytes())


# ---------------------------------------------------------------------------
# topology / mass properties

def face_normals(vertices, faces):
    v = np.asarray(vertices, dtype=np.float64)[np.asarray(faces)]
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    mag = np.sqrt(np.einsum("ij,ij->i", n, n))
    if np.any(mag == 0.0):
        raise ValueError("mesh has a zero-area face")
    return n / mag[:, None]


def signed_volume(vertices, faces):
    v = np.asarray(vertices, dtype=np.float64)[np.asarray(faces)]
    return float(np.einsum("ij,ij->i", v[:, 0], np.cross(v[:, 1], v[:, 2])).sum() / 6.0)


def center_of_mass(vertices, faces):
    """Volume centroid of a closed surface (what trimesh calls `center_mass`, which the
    reference subtracts from the vertices, geometry/mesh.py:17); area centroid if the signed
    volume vanishes."""
    v = np.asarray(vertices, dtype=np.float64)[np.asarray(faces)]
    vol = np.einsum("ij,ij->i", v[:, 0], np.cross(v[:, 1], v[:, 2])) / 6.0
    total = float(vol.sum())
    if abs(total) > 0.0:
        return (vol[:, None] * v.sum(axis=1) / 4.0).sum(axis=0) / total
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    area = 0.5 * np.sqrt(np.einsum("ij,ij->i", n, n))
    return (area[:, None] * v.mean(axis=1)).sum(axis=0) / float(area.sum())


def is_watertight(faces):
    """Every undirected edge is used by exactly two faces, once in each direction (closed,
    consistently wound).  The container rule of the tracer needs exactly this."""
    f = np.asarray(faces, dtype=np.int64)
    if f.size == 0:
        return False
    e = np.concatenate((f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]))
    directed = e[:, 0] * (int(f.max()) + 1) + e[:, 1]
    if np.unique(directed).size != directed.size:
        return False                                   # an edge used twice in the same direction
    reverse = e[:, 1] * (int(f.max()) + 1) + e[:, 0]
    return bool(np.array_equal(np.sort(directed), np.sort(reverse)))


# ---------------------------------------------------------------------------
# host restatement of the device ray/triangle test (csrc/pvt_trace_kernel.h, mesh branch of the
# node loop; oracle/pvt_oracle.c `tri_hit`): watertight shear + edge functions + half-plane tie rule

def ray_triangle_distances(vertices, faces, origin, direction):
    """(t, face index) of every crossing with t > EPS_ZERO, sorted by (t, face)."""
    o = np.asarray(origin, dtype=np.float64)
    d = np.asarray(direction, dtype=np.float64)
    mag = np.abs(d)
    kz = 0 if (mag[0] >= mag[1] and mag[0] >= mag[2]) else (1 if mag[1] >= mag[2] else 2)
    kx, ky = (kz + 1) % 3, (kz + 2) % 3
    if d[kz] < 0.0:
        kx, ky = ky, kx
    sx, sy, sz = d[kx] / d[kz], d[ky] / d[kz], 1.0 / d[kz]
    tri = np.asarray(vertices, dtype=np.float64)[np.asarray(faces)] - o
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    ax, ay = a[:, kx] - sx * a[:, kz], a[:, ky] - sy * a[:, kz]
    bx, by = b[:, kx] - sx * b[:, kz], b[:, ky] - sy * b[:, kz]
    cx, cy = c[:, kx] - sx * c[:, kz], c[:, ky] - sy * c[:, kz]
    u = cx * by - cy * bx
    v = ax * cy - ay * cx
    w = bx * ay - by * ax
    mixed = ((u < 0) | (v < 0) | (w < 0)) & ((u > 0) | (v > 0) | (w > 0))
    det = u + v + w
    ok = ~mixed & (det != 0.0)
    sg = np.where(det < 0.0, -1.0, 1.0)

    def owned(gx, gy):
        return (gx > 0.0) | ((gx == 0.0) & (gy > 0.0))

    ok &= (u != 0.0) | owned(sg * (cy - by), sg * (bx - cx))
    ok &= (v != 0.0) | owned(sg * (ay - cy), sg * (cx - ax))
    ok &= (w != 0.0) | owned(sg * (by - ay), sg * (ax - bx))
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (u * (sz * a[:, kz]) + v * (sz * b[:, kz]) + w * (sz * c[:, kz])) / det
    ok &= t > EPS_ZERO
    idx = np.nonzero(ok)[0]
    order = np.lexsort((idx, t[idx]))
    return t[idx][order], idx[order]
