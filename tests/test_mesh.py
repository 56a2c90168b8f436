"""Triangle meshes: host geometry queries, the watertight ray/triangle test
and the oracle's mesh path.  The reference engine has no mesh path (compiler.py:220-223) and
its Python tracer's mesh arithmetic is trimesh's (absent here), so parity is pinned through
properties: the known answers of the reference's tests/test_mesh.py that do not depend on
trimesh's t = 0 convention, a mesh box against the analytic box, watertightness, and GPU ==
oracle bit for bit (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd import Material, Mesh, Node, Scene, Sphere, Light
from pvtrace_amd import mesh as M
from pvtrace_amd.common import GeometryError
from pvtrace_amd.engine import compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes


# -- known answers of the reference's tests/test_mesh.py (unit icosphere) -----------------
def test_icosphere_intersections_like_the_reference_tests():
    m = Mesh.icosphere()
    assert len(m.faces) == 1280
    # tests/test_mesh.py:13-18 (origin ON the surface: the kernel drops t <= EPS, trimesh keeps it)
    assert np.allclose(m.intersections((0, 0, -1), (0, 0, 1)), ((0.0, 0.0, 1.0),))
    assert np.allclose(m.intersections((0, 0, -2), (0, 0, 1)), ((0.0, 0.0, -1.0), (0.0, 0.0, 1.0)))
    assert np.allclose(m.intersections((0, 0, 2), (0, 0, -1)), ((0.0, 0.0, 1.0), (0.0, 0.0, -1.0)))
    assert len(m.intersections((0, 0, -1.1), (0, 0, -1))) == 0          # :44-48
    assert m.contains((0, 0, -1.1)) is False                            # :50-54
    assert m.contains((0, 0, 0.9)) is True                              # :56-60
    assert m.contains((0, 0, 1.0)) is False                             # :62-66
    assert m.is_on_surface((0, 0, -1.1)) is False                       # :68-72
    assert m.is_on_surface((0, 0, 0.9)) is False                        # :74-78
    assert m.is_on_surface((0, 0, 1.0)) is True                         # :80-84
    assert m.is_entering((0, 0, -1), (0, 0, 1)) is True                 # :86-91
    assert m.is_entering((0, 0, -1), (0, 0, -1)) is False               # :93-98
    with pytest.raises(GeometryError):                                  # :107-127
        m.is_entering((0, 0, -1.1), (0, 0, -1))
    with pytest.raises(GeometryError):
        m.is_entering((0, 0, -0.9), (0, 0, -1))


def test_mesh_is_recentred_wound_outwards_and_must_be_closed():
    v, f = M.icosphere(1, 2.0)
    shifted = Mesh((v + np.array([3.0, -1.0, 0.5]), f))
    assert np.allclose(shifted.vertices, v, atol=1e-12)                 # geometry/mesh.py:17
    flipped = Mesh((v, f[:, ::-1]))
    assert M.signed_volume(flipped.vertices, flipped.faces) > 0
    assert np.all(np.einsum("ij,ij->i", flipped.face_normals, flipped.vertices[flipped.faces[:, 0]]) > 0)
    with pytest.raises(GeometryError):
        Mesh((v, f[:-1]))                                               # a hole
    class Duck:                                                         # anything trimesh-shaped
        vertices, faces = v, f
    assert len(Mesh(Duck).faces) == len(f)


def test_stl_round_trip(tmp_path):
    v, f = M.icosphere(2, 1.5)
    M.save_stl(str(tmp_path / "ball.stl"), v, f)
    v2, f2 = M.load_stl(str(tmp_path / "ball.stl"))
    assert len(v2) == len(v) and len(f2) == len(f) and M.is_watertight(f2)
    assert np.isclose(M.signed_volume(v2, f2), M.signed_volume(v, f), rtol=1e-6)   # float32 file
    ascii_path = tmp_path / "box.stl"
    bv, bf = M.box_mesh((1, 2, 3))
    with open(ascii_path, "w") as fp:
        fp.write("solid box\n")
        for tri in bv[bf]:
            fp.write("facet normal 0 0 0\nouter loop\n")
            for p in tri:
                fp.write("vertex %r %r %r\n" % tuple(p.tolist()))
            fp.write("endloop\nendfacet\n")
        fp.write("endsolid box\n")
    m = Mesh.from_file(str(ascii_path))
    assert len(m.faces) == 12 and np.isclose(M.signed_volume(m.vertices, m.faces), 6.0)


# -- the ray/triangle test ------------------------------------------------------------------
def test_host_and_oracle_triangle_tests_are_bit_identical():
    m = Mesh.icosphere(2, 1.3)
    rng = np.random.default_rng(1)
    for _ in range(1500):
        o = rng.uniform(-2, 2, 3)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        t1, f1 = M.ray_triangle_distances(m.vertices, m.faces, o, d)
        t2, f2 = O.mesh_hits(m.vertices, m.faces, o, d)
        k = np.lexsort((f2, t2))
        assert np.array_equal(t1, t2[k]) and np.array_equal(f1, f2[k])


@pytest.mark.parametrize("make", [lambda: M.icosphere(2, 1.0), lambda: M.box_mesh((2.0, 1.0, 0.5))])
def test_rays_through_edges_and_vertices_cross_the_surface_exactly_once(make):
    """Watertightness incl. exact zeros of the edge functions: from an interior point every ray
    leaves through exactly one face, from outside it crosses an even number of times -- also
    when aimed EXACTLY at a vertex, an edge midpoint or along a coordinate axis."""
    v, f = make()
    mesh = Mesh((v, f), recenter=False)
    targets = [mesh.vertices, 0.5 * (mesh.vertices[mesh.faces[:, 0]] + mesh.vertices[mesh.faces[:, 1]]),
               mesh.vertices[mesh.faces].mean(axis=1), np.eye(3), -np.eye(3)]
    targets = np.concatenate(targets)
    for origin in (np.zeros(3), np.array([0.1, -0.05, 0.02])):
        for p in targets:
            d = p - origin
            d = d / np.linalg.norm(d)
            ts, _ = O.mesh_hits(mesh.vertices, mesh.faces, origin, d)
            assert len(ts) == 1, (origin, p, ts)
    outside = np.array([0.0, 0.0, -5.0])
    for p in targets:
        d = p - outside
        d = d / np.linalg.norm(d)
        ts, _ = O.mesh_hits(mesh.vertices, mesh.faces, outside, d)
        assert len(ts) % 2 == 0, (p, ts)     # (a graze of the silhouette counts 0 or 2, never 1)


# -- the oracle's mesh path -----------------------------------------------------------------
def test_compiled_tables_pool_the_meshes():
    c = compile_scene(scenes.mesh_gem())
    assert c.geom_type.tolist() == [3, 3, 1]
    assert c.mesh_face_start.tolist() == [0, 80, 0] and c.mesh_face_count.tolist() == [80, 320, 0]
    assert c.n_mesh_faces == 400 and c.mesh_faces.shape == (400, 3) and c.mesh_normals.shape == (400, 3)
    assert c.mesh_faces[80:].min() == c.n_mesh_vertices - 162          # second mesh indexes its own vertices
    assert np.allclose(np.linalg.norm(c.mesh_normals, axis=1), 1.0)


def test_mesh_box_traces_like_the_analytic_box():
    """Same scene, slab as 12 triangles: identical event sequences for every photon, hit points
    equal to rounding (the crossing distance comes from a different but equivalent formula)."""
    a, b = scenes.lsc_equivalent(), scenes.mesh_lsc()
    ca, cb = compile_scene(a), compile_scene(b)
    n, me = 6000, 64
    pos, dirs, wl, _ = emit_bundle(a, n, seed=5)
    ra = O.trace_bundle(ca, pos, dirs, wl, 7, 1000, me, 0, 4, 1)
    rb = O.trace_bundle(cb, pos, dirs, wl, 7, 1000, me, 0, 4, 1)
    for key in ("counts", "kind", "hit", "container", "adjacent", "component", "source", "rec_distinct",
                "rec_crossings", "rec_bins"):
        assert np.array_equal(ra[key], rb[key]), key
    assert np.array_equal(ra["normal"], rb["normal"], equal_nan=True)
    assert np.array_equal(ra["wavelength"], rb["wavelength"], equal_nan=True)
    assert np.nanmax(np.abs(ra["position"] - rb["position"])) < 1e-11
    assert np.nanmax(np.abs(ra["travelled"] - rb["travelled"])) < 1e-10


def test_fine_icosphere_approaches_the_analytic_sphere():
    def ball(geometry):
        world = Node(name="world", geometry=Sphere(10.0, material=Material(refractive_index=1.0)))
        Node(name="ball", parent=world, geometry=geometry).location = (0.0, 0.0, 2.0)
        from pvtrace_amd.engine import Recorder
        world.children[0].recorders = [Recorder("in", event="entering"), Recorder("refl", event="reflected")]
        Node(name="lamp", parent=world, light=Light(name="lamp"))
        return Scene(world)
    glass = Material(refractive_index=1.5)
    n = 20000
    out = []
    for g in (Sphere(1.0, material=glass), Mesh.icosphere(3, 1.0, material=glass)):
        sc = ball(g)
        pos, dirs, wl, _ = emit_bundle(sc, n, seed=9)
        pos = pos + np.random.default_rng(2).uniform(-0.6, 0.6, (n, 3)) * np.array([1.0, 1.0, 0.0])
        r = O.trace_bundle(compile_scene(sc), pos, dirs, wl, 3, 1000, 32, 0, 8, 0)
        out.append(r["rec_distinct"][:2] / n)
    (in_s, refl_s), (in_m, refl_m) = out
    assert abs(in_s - in_m) < 0.01 and abs(refl_s - refl_m) < 0.01, out


def test_host_built_bvh_is_a_proper_depth_first_tree():
    """The library's BVH builder (csrc/pvt_bvh.h) checked on the host: every face in exactly one
    leaf and inside all enclosing boxes, skip links nest properly, the copy of the top levels for LDS walks
    like the plain tree.  One triangle per leaf."""
    from pvtrace_amd.engine import native

    scene = scenes.mesh_gem()
    compiled = compile_scene(scene)
    nodes, leaves, depth = native.mesh_bvh_check(compiled, 1)          # 320-face gem
    assert leaves == 320 and nodes == 2 * 320 - 1 and 9 <= depth <= 11
    nodes, leaves, depth = native.mesh_bvh_check(compiled, 0)          # 80-face world
    assert leaves == 80 and nodes == 159
    small = compile_scene(scenes.mesh_lsc())
    nodes, leaves, depth = native.mesh_bvh_check(small, 1)             # 12 faces
    assert leaves == 12 and nodes == 23 and 4 <= depth <= 7              # (the SAH split need not be even)
    big = Node(name="w", geometry=Mesh.icosphere(5, 3.0, material=Material(1.0)))
    nodes, leaves, depth = native.mesh_bvh_check(compile_scene(Scene(big)), 0)
    assert leaves == 20480 and 15 <= depth <= 20   # (balanced: 15; the SAH tree is a little deeper)
    with pytest.raises(Exception):
        native.mesh_bvh_check(compiled, 2)                              # the analytic sphere


def test_the_lds_copy_of_the_tree_tops_is_shared_out_between_meshes_of_different_size():
    """pvt_mesh_bvh_check also stages ALL meshes of the scene together (pvt_bvh.h: stage_top) at five budgets -- none, a
    handful of records, more than the small trees need, everything -- and replays every tree through its cursors: the
    records of the plain tree in order, the same successor after a hit and after a miss."""
    from pvtrace_amd.engine import native

    world = Node(name="w", geometry=Mesh.icosphere(1, 20.0, material=Material(1.0)))
    for k, (sub, radius) in enumerate([(4, 1.0), (0, 0.4), (2, 0.7), (1, 0.2)]):
        n = Node(name=f"m{k}", parent=world, geometry=Mesh.icosphere(sub, radius, material=Material(1.5)))
        n.location = (3.0 * k - 4.0, 0.5 * k, 0.0)
    compiled = compile_scene(Scene(world))
    faces = [80, 5120, 20, 320, 80]
    for node, count in enumerate(faces):
        nodes, leaves, depth = native.mesh_bvh_check(compiled, node)
        assert leaves == count and nodes == 2 * count - 1


def test_non_convex_mesh_holds_a_ray_that_crosses_it_three_times():
    """A ray that starts inside one arm of the L-prism and heads across the notch crosses the mesh's
    surface three times before the world's.  The reference's container rule -- the nearest node crossed
    exactly ONCE (pvtrace/algorithm/photon_tracer.py:26-57, _kernel.pyx:696-706) -- would put such a ray
    in the world; a mesh therefore holds a ray when it is crossed an odd number of times (what
    `Mesh.contains`, geometry/mesh.py:29-32, answers), and beyond its surface lies the next node that
    holds the ray, not the second-nearest crossing (which is the same mesh again)."""
    from oracle import oracle as O
    from pvtrace_amd.engine import compile_scene
    from tests import scenes

    prism = scenes.l_prism_mesh()
    assert prism.contains((1.6, 0.5, 0.5)) and prism.contains((0.5, 1.6, 0.5)) and not prism.contains((1.6, 1.6, 0.5))
    world = Node(name="world", geometry=Sphere(10.0, material=Material(refractive_index=1.0)))
    Node(name="L", parent=world, geometry=scenes.l_prism_mesh(Material(refractive_index=1.0)))   # index-matched: straight lines
    Node(name="lamp", parent=world, light=Light())
    scene = Scene(world)
    compiled = compile_scene(scene)
    # from (1.6, 0.5, 0.5) towards -x+y: leaves the x-arm through the notch wall x... crosses y = 1 at x = 1.1
    start, direction = np.array([[1.6, 0.5, 0.5]]), np.array([[-np.sqrt(0.5), np.sqrt(0.5), 0.0]])
    out = O.trace_bundle(compiled, start, direction, np.array([555.0]), 1, 1000, 16, 0, 1, 1)
    n = int(out["counts"][0])
    kinds = out["kind"][:n].tolist()
    ids = list(zip(out["hit"][:n].tolist(), out["container"][:n].tolist(), out["adjacent"][:n].tolist()))
    assert kinds == [0, 2, 2, 2, 7]                      # GENERATE, out of the arm, into the other arm, out again, EXIT
    assert ids[1] == (1, 1, 0)                           # inside L, the world beyond (NOT L again)
    assert ids[2] == (1, 0, 1)                           # in the notch: the world holds the ray, L ahead
    assert ids[3] == (1, 1, 0)
    assert np.allclose(out["position"][1], (1.1, 1.0, 0.5)) and np.allclose(out["position"][2], (1.0, 1.1, 0.5))
    assert np.allclose(out["position"][3], (0.1, 2.0, 0.5))
    # the same history from the Python tracer restatement (walks the node tree, no tables)
    from oracle.py_tracer import next_hit
    from pvtrace_amd.light import Ray

    hit, (container, adjacent), point, distance = next_hit(scene, Ray(position=(1.6, 0.5, 0.5), direction=tuple(direction[0]), wavelength=555.0))
    assert (hit.name, container.name, adjacent.name) == ("L", "L", "world") and np.allclose(point, (1.1, 1.0, 0.5))
