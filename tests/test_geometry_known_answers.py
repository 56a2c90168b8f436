"""Known answers of the reference's geometry unit tests (tests/test_box.py, test_sphere.py,
test_cylinder.py, test_geometry_utils.py) for the host-side shape queries: constants below are
those files' expected values.  (Box is analytic here, like the reference KERNEL; the reference's
Python Box goes through trimesh.)"""
import numpy as np

from pvtrace_amd import Box, Cylinder, Sphere


def unit(v):
    v = np.asarray(v, dtype=float)
    return tuple((v / np.linalg.norm(v)).tolist())


def test_box_known_answers():                                    # tests/test_box.py:14-66
    b = Box(size=(1, 1, 1))
    for p in ((0.5, 0, 0), (0, 0.5, 0), (0, 0, 0.5), (-0.5, 0, 0), (0, -0.5, 0), (0, 0, -0.5)):
        assert b.is_on_surface(p) is True
        assert np.allclose(b.normal(p), np.sign(p))
    assert b.is_on_surface((0, 0, 0)) is False and b.is_on_surface((0.501, 0, 0)) is False
    assert Box((1.0, 1.0, 0.02)).is_on_surface((0.06608370507653762, 0.5, -0.007798573829629238)) is True
    assert Box((1.0, 3.0, 0.02)).is_on_surface((-0.5, -0.2415708917159319, -0.008736363958583498)) is True
    assert b.contains((0, 0, 0)) is True and b.contains((0, 0, 0.5)) is False and b.contains((0, 0, 1.0)) is False
    assert b.intersections((-2.0, 0.0, 0.0), (1.0, 0.0, 0.0)) == ((-0.5, 0.0, 0.0), (0.5, 0.0, 0.0))
    for p, d, want in (((0.5, 0, 0), (-1, 0, 0), True), ((0.5, 0, 0), (1, 0, 0), False),
                       ((0, 0.5, 0), (0, -1, 0), True), ((0, 0.5, 0), (0, 1, 0), False),
                       ((0, 0, 0.5), (0, 0, -1), True), ((0, 0, 0.5), (0, 0, 1), False)):
        assert b.is_entering(p, d) is want


def test_sphere_known_answers():                                 # tests/test_sphere.py:14-42
    s = Sphere(radius=1)
    assert s.is_on_surface((0, 0, 1.0)) and not s.is_on_surface((0, 0, 0))
    assert not s.contains((0, 0, 2.0)) and not s.contains((0, 0, 1.0)) and s.contains((0, 0, 0))
    assert s.intersections((-2.0, 0.0, 0.0), (1.0, 0.0, 0.0)) == ((-1.0, 0.0, 0.0), (1.0, 0.0, 0.0))
    assert np.allclose(s.normal((0, 0, 1.0)), (0, 0, 1.0))
    assert s.is_entering((-1.0, 0, 0), (1.0, 0, 0)) is True and s.is_entering((-1.0, 0, 0), (-1.0, 0, 0)) is False


def test_cylinder_known_answers():                               # tests/test_cylinder.py:14-73
    c = Cylinder(length=1.0, radius=1.0)
    for p, want in (((0, 0, 0.5), True), ((0, 0, -0.5), True), ((0, 1.0, 0), True), ((-1.0, 0, 0), True),
                    ((0, 0, 0.6), False), ((0, 1.1, 0), False)):
        assert c.is_on_surface(p) is want
        assert c.contains(p) is False
    assert c.contains((0, 0, 0)) and c.contains((0.25, 0.25, 0.25))
    points = c.intersections((-2, 0.2, 0.0), unit((1.0, 0.2, -0.2)))
    assert np.allclose(points, ((-0.9082895433880116, 0.41834209132239775, -0.2183420913223977), (0.5, 0.7, -0.5)))
    assert np.allclose(c.normal((0, 0, 0.5)), (0, 0, 1)) and np.allclose(c.normal((0, 0, -0.5)), (0, 0, -1))
    assert np.allclose(c.normal((0, 1.0, 0)), (0, 1, 0)) and np.allclose(c.normal((0, -1.0, 0)), (0, -1, 0))
    for p, d, want in (((0, 0, 0.5), (1, 1, -1), True), ((0, 0, 0.5), (1, 1, 1), False),
                       ((0, 0, -0.5), (1, 1, 1), True), ((0, 0, -0.5), (1, 1, -1), False),
                       ((0, 1.0, 0), (1, -1, 1), True), ((0, 1.0, 0), (1, 1, 1), False),
                       ((-1.0, 0, 0), (1, 1, 1), True), ((-1.0, 0, 0), (-1, 1, 1), False)):
        assert c.is_entering(p, unit(d)) is want


def test_ray_cylinder_known_points():                            # tests/test_geometry_utils.py:63-101
    c = Cylinder(length=1.0, radius=1.0)
    assert np.allclose(c.intersections((0.2, 0.2, -1), unit((0, 0, 1.0))), ((0.2, 0.2, -0.5), (0.2, 0.2, 0.5)))
    assert np.allclose(c.intersections((-2, 0.2, 0.0), unit((1.0, 0.2, 0.2))),
                       ((-0.9082895433880116, 0.41834209132239775, 0.2183420913223977), (0.5, 0.7, 0.5)))
    touching = c.intersections((0.0, 0.0, -1.5), unit((0.0, 1.0, 1.0)))
    assert np.allclose(touching[0], (0.0, 1.0, -0.5))
    # the same answers from the C referee's intersector (what the kernel restates)
    from oracle import oracle as O

    ts = O.intersect(2, (1.0, 1.0), (-2, 0.2, 0.0), unit((1.0, 0.2, -0.2)))
    hit = np.array((-2, 0.2, 0.0)) + np.sort(ts)[:, None] * np.array(unit((1.0, 0.2, -0.2)))
    assert np.allclose(hit, ((-0.9082895433880116, 0.41834209132239775, -0.2183420913223977), (0.5, 0.7, -0.5)))


def test_distribution_end_points_and_step_tables():              # tests/test_distibution.py:9-42
    from pvtrace_amd.material import Distribution, bandgap, thermodynamic_emission

    x = np.linspace(400, 1010, 2000)
    ems = thermodynamic_emission(np.column_stack((x, bandgap(x, 600, 1000))), T=300, mu=0.1)
    dist = Distribution(ems[:, 0], ems[:, 1])
    assert np.isclose(dist.sample(0), x.min()) and np.isclose(dist.sample(1), x.max())
    assert np.isclose(dist.lookup(dist.sample(0)), 0.0)
    nmedge, nmmin, nmmax, spacing = 600.0, 400.0, 800.0, 1.0
    xs = np.arange(nmmin, nmmax + spacing, spacing)
    step = Distribution(xs, bandgap(xs, nmedge, 1.0), hist=True)
    assert np.isclose(step.sample(0), nmmin) and np.isclose(step.sample(1), nmedge)
    assert 0.0 <= step.lookup(nmmin) <= step.lookup(nmmin + spacing) and step.lookup(nmmax) == 1.0
    values = step.sample(np.linspace(step.lookup(599 - spacing), step.lookup(600 + spacing), 10000))
    assert len(set(np.asarray(values).tolist())) == 3


def test_vector_and_tolerance_helpers_known_answers():           # tests/test_geometry_utils.py:24-62
    from pvtrace_amd.geometry import (EPS_ZERO, allinrange, angle_between, close_to_zero, distance_between, flip, floats_close,
                                      intersection_point_is_ahead, magnitude, norm, points_equal, ray_z_cylinder,
                                      smallest_angle_between)

    zero = 0.0
    assert close_to_zero(zero) and close_to_zero(zero + 0.9 * EPS_ZERO) and close_to_zero(zero - 0.9 * EPS_ZERO)
    assert not close_to_zero(zero + EPS_ZERO) and not close_to_zero(zero - EPS_ZERO)
    a = 0.3711
    assert not floats_close(a, a + EPS_ZERO) and not floats_close(a, a - EPS_ZERO)
    assert floats_close(a, a + 0.9 * EPS_ZERO) and floats_close(a, a - 0.9 * EPS_ZERO) and floats_close(a, a)
    v = (1.0, 1.0, 1.0)
    assert np.isclose(magnitude(v), np.sqrt(3.0)) and np.allclose(norm(v), np.array(v) / magnitude(v))
    normal, vector = norm((1.0, 0.0, 0.0)), norm((1.0, 0.0, 0.0))
    assert angle_between(normal, vector) == 0.0 and angle_between(-normal, vector) == np.pi
    assert np.isclose(angle_between(norm((1.0, 1.0, 0.0)), norm((1.0, 0.0, 0.0))), np.pi / 4.0)
    v1, v2 = norm((1.0, 1.0, 0.0)), norm((-1.0, 0.0, 0.0))
    assert np.dot(v1, v2) < 0.0 and np.isclose(angle_between(v1, v2), np.pi - np.pi / 4.0)
    assert np.isclose(smallest_angle_between(v1, v2), np.pi - np.pi / 4.0)
    assert np.array_equal(flip((1.0, -2.0, 0.0)), (-1.0, 2.0, 0.0)) and distance_between((0, 0, 0), (3, 4, 0)) == 5.0
    assert points_equal((1.0, 2.0, 3.0), (1.0, 2.0, 3.0 + 0.5 * EPS_ZERO)) and not points_equal((1.0, 2.0, 3.0), (1.0, 2.0, 3.1))
    assert allinrange(np.array([1.0, 2.0]), (1.0, 2.0)) and not allinrange(2.5, (1.0, 2.0)) and allinrange(1.5, (1.0, 2.0))
    assert intersection_point_is_ahead((0, 0, 0), (0, 0, 1), (0, 0, 2)) and not intersection_point_is_ahead((0, 0, 0), (0, 0, 1), (0, 0, -2))
    points, distances = ray_z_cylinder(1.0, 1.0, (0.2, 0.2, -1), unit((0, 0, 1.0)))     # :63-101, both caps
    assert np.allclose(points, ((0.2, 0.2, -0.5), (0.2, 0.2, 0.5))) and np.allclose(distances, (0.5, 1.5))
    assert ray_z_cylinder(1.0, 1.0, (5.0, 5.0, -1.0), unit((0, 0, 1.0))) == ([], [])
