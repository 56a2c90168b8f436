"""The node grid of many-node scenes (include/pvtrace_hip.h: pvt_node_grid_plan; kernel GRID variants), checked on
the host without a GPU.

The reference finds the next interaction by intersecting EVERY node and scanning the sorted crossings
(pvtrace/engine/_kernel.pyx:666-714: nearest crossing, second-nearest crossing, nearest node crossed exactly once).
The HIP kernel's grid walk visits only the nodes filed under the cells a ray passes, and stops early.  This module
restates both in a few lines of numpy -- the brute-force scan and the walk (same rules as the kernel: front-to-back
cells, each node once, fold by (t, node), exit once two crossings lie nearer than the end of the cells visited by the
guard; cylinder scenes additionally wait for the container) -- and requires that they agree on (hit, second, container)
for thousands of rays from everywhere in the scene, including rays that start on surfaces and rays along shared faces.
A negative control files the nodes' boxes too small and must be caught."""
import numpy as np
import pytest

from benchmarks.configs import cfg2_lsc, cfg4_nested_cylinders, tiles_lsc
from pvtrace_amd.engine import compile_scene, native
from tests import scenes
from tests.fuzz import random_many_scene

EPS = 2.220446049250313e-13      # _kernel.pyx:29
BOX, SPHERE, CYLINDER = 0, 1, 2


def crossings(tab, node, pos, dirn):
    """Forward crossings (t > EPS) of node `node` by the world ray, the reference's formulas (_kernel.pyx:245-345)."""
    w = tab["world_to_local"][node]
    o = w[:3, :3] @ pos + w[:3, 3]
    d = w[:3, :3] @ dirn
    g = tab["geom_params"][node]
    kind = int(tab["geom_type"][node])
    out = []
    if kind == BOX:
        tmin, tmax = -np.inf, np.inf
        for a in range(3):
            lo, hi = -0.5 * g[a], 0.5 * g[a]
            if abs(d[a]) < 1e-300:
                if o[a] < lo or o[a] > hi:
                    return []
                continue
            ta, tb = (lo - o[a]) / d[a], (hi - o[a]) / d[a]
            tmin, tmax = max(tmin, min(ta, tb)), min(tmax, max(ta, tb))
        if not tmax < tmin:
            out = [t for t in (tmin, tmax) if t > EPS]
    elif kind == SPHERE:
        a, b, c = d @ d, 2.0 * (d @ o), o @ o - g[0] * g[0]
        disc = b * b - 4.0 * a * c
        if not disc < 0.0:
            sq = np.sqrt(disc)
            out = [t for t in ((-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)) if t > EPS]
    else:
        half, radius = 0.5 * g[0], g[1]
        a = d[0] * d[0] + d[1] * d[1]
        if a > 1e-300:
            b = 2.0 * (o[0] * d[0] + o[1] * d[1])
            c = o[0] * o[0] + o[1] * o[1] - radius * radius
            disc = b * b - 4.0 * a * c
            if disc >= 0.0:
                sq = np.sqrt(disc)
                for t in ((-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)):
                    z = o[2] + t * d[2]
                    if -half < z < half and t > EPS:
                        out.append(t)
        if abs(d[2]) > 1e-300:
            for t in ((-half - o[2]) / d[2], (half - o[2]) / d[2]):
                x, y = o[0] + t * d[0], o[1] + t * d[1]
                if x * x + y * y <= radius * radius and t > EPS:
                    out.append(t)
    return out


class Fold:
    """nearest / second-nearest crossing and the nearest node crossed exactly once, by the key (t, node)."""

    def __init__(self):
        self.nhits, self.t1, self.n1, self.t2, self.n2, self.cbest, self.cnode = 0, np.inf, -1, np.inf, -1, np.inf, -1

    def add(self, node, ts):
        for t in ts:
            if self.nhits == 0 or (t, node) < (self.t1, self.n1):
                if self.nhits:
                    self.t2, self.n2 = self.t1, self.n1
                self.t1, self.n1 = t, node
            elif self.n2 < 0 or (t, node) < (self.t2, self.n2):
                self.t2, self.n2 = t, node
            self.nhits += 1
        if len(ts) == 1 and (ts[0], node) < (self.cbest, self.cnode if self.cnode >= 0 else 1 << 30):
            self.cbest, self.cnode = ts[0], node

    def outcome(self):
        """(hit, adjacent-candidate, container) as the step uses them (_kernel.pyx:684-714)."""
        if self.nhits == 0:
            return None
        if self.nhits == 1:
            return self.n1, -1, self.n1
        container = self.cnode if self.cnode >= 0 else self.n1
        return self.n1, (self.n2 if container == self.n1 else self.n1), container


def brute_force(tab, pos, dirn):
    """The reference's scan: every node, ascending."""
    f = Fold()
    for node in range(len(tab["geom_type"])):
        f.add(node, crossings(tab, node, pos, dirn))
    return f


def grid_walk(tab, plan, pos, dirn, stats=None):
    """The kernel's walk (pvt_trace_kernel.h, GRID block), restated."""
    dims, lo, cell, guard, odd = plan["dims"], plan["lo"], plan["cell"], plan["guard"], plan["odd"]
    hi = lo + cell * np.array(dims)
    root = int(tab["root_id"])
    f, seen = Fold(), set()
    t_in, t_out, walk = 0.0, np.inf, True
    for a in range(3):
        if abs(dirn[a]) < 1e-20:
            walk &= lo[a] <= pos[a] <= hi[a]
        else:
            ta, tb = (lo[a] - pos[a]) / dirn[a], (hi[a] - pos[a]) / dirn[a]
            t_in, t_out = max(t_in, min(ta, tb)), min(t_out, max(ta, tb))
    walk &= t_in <= t_out
    if walk:
        c = [min(max(int((pos[a] + dirn[a] * t_in - lo[a]) / cell[a]), 0), dims[a] - 1) for a in range(3)]
        tm = [np.inf if abs(dirn[a]) < 1e-20 else (lo[a] + (c[a] + (0 if dirn[a] < 0 else 1)) * cell[a] - pos[a]) / dirn[a]
              for a in range(3)]
    while walk:
        word = plan["masks"][(c[2] * dims[1] + c[1]) * dims[0] + c[0]]
        for node in [n for n in range(128) if (int(word[n >> 6]) >> (n & 63)) & 1]:
            if node not in seen:
                seen.add(node)
                f.add(node, crossings(tab, node, pos, dirn))
        t_cell = min(tm)
        enough = f.nhits >= 2 and f.t2 + guard < t_cell and (not odd or (f.cnode >= 0 and f.cbest + guard < t_cell))
        ax = int(np.argmin(tm))
        nxt = c[ax] + (-1 if dirn[ax] < 0 else 1)
        if enough or not np.isfinite(t_cell) or nxt < 0 or nxt >= dims[ax]:
            break
        c[ax] = nxt
        tm[ax] += cell[ax] / abs(dirn[ax])
    f.add(root, crossings(tab, root, pos, dirn))
    if stats is not None:
        stats.append(len(seen))
    return f


def sample_rays(tab, plan, n, rng):
    """Origins all over the grid's box (and a shell around it), half of them moved ONTO a node's surface with the
    ray leaving along / across it; directions isotropic, some axis-parallel."""
    lo, hi = plan["lo"], plan["lo"] + plan["cell"] * np.array(plan["dims"])
    span = hi - lo
    rays = []
    for i in range(n):
        pos = lo - 0.2 * span + rng.random(3) * 1.4 * span
        d = rng.normal(size=3)
        if i % 7 == 0:
            d = np.zeros(3); d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
        d /= np.linalg.norm(d)
        if i % 2 == 0:      # advance to the first crossing of anything: the photon now sits on a surface
            f = brute_force(tab, pos, d)
            if f.nhits and np.isfinite(f.t1):
                pos = pos + d * f.t1
                if i % 4 == 0:
                    d = rng.normal(size=3); d /= np.linalg.norm(d)
        rays.append((pos, d))
    return rays


def tables(scene):
    compiled = compile_scene(scene)
    return compiled, {k: np.asarray(v) for k, v in compiled.tables().items()}


def assert_walk_equals_scan(scene, n, seed, expect_saving=None):
    compiled, tab = tables(scene)
    plan = native.node_grid_plan(compiled)
    assert plan is not None
    rng = np.random.default_rng(seed)
    visited = []
    for pos, d in sample_rays(tab, plan, n, rng):
        a, b = brute_force(tab, pos, d), grid_walk(tab, plan, pos, d, visited)
        assert a.outcome() == b.outcome(), (pos, d, a.outcome(), b.outcome())
        assert (a.t1 == b.t1) and (a.nhits < 2 or a.t2 == b.t2 or a.outcome()[1] == b.outcome()[1])
    if expect_saving is not None:
        assert np.mean(visited) < expect_saving * (len(tab["geom_type"]) - 1), np.mean(visited)
    return plan


def test_small_scenes_and_mesh_scenes_get_no_grid():
    for scene in (cfg2_lsc(), cfg4_nested_cylinders(), scenes.kitchen_sink(), scenes.mesh_gem(), tiles_lsc(2)):
        assert native.node_grid_plan(compile_scene(scene)) is None


@pytest.mark.parametrize("k", [3, 6, 11])
def test_every_tile_is_filed_under_the_cells_it_touches(k):
    compiled, tab = tables(tiles_lsc(k))
    plan = native.node_grid_plan(compiled)
    dims, lo, cell = plan["dims"], plan["lo"], plan["cell"]
    assert plan["dims"][2] == 1 and not plan["odd"] and plan["guard"] > 1e3 * 2.3e-13 * 250
    filed = np.zeros(128, dtype=int)
    for ci, word in enumerate(plan["masks"]):
        x, y = ci % dims[0], ci // dims[0]
        c_lo, c_hi = lo + cell * np.array([x, y, 0]), lo + cell * np.array([x + 1, y + 1, 1])
        for node in range(1, k * k + 1):
            centre = tab["local_to_world"][node][:3, 3]
            b_lo, b_hi = centre - 0.5 * tab["geom_params"][node][:3], centre + 0.5 * tab["geom_params"][node][:3]
            touches = bool(np.all(b_lo <= c_hi) and np.all(b_hi >= c_lo))
            bit = (int(word[node >> 6]) >> (node & 63)) & 1
            assert bit or not touches, (ci, node)          # a tile that touches the cell is filed there
            filed[node] += bit
        assert not int(word[0]) & 1                          # the root never is
    assert np.all(filed[1:k * k + 1] >= 1)


@pytest.mark.parametrize("k", [3, 6, 11])
def test_walk_equals_scan_on_tile_arrays(k):
    assert_walk_equals_scan(tiles_lsc(k), 1500, k, expect_saving=0.5 if k >= 6 else None)


def test_walk_equals_scan_on_tiles_that_share_faces():
    from tests.test_gpu_grid import touching_tiles
    assert_walk_equals_scan(touching_tiles(), 1500, 99)


@pytest.mark.parametrize("seed", range(12))
def test_walk_equals_scan_on_random_scenes(seed):
    scene = random_many_scene(seed)
    if native.node_grid_plan(compile_scene(scene)) is None:
        pytest.skip("scene is served by the plain node loop")
    assert_walk_equals_scan(scene, 600, seed)


def test_negative_control_boxes_filed_too_small_are_caught(monkeypatch):
    monkeypatch.setenv("PVT_GRID_DEV_SHRINK", "1")
    with pytest.raises(AssertionError):
        assert_walk_equals_scan(tiles_lsc(6), 1500, 6)
