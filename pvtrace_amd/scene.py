"""Scene graph: `Node` (a coordinate frame that may carry a geometry, a light
and recorders) and `Scene` (a root node).

API mirror of the reference's pvtrace/scene/node.py:15-201 and
pvtrace/scene/scene.py:92-151, without the third-party tree library: parent /
children bookkeeping, pre-order and level-order walks are implemented here.
Tracing itself is not done on these objects — `pvtrace_amd.engine` flattens the
graph into SoA tables and runs the HIP kernel.
"""
import numpy as np

from pvtrace_amd.common import AppError
from pvtrace_amd.geometry import Transformable
from pvtrace_amd.light import Event


class Intersection(object):
    """A crossing of a query ray with one node's surface (reference geometry/intersection.py:16-49):
    `point` in the frame of `coordsys`, `hit` the node that owns the surface, `distance` from the
    ray origin."""

    __slots__ = ("coordsys", "point", "hit", "distance")

    def __init__(self, coordsys, point, hit, distance):
        self.coordsys, self.hit, self.distance = coordsys, hit, float(distance)
        self.point = tuple(float(v) for v in point)

    def to(self, node):
        """The same crossing with its point expressed in `node`'s frame."""
        return Intersection(node, self.coordsys.point_to_node(self.point, node), self.hit, self.distance)

    def __eq__(self, other):
        return (isinstance(other, Intersection) and self.coordsys is other.coordsys and self.hit is other.hit
                and np.allclose(self.point, other.point) and abs(self.distance - other.distance) < 1e-9)

    def __repr__(self):
        return f"Intersection(hit={self.hit.name!r}, point={self.point}, distance={self.distance:.6g})"


class Node(Transformable):
    """A frame positioned relative to its parent node."""

    def __init__(
        self,
        name=None,
        parent=None,
        location=None,
        geometry=None,
        light=None,
        recorders=None,
    ):
        super(Node, self).__init__(location=location)
        self.name = name
        self._parent = None
        self._children = []
        self.parent = parent
        self.geometry = geometry
        self.light = light
        self.recorders = [] if recorders is None else list(recorders)

    def __repr__(self):
        return "Node({})".format(self.name)

    # -- tree bookkeeping ------------------------------------------------
    @property
    def parent(self):
        return self._parent

    @parent.setter
    def parent(self, node):
        if node is not None:
            if not isinstance(node, Node):
                raise AppError("Parent must be a Node.")
            probe = node
            while probe is not None:
                if probe is self:
                    raise AppError("A node cannot be its own ancestor.")
                probe = probe._parent
        if self._parent is not None:
            self._parent._children.remove(self)
        self._parent = node
        if node is not None:
            node._children.append(self)

    @property
    def children(self):
        return tuple(self._children)

    @property
    def is_root(self):
        return self._parent is None

    @property
    def is_leaf(self):
        return len(self._children) == 0

    @property
    def root(self):
        node = self
        while node._parent is not None:
            node = node._parent
        return node

    @property
    def path(self):
        """Nodes from the root down to (and including) this node."""
        chain = []
        node = self
        while node is not None:
            chain.append(node)
            node = node._parent
        return tuple(reversed(chain))

    @property
    def ancestors(self):
        return self.path[:-1]

    @property
    def leaves(self):
        return tuple(n for n in self.preorder() if n.is_leaf)

    def preorder(self):
        """Depth-first, parent before children, children in insertion order."""
        stack = [self]
        while stack:
            node = stack.pop()
            yield node
            stack.extend(reversed(node._children))

    def levelorder(self):
        queue = [self]
        while queue:
            node = queue.pop(0)
            yield node
            queue.extend(node._children)

    # -- orientation -----------------------------------------------------
    def look_at(self, vector):
        """Rotate so the node's +z axis points along `vector`."""
        a = np.array([0.0, 0.0, 1.0])
        b = np.asarray(vector, dtype=np.float64)
        b = b / np.linalg.norm(b)
        c = float(np.dot(a, b))
        if np.isclose(c, -1.0):
            self.rotate(np.pi, [0, 1, 0])
            return
        if np.isclose(c, 1.0):
            return
        axis = np.cross(a, b)
        angle = float(np.arccos(np.clip(c, -1.0, 1.0)))
        self.rotate(angle, axis)

    # -- frame conversion ------------------------------------------------
    def walk_to(self, node):
        """(upwards, common, downwards) node tuples between self and `node`."""
        mine, theirs = self.path, node.path
        if mine[0] is not theirs[0]:
            raise AppError("Nodes are not in the same tree.")
        k = 0
        while k < min(len(mine), len(theirs)) and mine[k] is theirs[k]:
            k += 1
        common = mine[k - 1]
        upwards = tuple(reversed(mine[k:]))
        downwards = tuple(theirs[k:])
        return upwards, common, downwards

    def path_to(self, node):
        upwards, common, downwards = self.walk_to(node)
        return upwards + (common,) + downwards

    def transformation_to(self, node):
        """4x4 matrix taking coordinates in this node's frame to `node`'s frame."""
        if self is node:
            return np.identity(4)
        upwards, _, downwards = self.walk_to(node)
        chain = [n.pose for n in upwards] + [np.linalg.inv(n.pose) for n in downwards]
        if len(chain) == 1:
            return chain[0]
        return np.linalg.multi_dot(chain[::-1])

    def point_to_node(self, point, node):
        m = self.transformation_to(node)
        p = np.ones(4)
        p[:3] = point
        return tuple((m @ p)[:3])

    def vector_to_node(self, vector, node):
        m = self.transformation_to(node)[:3, :3]
        return tuple(m @ np.asarray(tuple(vector), dtype=np.float64))

    # -- host-side ray queries (debugging / plumbing; the engine never calls these) -----------
    def intersections(self, ray_origin, ray_direction):
        """Crossings of a ray, given in THIS node's frame, with this node's geometry and every
        geometry below it, each in the frame of the node it hits (reference node.py:137-174).
        Geometry queries follow the kernel: only crossings further than EPS_ZERO count."""
        found = []
        for node in self.preorder():
            if node.geometry is None:
                continue
            origin = self.point_to_node(ray_origin, node) if node is not self else tuple(ray_origin)
            direction = self.vector_to_node(ray_direction, node) if node is not self else tuple(ray_direction)
            for point in node.geometry.intersections(origin, direction):
                found.append(Intersection(node, point, node, float(np.linalg.norm(np.subtract(point, origin)))))
        return tuple(found)

    # -- light -----------------------------------------------------------
    def emit(self, num_rays=None):
        if self.light is None:
            raise AppError("Not a lighting node.")
        for ray in self.light.emit(num_rays=num_rays):
            yield ray


def is_end_ray(event, metadata):
    """An "end ray": a ray generated, entering, leaving or reflected from a node, or ended (reference
    pvtrace/scene/scene.py:32-58) -- what `Scene.simulate(..., queue=q, end_rays=True)` sends to the queue."""
    if event in (Event.EMIT, Event.SCATTER, Event.ABSORB):
        return False
    if event in (Event.GENERATE, Event.NONRADIATIVE, Event.REACT, Event.KILL, Event.EXIT):
        return True
    if event in (Event.REFLECT, Event.TRANSMIT):
        if metadata["hit"] == metadata["adjacent"]:
            return True    # reflected from, or transmitted into, a node
        if metadata["hit"] == metadata["container"] and event == Event.TRANSMIT:
            return True    # escaped a node
    return False


def do_simulation(scene, num_rays, seed):
    """One worker's share (reference scene/scene.py:20-30): its numpy generator re-seeded, `num_rays` histories returned."""
    return scene.simulate(num_rays, workers=1, seed=seed)


def do_simulation_add_to_queue(scene, num_rays, seed, queue, end_rays):
    """The same with every event put on `queue` (reference :60-89); returns the pid."""
    return scene.simulate(num_rays, workers=1, seed=seed, queue=queue, end_rays=end_rays)


class Scene(object):
    """A scene graph of nodes rooted at `root`."""

    def __init__(self, root=None):
        super(Scene, self).__init__()
        self.root = root

    @property
    def light_nodes(self):
        """Nodes carrying a Light, in level order (the order rays cycle in)."""
        from pvtrace_amd.light import Light

        if self.root is None:
            return []
        return [n for n in self.root.levelorder() if isinstance(n.light, Light)]

    def finalise_nodes(self):
        """The reference refreshes the nodes' bounding boxes here before tracing (scene/scene.py:99-120; its renderer and its
        Python tracer's culling use them).  Nothing is cached on the nodes here -- the scene is flattened anew whenever it
        is traced -- so there is nothing to refresh; kept so that code written against the reference runs."""
        return None

    @property
    def component_nodes(self):
        found = []
        for node in self.root.levelorder():
            if node.geometry is not None and node.geometry.material is not None:
                found.extend(node.geometry.material.components)
        return found

    def emit(self, num_rays):
        """Rays in the root frame, cycling round-robin over the lights."""
        lights = self.light_nodes
        for idx in range(num_rays):
            node = lights[idx % len(lights)]
            for ray in node.emit(1):
                yield ray.representation(node, self.root)

    def intersections(self, ray_origin, ray_direction):
        """Forward crossings of a root-frame ray with every geometry of the scene, nearest first,
        points in the root frame (reference scene.py:153-195)."""
        if self.root is None:
            return tuple()
        crossings = [x.to(self.root) for x in self.root.intersections(ray_origin, ray_direction)]
        return tuple(sorted(crossings, key=lambda x: x.distance))

    def simulate(self, num_rays, workers=None, seed=None, queue=None, end_rays=False, maxsteps=1000, emit_method="kT"):
        """Emit `num_rays` from the lights and return the list of ray histories, each `[(Ray, Event), ...]` -- the
        reference's contract (pvtrace/scene/scene.py:197-313: `len(results) == num_rays`, equal results for equal
        `seed`, a `ValueError` for a seed with several workers).  The reference fans its Python tracer over a process
        pool; here the rays are traced by the engine on the GPU, 8 192 at a time with room for every event of a
        history, and the histories are rebuilt on the host -- for the engine's own result object (recorder tallies,
        packed logs, 10^6+ rays) call `engine.simulate(scene, num_rays, ...)` instead.

        `workers` only decides, as in the reference, whether a `seed` is allowed (it is for one worker: the seed
        re-seeds numpy's generator, which samples the lights and names the kernel's streams).  `queue`: every event goes
        to the queue as `(pid, ray index, Ray, Event, metadata)` instead of being returned (`end_rays`: only the events
        `is_end_ray` keeps, :32-58); the call then returns the pid, like the reference's worker."""
        import multiprocessing
        import os

        from pvtrace_amd import engine

        if workers is None:
            workers = max(1, multiprocessing.cpu_count() // 2)
        if workers != 1 and num_rays // workers > 0 and seed is not None:
            raise ValueError("Seed must be None to ensure different quasi-random sequences in each process")
        if seed is not None:
            np.random.seed(seed)
        results, pid, done = [], os.getpid(), 0
        while done < num_rays:
            n = min(8192, num_rays - done)
            result = engine.simulate(self, n, seed=None, maxsteps=maxsteps, max_events=2 * int(maxsteps) + 8,
                                     emit_method=emit_method, record_every=1, emission="host", packed_log=True)
            for k, history in enumerate(result.histories()):
                if queue is None:
                    results.append([(ray, event) for ray, event, _ in history])
                    continue
                for ray, event, metadata in history:
                    metadata = None if event == Event.GENERATE else metadata
                    if not end_rays or is_end_ray(event, metadata):
                        queue.put((pid, done + k, ray, event, metadata))
            done += n
        return pid if queue is not None else results
