"""Flatten a Scene into structure-of-arrays tables for the HIP tracer.

`CompiledScene` carries, field for field, what the reference's lowering
produces (pvtrace/engine/compiler.py:57-204: geometry / transform / material /
component / pooled-spectra / recorder / histogram tables, same dtypes, same
node order = pre-order over nodes that have a geometry, same error cases) so it
can be handed unchanged to the reference kernel — that is how the golden
fixtures are made — plus one extension the reference engine lacks: a per-node
table of declarative surface coatings (`coat_*`), which is what lets `LSC()`
(whose delegate the reference compiler rejects, compiler.py:237-247) and the
Coatings-notebook scene run on the device.

Everything here is host-side numpy executed once per `simulate` call; the
tables (a few KB) are packed and uploaded to HBM once by `native.DeviceScene`.
"""
import numpy as np

from pvtrace_amd.engine.recorder import (
    EVENTS,
    PROPERTIES,
    SOURCE_COMPONENT,
    SOURCE_COMPONENTS,
    SOURCE_LIGHTS,
    VOLUME_EVENTS,
    Heatmap,
    Recorder,
)
from pvtrace_amd.geometry import Box, Cylinder, Mesh, Sphere
from pvtrace_amd.material import (
    Absorber,
    CoatedSurfaceDelegate,
    Coating,
    Cone,
    FresnelSurfaceDelegate,
    HenyeyGreenstein,
    Luminophore,
    NullSurfaceDelegate,
    Reactor,
    Scatterer,
    isotropic,
)

MAX_NODES = 128       # device limit (reference _kernel.pyx:66, :929-930)
MAX_RECORDERS = 256   # per-photon distinct-ray bitmask width (compiler.py:23)

GEOM_BOX, GEOM_SPHERE, GEOM_CYLINDER, GEOM_MESH = 0, 1, 2, 3
SURF_FRESNEL, SURF_NULL = 0, 1
COMP_ABSORBER, COMP_SCATTERER, COMP_LUMINOPHORE, COMP_REACTOR = 0, 1, 2, 3
PHASE_ISOTROPIC, PHASE_HENYEY_GREENSTEIN, PHASE_CONE = 0, 1, 2
PHASE_LAMBERTIAN = 3   # extension: the reference compiler rejects it (compiler.py:300-310); cli/parse.py:166-167 builds it
EMIT_KT, EMIT_REDSHIFT, EMIT_FULL = 0, 1, 2
EMIT_METHODS = {"kT": EMIT_KT, "redshift": EMIT_REDSHIFT, "full": EMIT_FULL}

_F64 = np.float64
_I32 = np.int32


class UnsupportedSceneError(Exception):
    """The scene uses something the device engine cannot lower to tables."""


def _phase_of(node, component):
    """(tag, parameter) for a component's phase function."""
    import functools

    from pvtrace_amd import material as M

    phase = component.phase_function
    if phase is isotropic:
        return PHASE_ISOTROPIC, 0.0
    if isinstance(phase, HenyeyGreenstein):
        return PHASE_HENYEY_GREENSTEIN, float(phase.g)
    if isinstance(phase, Cone):
        return PHASE_CONE, float(phase.theta_max)
    if phase is M.lambertian:   # (reference material/utils.py:176-186; what `phase-function: {lambertian:}` parses to)
        return PHASE_LAMBERTIAN, 0.0
    # functools.partial spellings of the same built-ins (not recognised by the
    # reference compiler, which raises for them)
    if isinstance(phase, functools.partial) and not phase.keywords:
        if phase.func is M.cone and len(phase.args) == 1:
            return PHASE_CONE, float(phase.args[0])
        if phase.func is M.henyey_greenstein and len(phase.args) == 1:
            return PHASE_HENYEY_GREENSTEIN, float(phase.args[0])
    raise UnsupportedSceneError(
        f"Node {node.name!r}: custom phase functions are not supported."
    )


class CompiledScene:
    """SoA tables describing `scene` (see module docstring)."""

    def __init__(self, scene):
        root = scene.root
        if root is None:
            raise UnsupportedSceneError("Scene has no geometry nodes.")
        nodes = [n for n in root.preorder() if n.geometry is not None]
        if not nodes:
            raise UnsupportedSceneError("Scene has no geometry nodes.")
        if root.geometry is None:
            raise UnsupportedSceneError("Root node must have a geometry.")

        count = len(nodes)
        self.scene = scene
        self.nodes = nodes
        self.node_names = [n.name for n in nodes]
        self.root_id = nodes.index(root)

        self.geom_type = np.zeros(count, dtype=_I32)
        self.geom_params = np.zeros((count, 4), dtype=_F64)
        self.local_to_world = np.zeros((count, 4, 4), dtype=_F64)
        self.world_to_local = np.zeros((count, 4, 4), dtype=_F64)
        self.refractive_index = np.zeros(count, dtype=_F64)
        self.surface_type = np.zeros(count, dtype=_I32)
        self.comp_start = np.zeros(count, dtype=_I32)
        self.comp_count = np.zeros(count, dtype=_I32)
        self.coat_start = np.zeros(count, dtype=_I32)
        self.coat_count = np.zeros(count, dtype=_I32)
        self.mesh_face_start = np.zeros(count, dtype=_I32)
        self.mesh_face_count = np.zeros(count, dtype=_I32)
        self._mesh_pool = {"vertices": [], "faces": [], "normals": [], "nv": 0, "nf": 0}

        pools = {"abs_x": [], "abs_y": [], "ems_x": [], "ems_cdf": []}
        comp_cols = {
            key: []
            for key in (
                "type", "qy", "tau_rad", "tau_nr", "phase_type", "phase_param",
                "abs_start", "abs_n", "ems_start", "ems_n", "abs_hist", "ems_hist",
            )
        }
        coat_rows = []
        self.component_names = []

        for i, node in enumerate(nodes):
            geometry = node.geometry
            self._lower_geometry(i, geometry)
            self._lower_transform(i, node, root)

            material = geometry.material
            if material is None:
                raise UnsupportedSceneError(
                    f"Node {node.name!r} has geometry without a material."
                )
            self.refractive_index[i] = float(material.refractive_index)

            self.coat_start[i] = len(coat_rows)
            self.surface_type[i] = self._lower_surface(node, material, coat_rows)
            self.coat_count[i] = len(coat_rows) - self.coat_start[i]

            self.comp_start[i] = len(comp_cols["type"])
            for component in material.components:
                self._lower_component(node, component, comp_cols, pools)
                self.component_names.append(component.name)
            self.comp_count[i] = len(material.components)

        self.comp_type = np.array(comp_cols["type"], dtype=_I32)
        self.comp_qy = np.array(comp_cols["qy"], dtype=_F64)
        self.comp_tau_rad = np.array(comp_cols["tau_rad"], dtype=_F64)
        self.comp_tau_nr = np.array(comp_cols["tau_nr"], dtype=_F64)
        self.comp_phase_type = np.array(comp_cols["phase_type"], dtype=_I32)
        self.comp_phase_param = np.array(comp_cols["phase_param"], dtype=_F64)
        self.comp_abs_start = np.array(comp_cols["abs_start"], dtype=_I32)
        self.comp_abs_n = np.array(comp_cols["abs_n"], dtype=_I32)
        self.comp_ems_start = np.array(comp_cols["ems_start"], dtype=_I32)
        self.comp_ems_n = np.array(comp_cols["ems_n"], dtype=_I32)
        self.comp_abs_hist = np.array(comp_cols["abs_hist"], dtype=_I32)
        self.comp_ems_hist = np.array(comp_cols["ems_hist"], dtype=_I32)

        self.abs_x = np.array(pools["abs_x"], dtype=_F64)
        self.abs_y = np.array(pools["abs_y"], dtype=_F64)
        self.ems_x = np.array(pools["ems_x"], dtype=_F64)
        self.ems_cdf = np.array(pools["ems_cdf"], dtype=_F64)

        # Coating table: one row per Coating, grouped per node.
        ncoat = len(coat_rows)
        self.coat_facet = np.zeros((max(ncoat, 1), 3), dtype=_F64)
        self.coat_lo = np.full((max(ncoat, 1), 3), -np.inf, dtype=_F64)
        self.coat_hi = np.full((max(ncoat, 1), 3), np.inf, dtype=_F64)
        self.coat_reflectivity = np.full(ncoat, -1.0, dtype=_F64)
        self.coat_reflect_mode = np.zeros(ncoat, dtype=_I32)
        self.coat_transmit_mode = np.zeros(ncoat, dtype=_I32)
        for r, coating in enumerate(coat_rows):
            self.coat_facet[r] = coating.facet
            self.coat_lo[r] = [b[0] for b in coating.region]
            self.coat_hi[r] = [b[1] for b in coating.region]
            if coating.reflectivity is not None:
                self.coat_reflectivity[r] = coating.reflectivity
            self.coat_reflect_mode[r] = Coating.REFLECTION_MODES[coating.reflection]
            self.coat_transmit_mode[r] = Coating.TRANSMISSION_MODES[coating.transmission]
        self.n_coatings = ncoat

        pool = self._mesh_pool
        self.n_mesh_vertices, self.n_mesh_faces = pool["nv"], pool["nf"]
        self.mesh_vertices = (np.concatenate(pool["vertices"]) if pool["nv"] else np.zeros((0, 3), dtype=_F64))
        self.mesh_faces = (np.concatenate(pool["faces"]).astype(_I32) if pool["nf"] else np.zeros((0, 3), dtype=_I32))
        self.mesh_normals = (np.concatenate(pool["normals"]) if pool["nf"] else np.zeros((0, 3), dtype=_F64))
        del self._mesh_pool

        self._lower_recorders(nodes)

    # -- geometry & pose -------------------------------------------------
    def _lower_geometry(self, i, geometry):
        if isinstance(geometry, Box):
            self.geom_type[i] = GEOM_BOX
            self.geom_params[i, :3] = np.asarray(geometry._size, dtype=_F64)
        elif isinstance(geometry, Sphere):
            self.geom_type[i] = GEOM_SPHERE
            self.geom_params[i, 0] = float(geometry.radius)
        elif isinstance(geometry, Cylinder):
            self.geom_type[i] = GEOM_CYLINDER
            self.geom_params[i, 0] = float(geometry.length)
            self.geom_params[i, 1] = float(geometry.radius)
        elif isinstance(geometry, Mesh):
            # EXTENSION: the reference engine rejects meshes (compiler.py:220-223)
            pool = self._mesh_pool
            self.geom_type[i] = GEOM_MESH
            self.mesh_face_start[i] = pool["nf"]
            self.mesh_face_count[i] = len(geometry.faces)
            pool["vertices"].append(np.asarray(geometry.vertices, dtype=_F64))
            pool["faces"].append(np.asarray(geometry.faces, dtype=_I32) + pool["nv"])
            pool["normals"].append(np.asarray(geometry.face_normals, dtype=_F64))
            pool["nv"] += len(geometry.vertices)
            pool["nf"] += len(geometry.faces)
            lo, hi = geometry.vertices.min(axis=0), geometry.vertices.max(axis=0)
            self.geom_params[i, :3] = hi - lo          # informational: local AABB size
        else:
            raise UnsupportedSceneError(
                f"Geometry type {type(geometry).__name__} is not supported."
            )

    def _lower_transform(self, i, node, root):
        l2w = np.asarray(node.transformation_to(root), dtype=_F64)
        rot = l2w[:3, :3]
        if not np.allclose(rot @ rot.T, np.eye(3), atol=1e-9):
            raise UnsupportedSceneError(
                f"Node {node.name!r} transform is not rigid (has scale or shear)."
            )
        self.local_to_world[i] = l2w
        self.world_to_local[i] = np.linalg.inv(l2w)

    # -- surfaces --------------------------------------------------------
    def _lower_surface(self, node, material, coat_rows):
        delegate = material.surface.delegate
        if type(delegate) is FresnelSurfaceDelegate:
            return SURF_FRESNEL
        if type(delegate) is NullSurfaceDelegate:
            return SURF_NULL
        if isinstance(delegate, CoatedSurfaceDelegate):
            # `coatings` may be computed lazily from mutable state (the LSC
            # builder's delegates do that), so read it at flatten time.
            for coating in delegate.coatings:
                if not isinstance(coating, Coating):
                    raise UnsupportedSceneError(
                        f"Node {node.name!r}: coatings must be Coating objects."
                    )
                coat_rows.append(coating)
            return SURF_FRESNEL
        raise UnsupportedSceneError(
            f"Node {node.name!r} uses surface delegate "
            f"{type(delegate).__name__}; only FresnelSurfaceDelegate, "
            "NullSurfaceDelegate and CoatedSurfaceDelegate (declarative "
            "coatings) are supported."
        )

    # -- components ------------------------------------------------------
    def _lower_component(self, node, component, cols, pools):
        # Subclass order matters: Reactor < Absorber < Scatterer > Luminophore
        if isinstance(component, Reactor):
            ctype = COMP_REACTOR
        elif isinstance(component, Absorber):
            ctype = COMP_ABSORBER
        elif isinstance(component, Luminophore):
            ctype = COMP_LUMINOPHORE
        elif isinstance(component, Scatterer):
            ctype = COMP_SCATTERER
        else:
            raise UnsupportedSceneError(
                f"Component type {type(component).__name__} is not supported."
            )
        phase_type, phase_param = _phase_of(node, component)
        a_start, a_n = self._pool_spectrum(
            node, component._abs_dist, pools["abs_x"], pools["abs_y"]
        )
        e_start, e_n, e_hist = 0, 0, 0
        if ctype == COMP_LUMINOPHORE:
            dist = component._ems_dist
            e_hist = 1 if dist.hist else 0   # histogram-sampled emission (extension, see header)
            e_start = len(pools["ems_x"])
            pools["ems_x"].extend(np.asarray(dist._x, dtype=_F64).tolist())
            pools["ems_cdf"].extend(np.asarray(dist._cdf, dtype=_F64).tolist())
            e_n = len(pools["ems_x"]) - e_start

        cols["type"].append(ctype)
        cols["qy"].append(float(component.quantum_yield))
        cols["tau_rad"].append(float(component.tau_rad) if component.tau_rad else 0.0)
        cols["tau_nr"].append(float(component.tau_nr) if component.tau_nr else 0.0)
        cols["phase_type"].append(phase_type)
        cols["phase_param"].append(phase_param)
        cols["abs_start"].append(a_start)
        cols["abs_n"].append(a_n)
        cols["ems_start"].append(e_start)
        cols["ems_n"].append(e_n)
        cols["abs_hist"].append(1 if (component._abs_dist.hist and component._abs_dist._x is not None) else 0)
        cols["ems_hist"].append(e_hist)

    def _pool_spectrum(self, node, dist, xs, ys):
        # hist=True spectra are pooled like interpolated ones; the per-component hist flag
        # switches the device lookup to the step-function rule (extension: the reference
        # compiler raises here, compiler.py:313-317)
        start = len(xs)
        if dist._x is None:
            # constant coefficient -> a one-point table
            xs.append(0.0)
            ys.append(float(dist._y))
            return start, 1
        xs.extend(np.asarray(dist._x, dtype=_F64).tolist())
        ys.extend(np.asarray(dist._y, dtype=_F64).tolist())
        return start, len(xs) - start

    # -- recorders -------------------------------------------------------
    def _lower_recorders(self, nodes):
        found = []
        for i, node in enumerate(nodes):
            for recorder in getattr(node, "recorders", []):
                if not isinstance(recorder, Recorder):
                    raise UnsupportedSceneError(
                        f"Node {node.name!r} recorders must be Recorder objects."
                    )
                if recorder.event in VOLUME_EVENTS and recorder.facet is not None:
                    raise UnsupportedSceneError(
                        f"Recorder {recorder.name!r}: facet filters only apply "
                        "to surface events."
                    )
                found.append((i, recorder))
        if len(found) > MAX_RECORDERS:
            raise UnsupportedSceneError(
                f"At most {MAX_RECORDERS} recorders are supported."
            )
        names = [rec.name for _, rec in found]
        if len(set(names)) != len(names):
            raise UnsupportedSceneError("Recorder names must be unique.")

        n = len(found)
        self.recorder_names = names
        self.recorder_specs = [rec for _, rec in found]
        self.rec_node = np.zeros(n, dtype=_I32)
        self.rec_event = np.zeros(n, dtype=_I32)
        self.rec_has_facet = np.zeros(n, dtype=_I32)
        self.rec_facet = np.zeros((max(n, 1), 3), dtype=_F64)
        self.rec_atol = np.zeros(n, dtype=_F64)
        self.rec_hist_start = np.zeros(n, dtype=_I32)
        self.rec_hist_n = np.zeros(n, dtype=_I32)
        self.rec_source_mode = np.zeros(n, dtype=_I32)
        self.rec_source_id = np.full(n, -1, dtype=_I32)

        hist = {k: [] for k in ("pa", "pb", "na", "nb", "loa", "hia", "lob", "hib", "off")}
        offset = 0
        for r, (node_index, recorder) in enumerate(found):
            self.rec_node[r] = node_index
            self.rec_event[r] = EVENTS[recorder.event]
            if recorder.facet is not None:
                self.rec_has_facet[r] = 1
                self.rec_facet[r] = recorder.facet
            self.rec_atol[r] = recorder.atol
            src = getattr(recorder, "source", None)
            if src is not None:
                if src == "lights":
                    self.rec_source_mode[r] = SOURCE_LIGHTS
                elif src == "components":
                    self.rec_source_mode[r] = SOURCE_COMPONENTS
                elif src in self.component_names:
                    if self.component_names.count(src) > 1:
                        # the filter is one component id; the reference's dataframe filter by name
                        # would match every component of that name
                        raise UnsupportedSceneError(
                            f"Recorder {recorder.name!r}: source {src!r} names "
                            f"{self.component_names.count(src)} components; give them distinct names."
                        )
                    self.rec_source_mode[r] = SOURCE_COMPONENT
                    self.rec_source_id[r] = self.component_names.index(src)
                else:
                    raise UnsupportedSceneError(
                        f"Recorder {recorder.name!r}: unknown source {src!r} (use 'lights', "
                        "'components' or a component name)."
                    )
            self.rec_hist_start[r] = len(hist["pa"])
            for spec in recorder.histograms:
                if isinstance(spec, Heatmap):
                    a, b = spec.a, spec.b
                    row = (PROPERTIES[a.prop], PROPERTIES[b.prop], a.bins, b.bins,
                           a.start, a.stop, b.start, b.stop)
                else:
                    row = (PROPERTIES[spec.prop], -1, spec.bins, 1,
                           spec.start, spec.stop, 0.0, 1.0)
                for key, value in zip(("pa", "pb", "na", "nb", "loa", "hia", "lob", "hib"), row):
                    hist[key].append(value)
                hist["off"].append(offset)
                offset += row[2] * row[3]
            self.rec_hist_n[r] = len(recorder.histograms)

        self.hist_prop_a = np.array(hist["pa"], dtype=_I32)
        self.hist_prop_b = np.array(hist["pb"], dtype=_I32)
        self.hist_na = np.array(hist["na"], dtype=_I32)
        self.hist_nb = np.array(hist["nb"], dtype=_I32)
        self.hist_lo_a = np.array(hist["loa"], dtype=_F64)
        self.hist_hi_a = np.array(hist["hia"], dtype=_F64)
        self.hist_lo_b = np.array(hist["lob"], dtype=_F64)
        self.hist_hi_b = np.array(hist["hib"], dtype=_F64)
        self.hist_offset = np.array(hist["off"], dtype=_I32)
        self.total_bins = int(offset)

    # -- introspection ----------------------------------------------------
    TABLE_FIELDS = (
        "geom_type", "geom_params", "local_to_world", "world_to_local",
        "refractive_index", "surface_type", "comp_start", "comp_count",
        "comp_type", "comp_qy", "comp_tau_rad", "comp_tau_nr",
        "comp_phase_type", "comp_phase_param", "comp_abs_start", "comp_abs_n",
        "comp_ems_start", "comp_ems_n", "comp_abs_hist", "comp_ems_hist", "abs_x", "abs_y", "ems_x", "ems_cdf",
        "rec_node", "rec_event", "rec_has_facet", "rec_facet", "rec_atol",
        "rec_hist_start", "rec_hist_n", "rec_source_mode", "rec_source_id", "hist_prop_a", "hist_prop_b", "hist_na",
        "hist_nb", "hist_lo_a", "hist_hi_a", "hist_lo_b", "hist_hi_b",
        "hist_offset",
        "coat_start", "coat_count", "coat_facet", "coat_lo", "coat_hi",
        "coat_reflectivity", "coat_reflect_mode", "coat_transmit_mode",
        "mesh_face_start", "mesh_face_count", "mesh_vertices", "mesh_faces", "mesh_normals",
    )

    def tables(self):
        """dict of every numeric table (for fixtures / debugging)."""
        out = {name: getattr(self, name) for name in self.TABLE_FIELDS}
        out["root_id"] = np.int32(self.root_id)
        out["total_bins"] = np.int32(self.total_bins)
        return out

    @property
    def has_coatings(self):
        return self.n_coatings > 0


def compile_scene(scene) -> CompiledScene:
    """Flatten `scene` into tables, or raise `UnsupportedSceneError`."""
    return CompiledScene(scene)
