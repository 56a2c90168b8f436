"""High-level luminescent-solar-concentrator builder.

Builds the same scene as the reference's pvtrace/device/lsc.py:89-219 — world
box 100x the slab, slab with Lumogen F Red 305 (peak 10 cm^-1) + 0.1 cm^-1
background absorber, 20-degree cone spotlight of 555 nm at z = 5*depth flipped
to point down — with the same `add_*` configuration methods.

The reference attaches Python-callback delegates to the slab
(`OptionalMirrorAndSolarCell`, lsc.py:22-62; `AirGapMirror`, :65-86), which its
own compiled engine refuses (compiler.py:237-247), so `LSC()` only ever runs on
the slow Python tracer there.  Here the two delegates keep their names but are
*declarative*: they expose `coatings` computed from the LSC's configuration and
the flattener lowers them into the device coating table, so every LSC variant
traces on the GPU.  With no cells and no mirror the slab is plain Fresnel,
exactly as in the reference (lsc.py:46-48, :60-62).
"""
import functools

import numpy as np

from pvtrace_amd.data import lumogen_f_red_305
from pvtrace_amd.engine.recorder import Histogram, Recorder
from pvtrace_amd.geometry import Box
from pvtrace_amd.light import Light
from pvtrace_amd.material import (
    Absorber, CoatedSurfaceDelegate, Coating, Luminophore, Material, Scatterer, Surface, cone,
)
from pvtrace_amd.scene import Node, Scene

# facet label -> outward local normal of the slab (reference lsc.py:33-41)
FACETS = {
    "left": (-1.0, 0.0, 0.0), "right": (1.0, 0.0, 0.0),
    "near": (0.0, -1.0, 0.0), "far": (0.0, 1.0, 0.0),
    "bottom": (0.0, 0.0, -1.0), "top": (0.0, 0.0, 1.0),
}


class OptionalMirrorAndSolarCell(CoatedSurfaceDelegate):
    """Ideal specular mirror on the bottom face (if requested) and perfectly
    index-matched, perfectly absorbing solar cells on chosen edge faces."""

    def __init__(self, lsc):
        super(OptionalMirrorAndSolarCell, self).__init__()
        self.lsc = lsc

    @property
    def coatings(self):
        rows = []
        if self.lsc._back_surface_mirror_info["want_back_surface_mirror"]:
            rows.append(Coating(FACETS["bottom"], reflectivity=1.0))
        for label in ("left", "right", "near", "far"):
            if label in self.lsc._solar_cell_surfaces:
                rows.append(Coating(FACETS[label], reflectivity=0.0, transmission="matched"))
        return rows


class AirGapMirror(CoatedSurfaceDelegate):
    """Perfect reflector sheet under the slab; specular or Lambertian."""

    def __init__(self, lsc):
        super(AirGapMirror, self).__init__()
        self.lsc = lsc

    @property
    def coatings(self):
        mode = "lambertian" if self.lsc._air_gap_mirror_info["lambertian"] else "specular"
        return [Coating(normal, reflectivity=1.0, reflection=mode) for normal in FACETS.values()]


class Spectrum(np.ndarray):
    """Wavelengths in nm, one per ray, reconstructed from a histogram: every ray at the centre of its bin.  `edges` (n + 1)
    and `counts` (n) are the histogram it came from, exact."""

    def __new__(cls, edges, counts):
        edges, counts = np.asarray(edges, dtype=np.float64), np.asarray(counts)
        obj = np.repeat(0.5 * (edges[:-1] + edges[1:]), counts.astype(np.int64)).view(cls)
        obj.edges, obj.counts = edges, counts
        return obj

    def __array_finalize__(self, obj):
        self.edges = getattr(obj, "edges", None)
        self.counts = getattr(obj, "counts", None)


class LSC(object):
    """Abstraction of a luminescent solar concentrator (high-level API)."""

    def __init__(self, size, wavelength_range=None, n0=1.0, n1=1.5):
        super(LSC, self).__init__()
        self.wavelength_range = (
            np.arange(400, 800) if wavelength_range is None else np.asarray(wavelength_range)
        )
        self.size = size  # centimetres
        self.n0 = n0
        self.n1 = n1
        self._solar_cell_surfaces = set()
        self._back_surface_mirror_info = {"want_back_surface_mirror": False}
        self._air_gap_mirror_info = {"want_air_gap_mirror": False, "lambertian": False}
        self._scene = None
        self._result = None
        self._user_lights = []
        self._user_components = []

    # -- configuration ----------------------------------------------------
    def add_luminophore(self, name, coefficient, emission, quantum_yield, phase_function=None):
        self._user_components.append({
            "cls": Luminophore, "name": name, "coefficient": coefficient, "emission": emission,
            "quantum_yield": quantum_yield, "phase_function": phase_function,
        })

    def add_absorber(self, name, coefficient):
        self._user_components.append({"cls": Absorber, "name": name, "coefficient": coefficient})

    def add_scatterer(self, name, coefficient, phase_function=None):
        self._user_components.append({
            "cls": Scatterer, "name": name, "coefficient": coefficient,
            "phase_function": phase_function,
        })

    def add_light(self, name, location, rotation=None, direction=None, wavelength=None,
                  position=None):
        self._user_lights.append({
            "name": name, "location": location, "rotation": rotation, "direction": direction,
            "wavelength": wavelength, "position": position,
        })

    def add_solar_cell(self, facets):
        if not isinstance(facets, (list, tuple, set)):
            raise ValueError("Facets should be a set. e.g. `{'left', 'right'}`")
        facets = set(facets)
        allowed = {"left", "near", "far", "right"}
        if not facets.issubset(allowed):
            raise ValueError("Solar cell have allowed surfaces", allowed)
        self._solar_cell_surfaces = facets.union(self._solar_cell_surfaces)

    def add_back_surface_mirror(self):
        self._back_surface_mirror_info = {"want_back_surface_mirror": True}

    def add_air_gap_mirror(self, lambertian=False):
        self._air_gap_mirror_info = {"want_air_gap_mirror": True, "lambertian": lambertian}

    # -- defaults -----------------------------------------------------------
    def _make_default_components(self):
        x = self.wavelength_range
        return [
            {
                "cls": Luminophore, "name": "Lumogen F Red 305",
                "coefficient": np.column_stack((x, lumogen_f_red_305.absorption(x) * 10.0)),
                "emission": np.column_stack((x, lumogen_f_red_305.emission(x))),
                "quantum_yield": 1.0, "phase_function": None,
            },
            {"cls": Absorber, "coefficient": 0.1, "name": "Background"},
        ]

    def _make_default_lights(self):
        return [{
            "name": "Light", "location": (0.0, 0.0, self.size[-1] * 5),
            "rotation": (np.radians(180), (1, 0, 0)),
            "direction": functools.partial(cone, np.radians(20)),
            "wavelength": None, "position": None,
        }]

    # -- scene ----------------------------------------------------------------
    def _make_scene(self):
        (l, w, d) = self.size
        world = Node(
            name="World",
            geometry=Box((l * 100, w * 100, d * 100), material=Material(refractive_index=self.n0)),
        )
        if len(self._user_components) == 0:
            self._user_components = self._make_default_components()
        components = []
        for spec in self._user_components:
            spec = dict(spec)
            cls = spec.pop("cls")
            coefficient = spec.pop("coefficient")
            components.append(cls(coefficient, **spec))

        Node(
            name="LSC",
            geometry=Box(
                (l, w, d),
                material=Material(
                    refractive_index=self.n1,
                    components=components,
                    surface=Surface(delegate=OptionalMirrorAndSolarCell(self)),
                ),
            ),
            parent=world,
            recorders=self._default_recorders(),
        )

        if self._air_gap_mirror_info["want_air_gap_mirror"]:
            sheet = 0.25 * d
            mirror = Node(
                name="Air Gap Mirror",
                geometry=Box(
                    (l, w, sheet),
                    material=Material(
                        refractive_index=self.n0, components=[],
                        surface=Surface(delegate=AirGapMirror(self)),
                    ),
                ),
                parent=world,
            )
            mirror.translate((0.0, 0.0, -(0.5 * d + sheet)))

        if len(self._user_lights) == 0:
            self._user_lights = self._make_default_lights()
        for spec in self._user_lights:
            light = Light(name=spec["name"], direction=spec["direction"],
                          wavelength=spec["wavelength"], position=spec["position"])
            node = Node(name=spec["name"], light=light, parent=world)
            node.location = spec["location"]
            if spec["rotation"]:
                node.rotate(*spec["rotation"])
        self._scene = Scene(world)

    def _default_recorders(self):
        """Recorders the report is built from.  The reference derives its "Solar In/Out,
        Luminescent In/Out" table from a per-ray pandas frame filtered by facet and by the
        ray's source (lsc.py:440-489); here the same split is tallied on the GPU by facet
        recorders with a source filter (light-emitted vs component-emitted photons)."""
        lo, hi = float(self.wavelength_range.min()), float(self.wavelength_range.max()) + 1.0
        nbins = max(int(round((hi - lo) / 5.0)), 1)
        hist = lambda: [Histogram("wavelength", lo, hi, nbins)]   # noqa: E731 -- what spectrum() reads
        recs = []
        for label, normal in FACETS.items():
            recs += [
                Recorder(f"solar-in-{label}", event="entering", facet=normal, source="lights", histograms=hist()),
                Recorder(f"solar-out-{label}", event="escaping", facet=normal, source="lights", histograms=hist()),
                Recorder(f"solar-reflected-{label}", event="reflected", facet=normal, source="lights", histograms=hist()),
                Recorder(f"lum-in-{label}", event="entering", facet=normal, source="components"),
                Recorder(f"lum-out-{label}", event="escaping", facet=normal, source="components", histograms=hist()),
            ]
        recs += [Recorder("lost", event="lost"), Recorder("killed", event="killed", histograms=hist()),
                 Recorder("lost-solar", event="lost", source="lights", histograms=hist()),
                 Recorder("lost-lum", event="lost", source="components", histograms=hist())]
        # Several emitting components: the luminescent tallies once more per component, so that spectrum(source=...)
        # can tell them apart (with one, "components" IS that component)
        emitters = self._emitting_components()
        if len(emitters) > 1:
            for k, name in enumerate(emitters):
                recs.append(Recorder(f"lost-lum-{k}", event="lost", source=name, histograms=hist()))
                for label, normal in FACETS.items():
                    recs.append(Recorder(f"lum-out-{label}-{k}", event="escaping", facet=normal, source=name, histograms=hist()))
        return recs

    def _emitting_components(self):
        """Names of the components that can be the source of a ray (luminophores and scatterers)."""
        return [c["name"] for c in self._user_components if c["cls"] in (Luminophore, Scatterer)]

    @property
    def scene(self):
        if self._scene is None:
            self._make_scene()
        return self._scene

    def component_names(self):
        if self._scene is None:
            raise ValueError("Run a simulation before calling this method.")
        return {c["name"] for c in self._user_components}

    def light_names(self):
        if self._scene is None:
            raise ValueError("Run a simulation before calling this method.")
        return {l["name"] for l in self._user_lights}

    # -- simulate ---------------------------------------------------------------
    def simulate(self, n, progress=None, emit_method="kT", seed=None, **kwargs):
        """Trace `n` photons on the GPU engine; tallies replace the reference's
        per-ray dataframe.  Returns the `EngineResult`."""
        from pvtrace_amd import engine

        kwargs.setdefault("record_every", 0)
        self._result = engine.simulate(self.scene, n, seed=seed, emit_method=emit_method, **kwargs)
        if progress:
            progress(n)
        return self._result

    def counts(self):
        """Surface counts per facet, columns as in the reference's report (lsc.py:440-489):
        Solar In / Solar Out / Luminescent Out / Luminescent In (distinct photons).
        "Solar Out" adds the photons reflected off a facet to those transmitted out of it,
        like the reference's exit-ray bookkeeping.  Returns a pandas DataFrame when pandas
        is available, else a dict of dicts."""
        if self._result is None:
            raise ValueError("Run a simulation before calling this method.")
        r = {name: rec.rays for name, rec in self._result.recorders.items()}
        table = {
            "Solar In": {f: r[f"solar-in-{f}"] for f in FACETS},
            "Solar Out": {f: r[f"solar-out-{f}"] + r[f"solar-reflected-{f}"] for f in FACETS},
            "Luminescent Out": {f: r[f"lum-out-{f}"] for f in FACETS},
            "Luminescent In": {f: r[f"lum-in-{f}"] for f in FACETS},
        }
        try:
            import pandas as pd

            return pd.DataFrame(table, index=list(FACETS))
        except ImportError:
            return table

    def summary(self):
        """Headline figures of the reference's LSC.summary() (lsc.py:583-620)."""
        counts = self.counts()
        get = (lambda col, f: int(counts[col][f]))
        cells = set(self._solar_cell_surfaces)
        lum_collected = sum(get("Luminescent Out", f) for f in cells)
        lum_escaped = sum(get("Luminescent Out", f) for f in FACETS if f not in cells)
        incident = sum(get("Solar In", f) for f in FACETS)
        lost = self._result.recorders["lost"].rays
        (l, w, d) = self.size
        cg = (w * l) / (2 * l * d + 2 * w * d)
        n = self.n1
        nan = float("nan")
        out = {
            "Optical Efficiency": lum_collected / incident if incident else nan,
            "Waveguide Efficiency": (lum_collected / (lum_collected + lum_escaped)
                                     if lum_collected + lum_escaped else nan),
            "Waveguide Efficiency (Thermodynamic Prediction)": n ** 2 / (cg + n ** 2),
            "Non-radiative Loss (fraction):": lost / incident if incident else nan,
            "Incident": incident,
            "Geometric Concentration": cg,
            "Refractive Index": n,
            "Cell Surfaces": cells,
            "Components": self.component_names(),
            "Lights": self.light_names(),
        }
        try:
            import pandas as pd

            return pd.Series(out)
        except ImportError:
            return out

    def spectrum(self, facets=set(), kind="last", source="all", events=None):
        """Wavelength spectrum of the rays the reference's `LSC.spectrum` would select (lsc.py:505-566): same arguments,
        same validation, same errors, and like the reference's a sequence of wavelengths with ONE ENTRY PER SELECTED RAY
        (`len(lsc.spectrum(...))` counts rays, `plt.hist(lsc.spectrum(...), bins=...)` plots them) -- answered from
        recorder tallies, so a ray's wavelength is known to its 5 nm bin and stands at the bin's centre; the histogram
        itself is exact and rides along as `.edges` / `.counts` (`Spectrum`).

        The reference keeps two rows per photon: its `kind="first"` row (the first event after generation: the photon
        transmitted into, or reflected off, a facet) and its `kind="last"` row (the photon when it was lost, or at its
        last interaction before it left the world: transmitted out of, or reflected off, a facet); `kind=None` selects
        both.  `source`: "all", a name, or a set of names out of `component_names() | light_names()` -- who emitted the
        ray in that row; `facets`: a set of facet labels (empty: no filter -- rows without a facet, i.e. photons lost
        inside the slab, then count too); `events`: a set of event names ("transmit", "reflect", "absorb", "kill", ...).
        Tallies count a photon once per facet recorder (its first matching crossing), the approximation `counts()`
        makes too; several LIGHTS are tallied together (the engine knows a light-emitted ray only as such), so a
        source set that names some but not all lights is refused."""
        if self._result is None:
            raise ValueError("Run a simulation before calling this method.")
        if kind is not None:
            if kind not in {"first", "last"}:
                raise ValueError("Direction must be either `'first'` or `'last'.`")
        all_sources = self.component_names() | self.light_names()
        if source == "all":
            want_sources = set(all_sources)
        else:
            if isinstance(source, str):
                source = {source}
            if not set(source).issubset(all_sources):
                raise ValueError("Unknown source requested.", set(source).difference(all_sources))
            want_sources = set(source)
        if isinstance(facets, (list, tuple, set)):
            want_facets = set(facets)
        else:
            raise ValueError("`'facets'` should be a set `{'left', 'right'}`", {"got": facets})
        if events is not None:
            from pvtrace_amd.light import Event

            all_events = {e.name.lower() for e in Event}
            if isinstance(events, (list, tuple, set)):
                events = set(events)
                if not events.issubset(all_events):
                    raise ValueError("Contained some unknown events", {"got": events, "expected": all_events})
            else:
                raise ValueError("Events must be set of event strings", {"allowed": all_events})
        lights = self.light_names()
        picked = want_sources & lights
        if picked and picked != lights:
            raise ValueError("the lights of an LSC are tallied together: name all of them or none", {"got": picked})
        emitters = self._emitting_components()
        # (recorder, row kind, event of the row, facet or None, sources it stands for)
        rows = []
        for f in FACETS:
            rows += [(f"solar-in-{f}", "first", "transmit", f, lights), (f"solar-reflected-{f}", "first", "reflect", f, lights),
                     (f"solar-out-{f}", "last", "transmit", f, lights), (f"solar-reflected-{f}", "last", "reflect", f, lights)]
            if len(emitters) > 1:
                rows += [(f"lum-out-{f}-{k}", "last", "transmit", f, {name}) for k, name in enumerate(emitters)]
            elif emitters:
                rows.append((f"lum-out-{f}", "last", "transmit", f, set(emitters)))
        rows.append(("lost-solar", "last", "absorb", None, lights))
        if len(emitters) > 1:
            rows += [(f"lost-lum-{k}", "last", "absorb", None, {name}) for k, name in enumerate(emitters)]
        elif emitters:
            rows.append(("lost-lum", "last", "absorb", None, set(emitters)))
        if want_sources == set(all_sources):
            rows.append(("killed", "last", "kill", None, set(all_sources)))
        recs = self._result.recorders
        edges, total = recs["lost-solar"].histogram(0)
        total = np.zeros_like(total)
        for name, row_kind, event, facet, stands_for in rows:
            if kind is not None and row_kind != kind:
                continue
            if want_facets and facet not in want_facets:
                continue
            if events is not None and event not in events:
                continue
            if not stands_for <= want_sources:
                continue
            total = total + recs[name].histogram(0)[1]
        return Spectrum(edges, total)

    def show(self, wireframe=True, baubles=True, bauble_radius=None, world_segment="short", short_length=None,
             open_browser=False):
        """The reference opens its meshcat renderer here (device/lsc.py:301-336) and draws the ray paths of the next
        `simulate` into it.  There is no renderer in this package (the visualiser is outside the traced path): the call is
        accepted, so that a notebook written for the reference runs on, and says so once."""
        import warnings

        warnings.warn("pvtrace_amd has no renderer: LSC.show() draws nothing (histories of a few rays: "
                      "engine.simulate(lsc.scene, n).histories())", stacklevel=2)

    def report(self):
        print("\nSimulation Report\n-----------------\n\nSurface Counts:")
        print(self.counts())
        print("\nSummary:")
        print(self.summary())
