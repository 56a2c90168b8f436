"""Scene specification (YAML / dict, version "1.0") -> `Scene`.

Front-end for the engine: accepts the reference's scene-spec vocabulary
(pvtrace/cli/parse.py:83-466; the format of examples/studio_lsc.yml): a `components`
section (absorber / scatterer / luminophore, constant coefficient or a named / CSV spectrum
scaled to a peak coefficient), a `nodes` section (box / sphere / cylinder with a material, or a
light with wavelength / position / direction masks; `parent`, `location`, `direction`), the
`record: true` shorthand and an explicit `recorders` section.  The spec is walked by a small
builder class rather than the reference's nested closures; `mesh` nodes load STL files (the
reference goes through trimesh's loaders, cli/parse.py:130-138).

    scene = pvtrace_amd.spec.load("scene.yml")
    result = pvtrace_amd.engine.simulate(scene, 10**6, record_every=0)
"""
import os

import numpy as np

from pvtrace_amd.data import fluro_red, lumogen_f_red_305
from pvtrace_amd.engine.compiler import UnsupportedSceneError
from pvtrace_amd.engine.instrument import auto_recorders, recorders_from_spec
from pvtrace_amd.geometry import Box, Cylinder, Mesh, Sphere
from pvtrace_amd.light import (
    CircularMask, ConstantWavelengthMask, CubeMask, Light, RectangularMask, SpectrumWavelengthMask,
)
from pvtrace_amd.material import (
    Absorber, Cone, Distribution, HenyeyGreenstein, Luminophore, Material, Scatterer, isotropic,
    lambertian,
)
from pvtrace_amd.scene import Node, Scene

NAMED_SPECTRA = {"lumogen-f-red-305": lumogen_f_red_305, "fluro-red": fluro_red}
SUPPORTED_VERSIONS = ("1.0",)


class SpecError(ValueError):
    """The scene specification is malformed."""


def load(source, base=None):
    """`source`: path of a YAML file, a YAML string containing a newline, or a dict.  `base`: the directory relative
    `file:` entries are resolved against (default: the YAML file's own directory, else the working directory)."""
    given = base
    base = os.getcwd()
    if isinstance(source, dict):
        spec = source
    else:
        import yaml

        if isinstance(source, str) and "\n" not in source:
            base = os.path.dirname(os.path.abspath(source))
            with open(source, "r") as fp:
                spec = yaml.safe_load(fp)
        else:
            spec = yaml.safe_load(source)
    return _Builder(spec, given if given is not None else base).scene()


parse = load  # the reference's name for it (pvtrace/cli/parse.py:72)


class _Builder:
    def __init__(self, spec, base):
        if not isinstance(spec, dict) or "nodes" not in spec:
            raise SpecError("a scene spec needs a `nodes` section")
        version = str(spec.get("version", "1.0"))
        if version not in SUPPORTED_VERSIONS:
            raise SpecError(f"Version {version} not supported")
        self.spec = spec
        self.base = base
        self.components = {name: self.component(name, entry)
                           for name, entry in (spec.get("components") or {}).items()}

    # -- spectra -----------------------------------------------------------
    def spectrum(self, entry, kind):
        """(n, 2) array from {file: csv} or {name: ..., range: {min, max, spacing}}."""
        if entry is None:
            return None
        if "file" in entry:
            path = entry["file"]
            if not os.path.isabs(path):
                path = os.path.join(self.base, path)
            table = np.genfromtxt(path, delimiter=",", skip_header=1)
            if table.ndim != 2 or table.shape[1] < 2:
                raise SpecError(f"{path}: expected CSV columns x, y")
            # reference reads columns 0 (index), 1, 2 and keeps the two after the index
            cols = table[:, 1:3] if table.shape[1] >= 3 else table[:, 0:2]
            return np.array(cols, dtype=float)
        if "name" in entry:
            if entry["name"] not in NAMED_SPECTRA:
                raise SpecError(f"unknown spectrum {entry['name']!r}; have {sorted(NAMED_SPECTRA)}")
            rng = entry["range"]
            x = np.arange(rng["min"], rng["max"] + rng["spacing"], rng["spacing"])
            module = NAMED_SPECTRA[entry["name"]]
            y = module.absorption(x) if kind == "absorption" else module.emission(x)
            return np.column_stack((x, y))
        raise SpecError("a spectrum needs `file` or `name`")

    @staticmethod
    def scaled(spectrum, coefficient):
        """Scale a line shape to a peak `coefficient` (cm^-1)."""
        out = np.array(spectrum, dtype=float)
        out[:, 1] = out[:, 1] / np.max(out[:, 1]) * float(coefficient)
        return out

    # -- angular / spatial distributions -------------------------------------
    @staticmethod
    def direction(entry):
        if isinstance(entry, str):
            entry = {entry: {}}
        if "isotropic" in entry:
            return isotropic
        if "lambertian" in entry:
            return lambertian
        if "cone" in entry:
            return Cone(float(np.radians(float(entry["cone"]["half-angle"]))))
        if "henyey-greenstein" in entry:
            return HenyeyGreenstein(float(entry["henyey-greenstein"]["g"]))
        raise SpecError(f"unknown direction / phase function {entry!r}")

    @staticmethod
    def position(entry):
        if "rect" in entry:
            return RectangularMask(*entry["rect"])
        if "cube" in entry:
            return CubeMask(*entry["cube"])
        if "circle" in entry:
            return CircularMask(entry["circle"])
        raise SpecError(f"unknown position mask {entry!r}")

    def wavelength(self, entry):
        if "nanometers" in entry:
            return ConstantWavelengthMask(float(entry["nanometers"]))
        if "spectrum" in entry:
            table = self.spectrum(entry["spectrum"], "absorption")
            return SpectrumWavelengthMask(Distribution(table[:, 0], table[:, 1]))
        raise SpecError(f"unknown wavelength mask {entry!r}")

    # -- components ----------------------------------------------------------
    def component(self, name, entry):
        if "absorber" in entry:
            e = entry["absorber"]
            table = self.spectrum(e.get("spectrum"), "absorption")
            coefficient = e.get("coefficient")
            if table is not None:
                data = self.scaled(table, coefficient) if coefficient else table
                return Absorber(data, name=name, hist=bool(e.get("hist", False)))
            if coefficient is None:
                raise SpecError(f"absorber {name!r}: needs `coefficient` or `spectrum`")
            return Absorber(float(coefficient), name=name)
        if "scatterer" in entry:
            e = entry["scatterer"]
            table = self.spectrum(e.get("spectrum"), "absorption")
            coefficient = e.get("coefficient")
            phase = self.direction(e["phase-function"]) if "phase-function" in e else None
            kwargs = dict(quantum_yield=e.get("quantum-yield", 1.0), phase_function=phase, name=name,
                          hist=bool(e.get("hist", False)))
            if table is not None:
                return Scatterer(self.scaled(table, coefficient) if coefficient else table, **kwargs)
            if coefficient is None:
                raise SpecError(f"scatterer {name!r}: needs `coefficient` or `spectrum`")
            return Scatterer(float(coefficient), **kwargs)
        if "luminophore" in entry:
            e = entry["luminophore"]
            absorb, emit = e.get("absorption") or {}, e.get("emission") or {}
            emission = self.spectrum(emit.get("spectrum"), "emission")
            if emission is None:
                raise SpecError("Luminophore must have an emission spectrum")
            table = self.spectrum(absorb.get("spectrum"), "absorption")
            coefficient = absorb.get("coefficient")
            phase = self.direction(emit["phase-function"]) if "phase-function" in emit else isotropic
            kwargs = dict(emission=emission, quantum_yield=emit.get("quantum-yield", 1.0),
                          phase_function=phase, name=name, hist=bool(e.get("hist", False)))
            if table is not None:
                return Luminophore(self.scaled(table, coefficient) if coefficient else table, **kwargs)
            if coefficient is None:
                raise SpecError(f"luminophore {name!r}: needs an absorption coefficient or spectrum")
            return Luminophore(float(coefficient), **kwargs)
        raise SpecError(f"component {name!r}: unknown type")

    # -- nodes ---------------------------------------------------------------
    def material(self, entry):
        keys = entry.get("components") or []
        missing = [k for k in keys if k not in self.components]
        if missing:
            raise SpecError(f"Missing {missing[0]} component")
        return Material(refractive_index=entry["refractive-index"],
                        components=[self.components[k] for k in keys])

    def node(self, name, entry):
        if "box" in entry:
            g = entry["box"]
            return Node(name=name, geometry=Box(g["size"], material=self.material(g["material"])))
        if "sphere" in entry:
            g = entry["sphere"]
            return Node(name=name, geometry=Sphere(g["radius"], material=self.material(g["material"])))
        if "cylinder" in entry:
            g = entry["cylinder"]
            return Node(name=name, geometry=Cylinder(g["length"], g["radius"],
                                                     material=self.material(g["material"])))
        if "mesh" in entry:   # reference: trimesh.exchange.load (cli/parse.py:130-138); here STL only
            g = entry["mesh"]
            path = g["file"] if os.path.isabs(g["file"]) else os.path.join(self.base, g["file"])
            if not path.lower().endswith(".stl"):
                raise UnsupportedSceneError(f"Node {name!r}: only STL mesh files are supported.")
            return Node(name=name, geometry=Mesh.from_file(path, material=self.material(g["material"])))
        if "light" in entry:
            e = entry["light"]
            wavelength = ConstantWavelengthMask(e["wavelength"]) if e.get("wavelength") else None
            position = direction = None
            mask = e.get("mask") or {}
            if mask.get("wavelength"):
                wavelength = self.wavelength(mask["wavelength"])
            if mask.get("position"):
                position = self.position(mask["position"])
            if mask.get("direction"):
                direction = self.direction(mask["direction"])
            return Node(name=name, light=Light(wavelength=wavelength, position=position,
                                               direction=direction, name=name))
        raise SpecError(f"node {name!r}: needs a geometry (box / sphere / cylinder) or a light")

    def scene(self):
        entries = self.spec["nodes"]
        if "world" not in entries:
            raise SpecError("the root node must be called `world`")
        nodes = {name: self.node(name, entry) for name, entry in entries.items()}
        for name, entry in entries.items():
            node = nodes[name]
            if name != "world":
                parent = entry.get("parent") or "world"
                if parent not in nodes:
                    raise SpecError(f"node {name!r}: unknown parent {parent!r}")
                node.parent = nodes[parent]
            if entry.get("location"):
                node.location = entry["location"]
            if entry.get("direction"):
                node.look_at(entry["direction"])
        explicit = dict(self.spec.get("recorders") or {})
        recorders_from_spec(explicit, nodes)
        for name, entry in entries.items():
            if entry.get("record"):
                taken = {r.name for n in nodes.values() for r in n.recorders}
                nodes[name].recorders.extend(r for r in auto_recorders(nodes[name]) if r.name not in taken)
        return Scene(nodes["world"])
