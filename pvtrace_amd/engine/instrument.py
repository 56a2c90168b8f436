"""Default instrumentation of a node: the `record: true` shorthand of the reference's
scene spec (pvtrace/cli/parse.py:469-525 `auto_recorders`) as a plain function, plus the
dict form of the `recorders:` section (parse.py:528-551) — so a front-end can hand either
to the engine without this package knowing about YAML files."""
from pvtrace_amd.engine.recorder import Heatmap, Histogram, Recorder
from pvtrace_amd.geometry import Box

_WAVELENGTH = (300.0, 1000.0, 100)
_ANGLE = (0.0, 1.5708, 18)
_FACES = (("top", (0, 0, 1)), ("bottom", (0, 0, -1)), ("east", (1, 0, 0)), ("west", (-1, 0, 0)),
          ("north", (0, 1, 0)), ("south", (0, -1, 0)))


def auto_recorders(node):
    """Recorders `record: true` stands for: boxes get one escaping recorder per face
    (wavelength, angle and a position heatmap over the face) plus a volume-loss recorder;
    other shapes get a whole-surface escaping recorder and the loss recorder."""
    name = node.name
    recs = [Recorder(f"{name}-lost", event="lost", histograms=[Histogram("wavelength", *_WAVELENGTH)])]
    if isinstance(node.geometry, Box):
        size = [float(v) for v in node.geometry.size]
        half = [v / 2.0 for v in size]
        axes = "xyz"
        for label, facet in _FACES:
            axis = [i for i, v in enumerate(facet) if v != 0][0]
            u, v = [i for i in range(3) if i != axis]
            bins_u = max(10, min(60, int(size[u] * 10)))
            bins_v = max(10, min(60, int(size[v] * 10)))
            recs.append(Recorder(
                f"{name}-{label}", event="escaping", facet=facet,
                histograms=[Histogram("wavelength", *_WAVELENGTH), Histogram("angle", *_ANGLE),
                            Heatmap(axes[u], axes[v], (-half[u], half[u], bins_u),
                                    (-half[v], half[v], bins_v))]))
    else:
        recs.append(Recorder(f"{name}-escaping", event="escaping",
                             histograms=[Histogram("wavelength", *_WAVELENGTH),
                                         Histogram("angle", *_ANGLE)]))
    return recs


def face_recorders(prefix="", wavelength=(400, 800, 80)):
    """The tally set of the headline benchmark: one `escaping` recorder per box face
    (top/bottom = +-z, right/left = +-x, far/near = +-y) with a wavelength histogram (`wavelength` =
    (start, stop, bins), None for no histogram), then `lost`, `entering`, `reflected` and `killed` for the
    whole node.  Attach the list to a Box node: ``node.recorders = face_recorders()``."""
    faces = (("top", (0, 0, 1)), ("bottom", (0, 0, -1)), ("right", (1, 0, 0)), ("left", (-1, 0, 0)),
             ("far", (0, 1, 0)), ("near", (0, -1, 0)))
    recs = []
    for label, normal in faces:
        hs = [Histogram("wavelength", *wavelength)] if wavelength else []
        recs.append(Recorder(f"{prefix}{label}", event="escaping", facet=normal, histograms=hs))
    recs += [Recorder(f"{prefix}lost", event="lost"), Recorder(f"{prefix}entering", event="entering"),
             Recorder(f"{prefix}reflected", event="reflected"), Recorder(f"{prefix}killed", event="killed")]
    return recs


def instrument(node, explicit=()):
    """Attach `auto_recorders(node)` to `node`; recorders in `explicit` with the same
    name take precedence (as explicit spec entries do in the reference)."""
    taken = {r.name for r in explicit} | {r.name for r in node.recorders}
    node.recorders.extend(r for r in auto_recorders(node) if r.name not in taken)
    node.recorders.extend(r for r in explicit if r.name not in {q.name for q in node.recorders})
    return node


def recorders_from_spec(spec, nodes):
    """Build and attach recorders from the dict form of a spec's `recorders:` section:
    {name: {node, event, facet?, atol?, source?, histograms: {prop: [start, stop, bins] |
    position: [prop_a, prop_b, range_a, range_b]}}}.  `nodes` maps node names to Nodes."""
    for name, entry in spec.items():
        node_name = entry["node"]
        if node_name not in nodes:
            raise ValueError(f"Recorder {name!r}: unknown node {node_name!r}")
        hists = []
        for prop, values in (entry.get("histograms") or {}).items():
            if prop == "position":
                prop_a, prop_b, range_a, range_b = values
                hists.append(Heatmap(prop_a, prop_b, range_a, range_b))
            else:
                start, stop, bins = values
                hists.append(Histogram(prop, start, stop, bins))
        nodes[node_name].recorders.append(Recorder(
            name, event=entry["event"], facet=entry.get("facet"), atol=entry.get("atol", 1e-6),
            histograms=hists, source=entry.get("source")))
