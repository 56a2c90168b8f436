"""Recorder (tally) specifications attached to scene nodes.

A recorder counts the rays that interact with its node in one particular way
and accumulates moments / histograms of their properties; memory scales with
the number of bins, never with the number of photons.  On the MI355X engine
the accumulators live in LDS per workgroup and are flushed with one atomic per
slot per workgroup, then summed across GPUs with a single RCCL all-reduce.

Numeric ids are part of the device ABI (include/pvtrace_hip.h) and equal the
reference's (pvtrace/engine/recorder.py:33-53; kernel side _kernel.pyx:166-172,
:482-498).
"""

# property id -> what the histogram axis measures.  x/y/z are in the frame of
# the node that owns the recorder.
PROPERTIES = {
    name: code
    for code, name in enumerate(
        ("wavelength", "angle", "duration", "pathlength", "x", "y", "z")
    )
}

# selector id -> which interaction fires the recorder.
#   surface: entering (transmit in from outside), escaping (transmit out from
#            inside), reflected (bounced off the outside)
#   volume : lost (non-radiative absorption), reacted (Reactor), killed
#   root   : exit (left the scene through the root surface)
EVENTS = {
    name: code
    for code, name in enumerate(
        ("entering", "escaping", "reflected", "lost", "reacted", "killed", "exit")
    )
}

VOLUME_EVENTS = frozenset(("lost", "reacted", "killed"))

# optional source filter (EXTENSION; the reference's recorders have none, its CLI count
# queries filter by source instead, pvtrace/cli/db.py:61-83): which emitter the photon's
# current incarnation came from
SOURCE_ANY, SOURCE_LIGHTS, SOURCE_COMPONENTS, SOURCE_COMPONENT = 0, 1, 2, 3


def _require(condition, message):
    if not condition:
        raise ValueError(message)


class Histogram:
    """Equal-width binning of one ray property: `bins` bins over [start, stop).

    Same constructor and attributes (`prop`, `start`, `stop`, `bins`) as the reference's
    Histogram (pvtrace/engine/recorder.py:56-72); the flattener turns it into one row of the
    `hist_*` tables."""

    __slots__ = ("prop", "start", "stop", "bins")

    def __init__(self, prop, start, stop, bins):
        _require(prop in PROPERTIES, f"Unknown property {prop!r}; use one of {sorted(PROPERTIES)}")
        lo, hi, count = float(start), float(stop), int(bins)
        _require(hi > lo, "Histogram range requires stop > start.")
        _require(count >= 1, "Histogram requires at least one bin.")
        self.prop, self.start, self.stop, self.bins = prop, lo, hi, count

    @property
    def size(self):
        """Number of accumulator slots."""
        return self.bins

    def __repr__(self):
        return "Histogram(%r, %s, %s, %d)" % (self.prop, self.start, self.stop, self.bins)


class Heatmap:
    """Two properties binned jointly; each range is `(start, stop, bins)`.  Slot of a sample is
    `ia * b.bins + ib` (reference recorder.py:75-83, kernel _kernel.pyx:540-553)."""

    __slots__ = ("a", "b")

    def __init__(self, prop_a, prop_b, range_a, range_b):
        self.a, self.b = Histogram(prop_a, *range_a), Histogram(prop_b, *range_b)

    @property
    def size(self):
        return self.a.bins * self.b.bins

    def __repr__(self):
        return "Heatmap(%r, %r)" % (self.a, self.b)


class Recorder:
    """What to count at a node (reference recorder.py:86-117, plus `source`).

    name        key of the result in `EngineResult.recorders`
    event       one of `EVENTS`: "entering" / "escaping" / "reflected" at the node's surface,
                "lost" / "reacted" / "killed" inside it, "exit" on the root
    facet       optional outward world normal a surface interaction must have (each component
                within `atol`) -- one recorder per face of a box, say
    histograms  `Histogram` / `Heatmap` specs filled by the first matching interaction of a ray
    source      None, "lights", "components" or a component name: only photons whose current
                incarnation was emitted there (extension; splits "solar" from "luminescent")

    A ray is counted once per recorder (`rays`, moments, histograms) however often it
    matches; `crossings` counts every match.
    """

    def __init__(self, name, event="entering", facet=None, atol=1e-6, histograms=None,
                 source=None):
        _require(event in EVENTS, f"Unknown event {event!r}; use one of {sorted(EVENTS)}")
        specs = list(histograms or ())
        _require(all(isinstance(spec, (Histogram, Heatmap)) for spec in specs),
                 "histograms must contain Histogram or Heatmap objects.")
        self.name = name
        self.event = event
        self.facet = tuple(map(float, facet)) if facet is not None else None
        self.atol = float(atol)
        self.histograms = specs
        self.source = source

    @property
    def is_volume(self):
        return self.event in VOLUME_EVENTS

    def __repr__(self):
        return "Recorder(%r, event=%r)" % (self.name, self.event)
