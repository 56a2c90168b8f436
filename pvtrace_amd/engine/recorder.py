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


class Histogram:
    """`bins` equal-width bins of `prop` over [start, stop)."""

    def __init__(self, prop, start, stop, bins):
        if prop not in PROPERTIES:
            raise ValueError(
                f"Unknown property {prop!r}; use one of {sorted(PROPERTIES)}"
            )
        if not stop > start:
            raise ValueError("Histogram range requires stop > start.")
        if bins < 1:
            raise ValueError("Histogram requires at least one bin.")
        self.prop = prop
        self.start = float(start)
        self.stop = float(stop)
        self.bins = int(bins)

    def __repr__(self):
        return f"Histogram({self.prop!r}, {self.start}, {self.stop}, {self.bins})"


class Heatmap:
    """2-D histogram over (prop_a, prop_b); ranges are (start, stop, bins)."""

    def __init__(self, prop_a, prop_b, range_a, range_b):
        self.a = Histogram(prop_a, *range_a)
        self.b = Histogram(prop_b, *range_b)

    def __repr__(self):
        return f"Heatmap({self.a!r}, {self.b!r})"


class Recorder:
    """Tally of rays interacting with a node.

    Parameters
    ----------
    name : str
        Key under which results are returned.
    event : str
        One of `EVENTS`.
    facet : 3-tuple, optional
        Restrict a surface recorder to interactions whose outward world normal
        equals this vector within `atol` per component.
    atol : float
    histograms : list of Histogram / Heatmap, optional
    source : None | "lights" | "components" | component name, optional
        Only tally photons last emitted by a light / by any luminophore or scatterer / by
        the named component (extension; splits e.g. "solar" from "luminescent" counts).

    Counts, moments and histograms are per *distinct* ray (first matching
    interaction); every matching interaction also increments `crossings`.
    """

    def __init__(self, name, event="entering", facet=None, atol=1e-6, histograms=None,
                 source=None):
        if event not in EVENTS:
            raise ValueError(f"Unknown event {event!r}; use one of {sorted(EVENTS)}")
        self.name = name
        self.event = event
        self.facet = None if facet is None else tuple(float(v) for v in facet)
        self.atol = float(atol)
        self.histograms = [] if histograms is None else list(histograms)
        self.source = source
        for hist in self.histograms:
            if not isinstance(hist, (Histogram, Heatmap)):
                raise ValueError("histograms must contain Histogram or Heatmap objects.")

    def __repr__(self):
        return f"Recorder({self.name!r}, event={self.event!r})"
