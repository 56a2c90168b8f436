"""Recorder statistics recomputed on the host from photon histories.

The device accumulates recorders inside the trace kernel (tally block of
csrc/pvt_trace_kernel.h).  This module answers the same question — "what should every recorder
of this scene hold?" — from `(ray, event, metadata)` histories such as `EngineResult.histories()`,
with the semantics of the reference's pvtrace/engine/tally.py:26-156.  The tests use it to prove
the kernel's accumulators exact on sampled runs.

Organisation: every recorder becomes a `_Probe` bound to its node.  Probes are filed under the
event kind that can fire them, each history is streamed once through the probes of its events,
a probe keeps the property values of the FIRST qualifying event of each ray, and moments and
histograms are computed from those columns with numpy at the end.
"""
import math

import numpy as np

from pvtrace_amd.engine.recorder import Heatmap
from pvtrace_amd.light import Event

# recorder selector -> (history event, metadata keys that must all name the recorder's node)
_TRIGGERS = {
    "entering": (Event.TRANSMIT, ("hit", "adjacent")),
    "escaping": (Event.TRANSMIT, ("hit", "container")),
    "reflected": (Event.REFLECT, ("hit", "adjacent")),
    "lost": (Event.NONRADIATIVE, ("container",)),
    "reacted": (Event.REACT, ("container",)),
    "killed": (Event.KILL, ("container",)),
    "exit": (Event.EXIT, ("hit",)),
}
_COLUMNS = ("wavelength", "angle", "duration", "pathlength", "x", "y", "z")


class _Probe:
    """One recorder attached to one node."""

    def __init__(self, node, recorder, root, component_names):
        self.node, self.recorder, self.root = node, recorder, root
        self.event, self.keys = _TRIGGERS[recorder.event]
        self.component_names = component_names
        self.crossings = 0
        self.rows = []          # one row of _COLUMNS per distinct ray
        self.open = True        # no crossing of the current ray counted yet

    # -- selection ---------------------------------------------------------------------
    def _from_wanted_source(self, source):
        want = getattr(self.recorder, "source", None)
        if want is None:
            return True
        emitted_by_component = source in self.component_names
        if want == "lights":
            return not emitted_by_component
        if want == "components":
            return emitted_by_component
        return source == want

    def _local(self, position):
        return tuple(position) if self.node is self.root else self.root.point_to_node(position, self.node)

    def offer(self, ray, meta, incoming):
        """Present one history event of this probe's kind; `incoming` is the ray as it arrived."""
        name = self.node.name
        if any(meta.get(key) != name for key in self.keys) or not self._from_wanted_source(ray.source):
            return
        normal = meta.get("normal")
        if normal is None and self.event == Event.EXIT:
            normal = self.node.vector_to_node(self.node.geometry.normal(self._local(ray.position)), self.root)
        facet = self.recorder.facet
        if facet is not None:
            if normal is None or max(abs(f - c) for f, c in zip(facet, normal)) > self.recorder.atol:
                return
        self.crossings += 1
        if not self.open:
            return
        self.open = False
        angle = 0.0
        if normal is not None:
            along = ray.direction if self.event == Event.EXIT else incoming.direction
            angle = math.acos(min(abs(float(np.dot(along, normal))), 1.0))
        x, y, z = self._local(ray.position)
        self.rows.append((ray.wavelength, angle, ray.duration, ray.travelled, x, y, z))

    # -- reduction ---------------------------------------------------------------------
    @staticmethod
    def _bin_indices(values, axis):
        """C-style truncation of (v - lo) / (hi - lo) * n, -1 outside [0, n) (tally.py:81-83)."""
        index = ((values - axis.start) / (axis.stop - axis.start) * axis.bins).astype(np.int64)
        return np.where((index >= 0) & (index < axis.bins), index, -1)

    def result(self):
        from pvtrace_amd.engine.api import RecorderResult

        table = np.array(self.rows, dtype=np.float64).reshape(len(self.rows), len(_COLUMNS))
        column = {name: table[:, k] for k, name in enumerate(_COLUMNS)}
        moments = np.zeros((4, 2))
        for k, name in enumerate(_COLUMNS[:4]):
            moments[k] = column[name].sum(), (column[name] * column[name]).sum()
        bins = []
        for spec in self.recorder.histograms:
            if isinstance(spec, Heatmap):
                ia, ib = self._bin_indices(column[spec.a.prop], spec.a), self._bin_indices(column[spec.b.prop], spec.b)
                inside = (ia >= 0) & (ib >= 0)
                flat = ia[inside] * spec.b.bins + ib[inside]
                bins.append(np.bincount(flat, minlength=spec.a.bins * spec.b.bins).astype(np.int64))
            else:
                index = self._bin_indices(column[spec.prop], spec)
                bins.append(np.bincount(index[index >= 0], minlength=spec.bins).astype(np.int64))
        return RecorderResult(self.recorder, len(self.rows), self.crossings, moments, bins)


def tally_histories(scene, histories):
    """{recorder name: RecorderResult} from one history per ray."""
    root = scene.root
    nodes = list(root.preorder())
    component_names = {component.name for node in nodes
                       if node.geometry is not None and node.geometry.material is not None
                       for component in node.geometry.material.components}
    probes = [_Probe(node, recorder, root, component_names)
              for node in nodes for recorder in getattr(node, "recorders", [])]
    by_event = {}
    for probe in probes:
        by_event.setdefault(probe.event, []).append(probe)

    for history in histories:
        for probe in probes:
            probe.open = True
        incoming = None
        for ray, event, meta in history:
            for probe in by_event.get(event, ()):
                probe.offer(ray, meta or {}, incoming or ray)
            incoming = ray
    return {probe.recorder.name: probe.result() for probe in probes}
