"""Recorder tallies computed in pure Python from (ray, event, metadata) histories.

Host-side mirror of the device tally (the tally block of csrc/pvt_trace_kernel.h), with
the semantics of the reference's pvtrace/engine/tally.py:26-156: it re-derives
what every recorder should hold from `EngineResult.histories()` (or any list of
histories) and is used by the tests to prove the kernel's accumulators exact.
"""
import math

import numpy as np

from pvtrace_amd.engine.recorder import Heatmap
from pvtrace_amd.light import Event

_SURFACE_SELECTOR = {"entering": "adjacent", "escaping": "container"}
_VOLUME_EVENT = {"lost": Event.NONRADIATIVE, "reacted": Event.REACT, "killed": Event.KILL}


def _source_ok(recorder, ray_source, component_names):
    want = getattr(recorder, "source", None)
    if want is None:
        return True
    from_component = ray_source in component_names
    if want == "lights":
        return not from_component
    if want == "components":
        return from_component
    return ray_source == want


def _fires(recorder, name, event, meta):
    kind = recorder.event
    if event == Event.TRANSMIT:
        key = _SURFACE_SELECTOR.get(kind)
        return key is not None and meta.get("hit") == name and meta.get(key) == name
    if event == Event.REFLECT:
        return kind == "reflected" and meta.get("hit") == name and meta.get("adjacent") == name
    if kind in _VOLUME_EVENT:
        return event == _VOLUME_EVENT[kind] and meta.get("container") == name
    if event == Event.EXIT:
        return kind == "exit" and meta.get("hit") == name
    return False


def _bin(value, spec):
    index = int((value - spec.start) / (spec.stop - spec.start) * spec.bins)
    return index if 0 <= index < spec.bins else -1


class _State:
    def __init__(self, recorder):
        self.rays = 0
        self.crossings = 0
        self.moments = np.zeros((4, 2))
        self.bins = [
            np.zeros(h.a.bins * h.b.bins if isinstance(h, Heatmap) else h.bins, dtype=np.int64)
            for h in recorder.histograms
        ]

    def add(self, recorder, values):
        self.rays += 1
        for k, prop in enumerate(("wavelength", "angle", "duration", "pathlength")):
            self.moments[k, 0] += values[prop]
            self.moments[k, 1] += values[prop] * values[prop]
        for spec, bins in zip(recorder.histograms, self.bins):
            if isinstance(spec, Heatmap):
                ia, ib = _bin(values[spec.a.prop], spec.a), _bin(values[spec.b.prop], spec.b)
                if ia >= 0 and ib >= 0:
                    bins[ia * spec.b.bins + ib] += 1
            else:
                i = _bin(values[spec.prop], spec)
                if i >= 0:
                    bins[i] += 1


def tally_histories(scene, histories):
    """dict recorder name -> RecorderResult, from one history per ray."""
    from pvtrace_amd.engine.api import RecorderResult

    root = scene.root
    slots = [(node, rec, _State(rec)) for node in root.preorder()
             for rec in getattr(node, "recorders", [])]

    def to_local(node, position):
        return tuple(position) if node is root else root.point_to_node(position, node)

    component_names = {c.name for n in root.preorder() if n.geometry is not None
                       and n.geometry.material is not None for c in n.geometry.material.components}
    for history in histories:
        seen = set()
        previous = None
        for ray, event, meta in history:
            meta = meta or {}
            for node, rec, state in slots:
                if not _fires(rec, node.name, event, meta):
                    continue
                if not _source_ok(rec, ray.source, component_names):
                    continue
                normal = meta.get("normal")
                if event == Event.EXIT and normal is None:
                    normal = node.vector_to_node(
                        node.geometry.normal(to_local(node, ray.position)), root)
                if rec.facet is not None:
                    if normal is None or any(
                            abs(f - c) > rec.atol for f, c in zip(rec.facet, normal)):
                        continue
                state.crossings += 1
                if rec.name in seen:
                    continue
                seen.add(rec.name)
                incident = ray.direction if event == Event.EXIT else (previous or ray).direction
                angle = 0.0
                if normal is not None:
                    angle = math.acos(min(abs(float(np.dot(incident, normal))), 1.0))
                local = to_local(node, ray.position)
                state.add(rec, {
                    "wavelength": ray.wavelength, "angle": angle, "duration": ray.duration,
                    "pathlength": ray.travelled, "x": local[0], "y": local[1], "z": local[2],
                })
            previous = ray
    return {rec.name: RecorderResult(rec, st.rays, st.crossings, st.moments, st.bins)
            for _, rec, st in slots}
