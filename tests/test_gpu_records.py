"""The event log as the kernel writes it (128-byte records, PvtEventRecords) and the column arrays made from
it (pvt_unpack_records_device, pvt_trace_device): both must reproduce the CPU referee's log bit for bit."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd.engine import _kernel, compile_scene, native
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
from tests.util import assert_bundles_identical

pytestmark = pytest.mark.gpu


def _case(name="lsc_equivalent", n=5000, every=3, max_events=24, seed=21):
    scene = scenes.ALL_SCENES[name]()
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=5)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 4, every, math_mode=O.MATH_PORTABLE)
    return compiled, (pos, dirs, wl), cpu


@pytest.mark.parametrize("name,every,max_events", [("lsc_equivalent", 1, 24), ("lsc_equivalent", 7, 300),
                                                   ("nested_cylinders", 3, 9), ("kitchen_sink", 2, 40)])
def test_records_decode_to_the_referee_log_and_unpack_to_its_columns(name, every, max_events):
    import torch

    n, seed = 6001, 21
    compiled, (pos, dirs, wl), cpu = _case(name, n, every, max_events, seed)
    dscene = native.DeviceScene(compiled, device=0)
    try:
        dev = torch.device("cuda", 0)
        rays = tuple(torch.from_numpy(a).to(dev) for a in (pos, dirs, wl))
        tallies = dscene.new_tallies()
        rec = dscene.new_event_log(n, every, max_events)
        rec["rows"].fill_(0x7777777777777777)      # rows no event reaches must stay untouched
        dscene.trace(rays, n, seed, tallies, log=rec, record_every=every, max_events=max_events)
        torch.cuda.synchronize()
        nrec = native.num_recorded(n, every)
        counts = rec["counts"][:nrec].cpu().numpy()
        assert np.array_equal(counts, cpu["counts"])
        rows = rec["rows"].cpu().numpy().reshape(nrec, max_events, native.RECORD_WORDS)
        written = np.arange(max_events)[None, :] < counts[:, None]
        assert np.all(rows[~written] == 0x7777777777777777)
        got = native.decode_records(rows[written])
        flat = written.reshape(-1)
        for key, value in got.items():
            assert np.array_equal(value, cpu[key][flat]), key
        assert np.array_equal(rows[written][:, 15], np.flatnonzero(flat))      # word 15: the row index
        # the column arrays, with and without the fill values
        for prefill in (True, False):
            cols = dscene.new_event_columns(n, every, max_events)
            for t in cols.values():
                t.fill_(77)
            dscene.unpack_records(rec, cols, nrec, max_events, prefill=prefill)
            torch.cuda.synchronize()
            for key, dtype, width in native.EVENT_LOG_COLUMNS:
                col = cols[key].cpu().numpy()
                col = col.reshape(-1, 3) if width == 3 else col
                want = cpu[key]
                if prefill:
                    assert np.array_equal(col, want), (key, prefill)
                else:
                    assert np.array_equal(col[flat], want[flat]) and np.all(col[~flat] == 77), (key, prefill)
    finally:
        dscene.close()


def test_a_small_staging_limit_splits_the_launch_without_changing_a_bit(monkeypatch):
    """pvt_trace_device stages the records of at most PVT_STAGE_BYTES per launch; a log larger than that is traced
    over consecutive ray ranges.  64 KiB holds 21 recorded rays of 24 events here: ~100 launches."""
    n, every, max_events, seed = 4001, 2, 24, 9
    compiled, (pos, dirs, wl), cpu = _case("lsc_equivalent", n, every, max_events, seed)
    whole = _kernel.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 1, every)
    monkeypatch.setenv("PVT_STAGE_BYTES", str(64 * 1024))
    split = _kernel.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 1, every)
    monkeypatch.delenv("PVT_STAGE_BYTES")
    assert_bundles_identical(whole, cpu, sums_rtol=1e-12, what="one launch")
    assert_bundles_identical(split, cpu, sums_rtol=1e-12, what="split launches")
