"""The C-ABI shared library: builds for gfx950 without a GPU, loads, exports every
symbol include/pvtrace_hip.h declares, agrees with the ctypes mirror on struct
layout, and fails LOUDLY (never falls back) when no GPU is present."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pvtrace_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pvt_[a-z_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol(built):
    from pvtrace_amd.engine import native

    names = declared_functions()
    assert len(names) >= 9 and "pvt_trace_bundle" in names and "pvt_trace_device" in names
    assert set(names) == set(native.ABI_SYMBOLS)
    lib = native.load_library()
    for name in names:
        assert hasattr(lib, name), name
    assert lib.pvt_abi_version() == 13 == native.ABI_VERSION
    # the embedded device code object really targets gfx950
    blob = open(native.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_ctypes_structs_match_the_header(tmp_path):
    from pvtrace_amd.engine import native as N

    structs = ["PvtSceneTables", "PvtEmitterTables", "PvtTraceParams", "PvtRays", "PvtTallies",
               "PvtEventLog", "PvtEventRecords"]
    probes = {"PvtSceneTables": ["n_nodes", "geom_type", "comp_type", "abs_x", "rec_node",
                                 "hist_prop_a", "coat_facet", "coat_transmit_mode", "rec_source_id", "comp_ems_hist"],
              "PvtEmitterTables": ["wl_type", "spec_cdf"],
              "PvtTraceParams": ["seed", "ray_offset", "record_every", "maxsteps", "emit_method"],
              "PvtRays": ["wavelength"], "PvtTallies": ["rec_bins"], "PvtEventLog": ["kind", "duration"],
              "PvtEventRecords": ["counts", "rows"]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for s in structs:
        lines.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in probes[s]:
            lines.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    for line in out.strip().splitlines():
        key, value = line.split()
        if "." in key:
            s, f = key.split(".")
            assert getattr(getattr(N, s), f).offset == int(value), key
        else:
            assert C.sizeof(getattr(N, key)) == int(value), key


def test_header_cites_the_reference_interface():
    text = open(HEADER).read()
    assert "_kernel.pyx:903-1115" in text and "compiler.py" in text and "extern \"C\"" in text


def test_no_gpu_means_loud_failure_not_fallback(built):
    from pvtrace_amd import engine
    from pvtrace_amd.engine import native
    from tests import scenes

    if native.is_available():
        pytest.skip("a GPU is visible here")
    assert engine.is_available() is False
    with pytest.raises(engine.EngineUnavailableError):
        engine.simulate(scenes.fresnel_box(), 10, seed=1)
    from pvtrace_amd.engine import _kernel, compile_scene
    c = compile_scene(scenes.fresnel_box())
    with pytest.raises(engine.EngineUnavailableError):
        _kernel.trace_bundle(c, np.zeros((1, 3)), np.tile((0.0, 0.0, 1.0), (1, 1)), np.full(1, 555.0),
                             1, 10, 8, 0, 1, 0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: no file of the product package may reference it."""
    pkg = os.path.join(ROOT, "pvtrace_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "pvt_oracle" in text.replace(
                        "oracle/pvt_oracle.c", "").replace("pvt_oracle_emit", ""):
                    offenders.append(os.path.join(dirpath, f))
    assert offenders == []


def test_shard_ranges_of_the_library_match_the_python_sharder(built):
    """pvt_shard_range (what pvt_trace_bundle_multi splits by) against engine.distributed.shard_range
    (what the one-process-per-GPU path splits by): contiguous, exhaustive, inner edges on multiples of
    record_every.  Pure host arithmetic."""
    from pvtrace_amd.engine import native as N
    from pvtrace_amd.engine.distributed import shard_range

    # (10^8 over 2 / 4 / 8 shards with record_every 0 / 1000 / 7: BASELINE configs[2], the job no 8-GPU node has run yet)
    for n in (0, 1, 63, 1000, 1_000_003, 100_000_000, 2 ** 31 - 1):
        for shards in (1, 2, 3, 4, 8):
            for align in (0, 1, 7, 1000):
                spans = [N.shard_range(n, g, shards, align) for g in range(shards)]
                assert spans == [shard_range(n, g, shards, align) for g in range(shards)]
                assert spans[0][0] == 0 and spans[-1][1] == n
                for (a, b), (c, d) in zip(spans, spans[1:]):
                    assert a <= b == c <= d and b % max(align, 1) == 0
    with pytest.raises(ValueError):
        N.shard_range(10, 3, 3)


def test_last_error_is_per_thread(built):
    """pvt_last_error() is thread-local: a worker thread's failure neither leaks into nor is overwritten by
    another thread's (the reference's consumer calls the engine from an executor thread, studio/server.py:229)."""
    import threading

    from pvtrace_amd.engine import native as N

    lib = N.load_library()
    a, b = C.c_int64(), C.c_int64()
    assert lib.pvt_shard_range(-1, 0, 1, 1, C.byref(a), C.byref(b)) != 0   # this thread's own last failure
    mine = lib.pvt_last_error().decode()
    seen = {}
    barrier = threading.Barrier(2)

    def shard_fails():
        a, b = C.c_int64(), C.c_int64()
        assert lib.pvt_shard_range(10, 5, 2, 1, C.byref(a), C.byref(b)) != 0
        barrier.wait()
        barrier.wait()      # the other thread has failed differently meanwhile
        seen["shard"] = lib.pvt_last_error().decode()

    def create_fails():
        barrier.wait()
        assert lib.pvt_scene_create(None, 0, None) != 0
        seen["create"] = lib.pvt_last_error().decode()
        barrier.wait()

    threads = [threading.Thread(target=shard_fails), threading.Thread(target=create_fails)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert "shard" in seen["shard"] and "null" in seen["create"]
    assert seen["shard"] != seen["create"]
    assert lib.pvt_last_error().decode() == mine   # untouched by the workers' failures


def test_event_record_layout_round_trips_on_the_host():
    """`native.decode_records` reads the 128-byte event records of include/pvtrace_hip.h (PvtEventRecords): build
    rows by hand in that layout and get the reference's columns back, dtypes included."""
    import numpy as np

    from pvtrace_amd.engine import native

    rng = np.random.default_rng(3)
    m = 257
    cols = {
        "hit": rng.integers(-1, 120, m).astype(np.int32), "container": rng.integers(-1, 120, m).astype(np.int32),
        "adjacent": rng.integers(-1, 120, m).astype(np.int32), "component": rng.integers(-1, 40, m).astype(np.int32),
        "source": rng.integers(-1, 40, m).astype(np.int32), "kind": rng.integers(0, 10, m).astype(np.uint8),
        "position": rng.normal(size=(m, 3)), "direction": rng.normal(size=(m, 3)), "normal": rng.normal(size=(m, 3)),
        "wavelength": rng.uniform(300, 900, m), "travelled": rng.uniform(0, 50, m), "duration": rng.uniform(0, 1e-8, m),
    }
    rows = np.zeros((m, native.RECORD_WORDS), dtype=np.int64)
    i32 = rows.view(np.int32).reshape(m, 2 * native.RECORD_WORDS)
    f64 = rows.view(np.float64).reshape(m, native.RECORD_WORDS)
    i32[:, 0], i32[:, 1], i32[:, 2], i32[:, 3], i32[:, 4] = (cols[k] for k in ("hit", "container", "adjacent", "component", "source"))
    i32[:, 5] = cols["kind"]
    f64[:, 3:6], f64[:, 6:9], f64[:, 9:12] = cols["position"], cols["direction"], cols["normal"]
    f64[:, 12], f64[:, 13], f64[:, 14] = cols["wavelength"], cols["travelled"], cols["duration"]
    rows[:, 15] = np.arange(m)
    got = native.decode_records(rows)
    assert set(got) == {name for name, _, _ in native.EVENT_LOG_COLUMNS}
    for name, dtype, width in native.EVENT_LOG_COLUMNS:
        assert got[name].dtype == np.dtype(dtype) and np.array_equal(got[name], cols[name]), name
        assert got[name].flags["C_CONTIGUOUS"]
    assert C.sizeof(native.PvtEventRecords) == 16
    text = open(HEADER).read()
    assert "typedef struct PvtEventRecords" in text and "_kernel.pyx:1035-1047" in text


def test_the_build_module_of_the_reference_has_its_counterpart():
    """reference `python -m pvtrace.engine.build` (engine/build.py; engine/__init__.py:11-13 tells users to run it): the same
    module name here runs the in-tree HIP build (a no-op when the library is newer than its sources)."""
    import importlib

    mod = importlib.import_module("pvtrace_amd.engine.build")
    mod.main()
    from pvtrace_amd.engine import native
    assert native.library_built()
