"""What the built library's kernels cost in registers, read from the code object inside libpvtrace_hip.so (no GPU
needed): the variants of analytic scenes must fit four waves per SIMD without spilling a vector register, and the
headline variant must stay under the scalar-spill budget the round-2 verdict set (<= 60; it was 137)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pvtrace_amd", "csrc", "libpvtrace_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
# bytes of private segment a kernel may have WITHOUT having spilled anything: the frames of the functions it calls
# (`tail_run`, the drain's last wave; `emit_chunk`): registers the callee saves on entry and restores on return
CALL_FRAME = 320


def _kernels(tmp_path):
    import __graft_entry__ as entry

    meta = entry.kernel_metadata(LIB)       # the same reader build() warns with
    if not meta:
        pytest.skip("library or LLVM binutils not present")
    return meta


def test_register_budgets_of_the_built_kernels(tmp_path):
    kernels = _kernels(tmp_path)
    analytic = {n: m for n, m in kernels.items() if "trace_kernel_w4" in n}
    mesh = {n: m for n, m in kernels.items() if "12trace_kernelILb" in n}
    grid = {n: m for n, m in kernels.items() if "trace_kernel_grid" in n}
    # {tally, history} x {tables in LDS, in global memory[, records in LDS and spectra in global memory: analytic scenes only]}
    # x {64, 256 recorders} x {rays, emitter}
    assert len(analytic) == 24 and len(mesh) == 16
    assert len(grid) == 8                                    # many-node scenes (tables in LDS): {tally, history} x {64, 256 recorders} x {rays, emitter}
    for name, m in grid.items():
        # four waves per SIMD; the walk of the node grid keeps nearest / second-nearest crossing, the cells' bookkeeping and
        # the photon in registers: nothing in scratch with <= 64 recorders (a scratch access inside the walk cost a third
        # of the throughput when the compiler put the fold's state there), a handful of spilled registers with 256
        assert m["vgpr_count"] <= 128, (name, m)
        if "ELi1E" in name and "gridILb0E" in name:   # tally launches: the walk's state all in registers
            assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] <= CALL_FRAME, (name, m)
        elif "ELi1E" in name:   # history launches carry the log cursor and (round 5) the step counters on top: a register or two
            assert m["vgpr_spill_count"] <= 4, (name, m)
        else:
            assert m["vgpr_spill_count"] <= 16, (name, m)
    for name, m in analytic.items():
        # four waves per SIMD (<= 128 registers), no register of the loop in scratch (the private segment that is left
        # is the frame of the functions the kernel calls -- the tail function's saved registers -- touched once per call)
        # (two history variants park ONE register around the call of the tail function, at the very end of the kernel --
        # the only scratch instructions of those kernels, either side of the s_swappc: checked in the ISA, round 5)
        assert m["vgpr_count"] <= 128 and m["vgpr_spill_count"] <= (1 if "w4ILb1E" in name else 0) and \
            m["private_segment_fixed_size"] <= CALL_FRAME, (name, m)
    for name, m in mesh.items():
        assert m["vgpr_count"] <= 128, (name, m)             # held to four waves; what does not fit is parked in scratch
    headline = [m for n, m in analytic.items() if "w4ILb0ELi1ELi1ELb0E" in n]      # tally, tables in LDS, <= 64 recorders, rays in
    assert len(headline) == 1 and headline[0]["sgpr_spill_count"] <= 60, headline
    for name, m in kernels.items():
        if "trace_kernel" not in name:                       # emit / unpack / pack / self-test kernels
            assert m["vgpr_spill_count"] == 0 and m["sgpr_spill_count"] == 0, (name, m)
    # Code size against the 64 KB instruction cache a pair of CUs shares: every variant of analytic scenes within 58 KB,
    # the emitter variants included -- the light sampler is a FUNCTION of its own (`emit_chunk`, 14 KB),
    # called by the wave that claims a chunk of rays, not code inlined into the step loop (where it ran every iteration:
    # -6 ... -16 % throughput) or into the kernel's text (66-69 KB)
    for name, m in {**analytic, **grid}.items():
        assert m["text_bytes"] <= 58 * 1024, (name, m)
        assert m["private_segment_fixed_size"] <= CALL_FRAME or "ELi4E" in name, (name, m)   # the calls cost a frame, no spills


def test_build_warns_when_the_headline_variant_leaves_its_budget():
    import __graft_entry__ as entry

    if not entry.kernel_metadata(LIB):
        pytest.skip("library or LLVM binutils not present")
    assert entry.check_kernel_budgets() == []
