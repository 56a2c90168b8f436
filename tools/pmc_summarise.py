"""Summarise the PMC passes of tools/gpu_pmc.sh into profiles/<tag>_pmc_summary.json (+ the copy
bench.py reads, profiles/pmc_summary.json).  Usage: python tools/pmc_summarise.py r01_c "stage text"

Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB, and on gfx950 FETCH_SIZE is half the bytes of a wide coalesced
read stream, so it is doubled; WRITE_SIZE is uncalibrated and taken as is."""
import csv, glob, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
stage = sys.argv[2] if len(sys.argv) > 2 else ""
mode = sys.argv[3] if len(sys.argv) > 3 else "pipelined"
cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
if cfg != "cfg2":
    mode = f"{mode}_{cfg}"
photons = 1_000_000
sums = collections.defaultdict(list)
launches = {}
STEPS = 6   # tools/gpu_pmc.sh: --steps 6 --warmup 1
kernel = None
PASSES = ("fetch", "write", "sq1", "sq2", "sq3", "grbm")   # tools/gpu_pmc.sh (exact names: pmc_pipelined_* would also match pmc_pipelined_cfg5_*)
for path in [os.path.join(ROOT, "gpurun_out", f"pmc_{mode}_{p}", "pmc_counter_collection.csv") for p in PASSES]:
    if not os.path.exists(path):
        continue
    rows = [r for r in csv.DictReader(open(path)) if "trace_kernel" in r["Kernel_Name"]]
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        kernel = r["Kernel_Name"]
    for name, vals in per.items():
        # With photons carried between launches a launch finishes photons of its predecessors and the window ends
        # with launches that take no new rays: the honest unit is the WINDOW -- every dispatch after the warm-up
        # launch, divided by the steps of the window (6) -- not the single launch.
        vals = vals[1:] if len(vals) > 1 else vals           # drop the warm-up launch
        sums[name] = sum(vals) / STEPS
        launches[name] = len(vals)
c = dict(sums)
read_b = c["FETCH_SIZE"] * 1024 * 2
write_b = c["WRITE_SIZE"] * 1024
out = {
    "round": 6, "stage": stage, "mode": mode, "config": cfg,
    "command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python bench.py --gpus 1 --steps 6 --warmup 1 "
               "--no-cpu-baseline --repeats 0 --sustained-s 0 --total-photons 0 --spinup-s 0 --ray-buffers 2 --extra-configs none --config " + cfg
               + (" --streams 1" if mode.startswith("serial") else "") + " (one counter set per run; tools/gpu_pmc.sh " + mode.split("_")[0] + " " + cfg + "); "
               "rocprofv3 serialises dispatches while sampling counters",
    "kernel": kernel, "photons_per_launch": photons, "counters_mean_per_launch": c,
    "counters_are": f"sums over the {max(launches.values()) if launches else 0} trace-kernel dispatches of the 6-step window "
                    "(6 bundles + the launches that finish carried photons), divided by 6",
    "hbm_read_bytes_per_launch_corrected": read_b, "hbm_write_bytes_per_launch": write_b,
    "hbm_bytes_per_launch": read_b + write_b,
    "correction": "FETCH_SIZE doubled: gfx950 rocprofv3 reports 1/2 of a wide coalesced read stream "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated, taken as is",
    "algorithmic_bytes_per_launch": (56 if cfg == "cfg2" else 0) * photons,
    "derived": {
        "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64) if "SQ_ACTIVE_INST_VALU" in c else None,
        "valu_wave_instructions_per_photon": c.get("SQ_INSTS_VALU", 0) / photons,
        "salu_instructions_per_photon": c.get("SQ_INSTS_SALU", 0) / photons,
        "lds_instructions_per_photon": c.get("SQ_INSTS_LDS", 0) / photons,
        "wait_any_fraction_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"],
        "wait_inst_any_fraction": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
        "active_inst_any_fraction": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_fraction": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        # VALU pipes busy, measured: 4 cycles per wave64 instruction over (SIMDs x kernel cycles); GRBM_GUI_ACTIVE
        # is summed over the 8 XCDs
        "valu_busy_measured": (4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8.0)) if "GRBM_GUI_ACTIVE" in c else None,
        "smem_instructions_per_wave": c.get("SQ_INSTS_SMEM", 0) / c["SQ_WAVES"],
        "vmem_instructions_per_wave": c.get("SQ_INSTS_VMEM", 0) / c["SQ_WAVES"],
    },
}
# the bench line printed under the sq1 pass carries what the kernel counted itself (trips of the photon loop per photon);
# with it the vector instructions PER TRIP of a wave, which bench.py scales by the trips it counts in its own run
for cand in ("sq1", "sq2", "sq3"):
    line = os.path.join(ROOT, "gpurun_out", f"pmc_{mode}_{cand}.json")
    try:
        doc = json.loads(open(line).read().strip().splitlines()[-1])
        ik = (doc["roofline"]["instruction_side"] or {}).get("in_kernel")   # (--config <cfg>: the main leg IS that config)
        # the device code the counters were sampled on (sha256 of its .text, as the sampled process itself read it from the
        # library it had loaded): bench.py compares it with the library it runs on and reports `instruction_side.stale`
        out["kernel_text_hash"] = (doc["roofline"]["instruction_side"] or {}).get("built_kernel_text_hash")
        if ik:
            out["in_kernel_of_the_pmc_run"] = ik
            out["derived"]["wave_iterations_per_photon"] = ik["wave_iterations_per_photon"]
            out["derived"]["valu_wave_instructions_per_wave_iteration"] = (
                out["derived"]["valu_wave_instructions_per_photon"] / ik["wave_iterations_per_photon"])
            break
    except (OSError, ValueError, KeyError, IndexError, TypeError):
        continue
live = {"pipelined": "pmc_summary.json"}.get(mode, f"pmc_summary_{cfg}.json" if mode == f"pipelined_{cfg}" else None)
for name in ((f"{tag}_pmc_summary.json", live) if live else (f"{tag}_pmc_summary.json",)):
    with open(os.path.join(ROOT, "profiles", name), "w") as fp:
        json.dump(out, fp, indent=1)
print(json.dumps(out["derived"], indent=1), out["hbm_bytes_per_launch"])
