"""Developer tool: compile pvt_trace.hip to gfx950 assembly in /tmp and print an opcode
histogram + register usage for one kernel variant (default: the bench variant)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else "trace_kernel_w4ILb0ELi1ELi1ELb0E"
tmp = tempfile.mkdtemp(prefix="isa_")
src = os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_trace.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                       "-fno-fast-math", "-munsafe-fp-atomics", "-mllvm", "-disable-machine-licm", "-fno-unroll-loops", "--cuda-device-only", "-S", src, "-o",
                       os.path.join(tmp, "k.s")])
s = open(os.path.join(tmp, "k.s")).read()
for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
    name = f.split("\n", 1)[0].strip()
    if variant not in name:
        continue
    body = f.split(".end_amdhsa_kernel")[0] if ".end_amdhsa_kernel" in f else f
    ops = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        m = re.match(r"^([a-z][a-z_0-9]+)\s", line)
        if m:
            ops[m.group(1)] += 1
    groups = collections.Counter()
    for k, v in ops.items():
        g = ("s_load" if k.startswith("s_load") else "global_load" if k.startswith("global_load")
             else "global_other" if k.startswith("global_") else "ds" if k.startswith("ds_")
             else "v_f64" if "_f64" in k else "valu_other" if k.startswith("v_")
             else "salu" if k.startswith("s_") else "other")
        groups[g] += v
    print(name, "static instructions:", sum(ops.values()))
    print(dict(groups))
    print(ops.most_common(45))
    for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
        m = re.search(r"\.name:\s+" + re.escape(name) + r".*?\." + key + r":\s+(\d+)", s, flags=re.S)
    meta = s[s.find(".name:           " + name):][:1500] if (".name:           " + name) in s else ""
    print(re.findall(r"\.(sgpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\d+)", meta))
print("asm kept in", tmp)
