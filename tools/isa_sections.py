"""Developer tool: static VALU/SALU/LDS instruction counts of the bench variant per source line of
pvt_trace_kernel.h (+ inlined pvt_math.h attributed to the kernel line that called it, via the
inlined-at chain approximated by 'last kernel-header line seen').  usage: isa_sections.py [src_root]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = "trace_kernel_w4ILb0ELi1ELi1ELb0E"
tmp = tempfile.mkdtemp(prefix="isas_")
src = os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_trace.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-g", "-std=c++17", "-ffp-contract=off",
                       "-fno-fast-math", "-munsafe-fp-atomics", "-mllvm", "-disable-machine-licm", "-fno-unroll-loops", "-DPVT_DEV_VARIANTS=1", "--cuda-device-only", "-S", src, "-o", os.path.join(tmp, "k.s")],
                      stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "k.s")).read()
files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)) or dict(re.findall(r'\.file\s+(\d+)\s+"([^"]+)"', s))
for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
    name = f.split("\n", 1)[0].strip()
    if variant not in name:
        continue
    body = f.split(".end_amdhsa_kernel")[0]
    cur_k = 0          # last line of pvt_trace_kernel.h seen
    cur = ("", 0)
    valu = collections.Counter(); salu = collections.Counter(); lds = collections.Counter()
    mathv = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            fn = os.path.basename(files.get(m.group(1), m.group(1)))
            cur = (fn, int(m.group(2)))
            if fn == "pvt_trace_kernel.h":
                cur_k = int(m.group(2))
            continue
        m = re.match(r"^([a-z][a-z_0-9]+)\s", line)
        if m:
            op = m.group(1)
            if op.startswith("v_"):
                valu[cur_k] += 1
                if cur[0] == "pvt_math.h": mathv[cur_k] += 1
            elif op.startswith("s_"):
                salu[cur_k] += 1
            elif op.startswith("ds_"):
                lds[cur_k] += 1
    lines = open(os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_trace_kernel.h")).read().split("\n")
    print("total valu", sum(valu.values()), "salu", sum(salu.values()), "lds", sum(lds.values()))
    for ln in sorted(set(valu) | set(salu) | set(lds)):
        text = lines[ln - 1].strip()[:100] if 0 < ln <= len(lines) else ""
        print(f"{ln:5d} v{valu[ln]:4d} (m{mathv[ln]:4d}) s{salu[ln]:4d} l{lds[ln]:3d} | {text}")
