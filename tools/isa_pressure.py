"""Developer tool: which source lines touch the highest-numbered VGPRs of one kernel variant
(a cheap pointer at the register-pressure peak)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else "trace_kernel_w4ILb0ELi1ELi1ELb0E"
floor = int(sys.argv[2]) if len(sys.argv) > 2 else 112
tmp = tempfile.mkdtemp(prefix="isap_")
src = os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_trace.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-g", "-std=c++17", "-ffp-contract=off",
                       "-fno-fast-math", "-munsafe-fp-atomics", "-mllvm", "-disable-machine-licm", "-fno-unroll-loops", "--cuda-device-only", "-S", src, "-o", os.path.join(tmp, "k.s")],
                      stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "k.s")).read()
files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)) or dict(re.findall(r'\.file\s+(\d+)\s+"([^"]+)"', s))
lines = {"pvt_trace_kernel.h": open(os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_trace_kernel.h")).read().split("\n"),
         "pvt_math.h": open(os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_math.h")).read().split("\n")}
for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
    name = f.split("\n", 1)[0].strip()
    if variant not in name:
        continue
    body = f.split(".end_amdhsa_kernel")[0]
    cur = (None, 0)
    hits = collections.Counter()
    order = []
    for line in body.split("\n"):
        line = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (os.path.basename(files.get(m.group(1), m.group(1))), int(m.group(2)))
            continue
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", line)] + [int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", line)]
        if regs and max(regs) >= floor:
            if cur not in hits:
                order.append(cur)
            hits[cur] += 1
    for key in order:
        fn, ln = key
        text = lines.get(fn, [""] * (ln + 1))[ln - 1].strip()[:110] if fn in lines and ln > 0 else ""
        print(f"{hits[key]:4d}  {fn}:{ln}: {text}")
    break
