"""Developer tool: static instruction counts per source-line bucket of pvt_trace.hip for one kernel variant."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else "trace_kernel_w4ILb0ELi1ELi1ELb0"
tmp = tempfile.mkdtemp(prefix="isal_")
src = os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_trace.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-g", "-std=c++17", "-ffp-contract=off",
                       "-fno-fast-math", "-munsafe-fp-atomics", "-mllvm", "-disable-machine-licm", "-fno-unroll-loops", "--cuda-device-only", "-S", src, "-o", os.path.join(tmp, "k.s")],
                      stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "k.s")).read()
files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)) or dict(re.findall(r'\.file\s+(\d+)\s+"([^"]+)"', s))
srclines = open(os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_trace_kernel.h")).read().split("\n")
mathlines = open(os.path.join(ROOT, "pvtrace_amd", "csrc", "pvt_math.h")).read().split("\n")
for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
    name = f.split("\n", 1)[0].strip()
    if variant not in name:
        continue
    body = f.split(".end_amdhsa_kernel")[0]
    cur = (None, 0)
    counts = collections.Counter(); valu = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (os.path.basename(files.get(m.group(1), m.group(1))), int(m.group(2)))
            continue
        m = re.match(r"^([a-z][a-z_0-9]+)\s", line)
        if m:
            counts[cur] += 1
            if m.group(1).startswith("v_"):
                valu[cur] += 1
    # bucket by file and 10-line windows -> print top lines
    print("total", sum(counts.values()), "valu", sum(valu.values()))
    byfile = collections.Counter()
    for (fn, ln), c in counts.items():
        byfile[fn] += c
    print(dict(byfile))
    top = sorted(counts.items(), key=lambda kv: -kv[1])[:70]
    for (fn, ln), c in top:
        text = ""
        if fn == "pvt_trace_kernel.h" and 0 < ln <= len(srclines): text = srclines[ln - 1].strip()[:90]
        if fn == "pvt_math.h" and 0 < ln <= len(mathlines): text = mathlines[ln - 1].strip()[:90]
        print(f"{c:5d} (valu {valu[(fn, ln)]:4d}) {fn}:{ln}: {text}")
