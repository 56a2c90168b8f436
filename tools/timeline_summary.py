"""Developer tool: summarise rocprofv3 kernel traces (start/end per dispatch): the last 150 trace kernels of each
run -- duration, start-to-start spacing, how many run concurrently, and the gaps on each queue."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    rows = []
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r["Grid_Size"]) // 256 if "Grid_Size" in r else 0))
    rows.sort()
    tk = [r for r in rows if "trace_kernel" in r[2]]
    tail = tk[-170:-20]
    t0 = tail[0][0]
    dur = [(e - s) / 1e3 for s, e, *_ in tail]
    spacing = [(tail[i + 1][0] - tail[i][0]) / 1e3 for i in range(len(tail) - 1)]
    span = (tail[-1][1] - tail[0][0]) / 1e3
    busy = sum(dur)
    print(d.split("/")[-1], f"kernels {len(tail)}  mean duration {sum(dur)/len(dur):.1f} us  mean start spacing {sum(spacing)/len(spacing):.1f} us  "
          f"mean concurrency {busy/span:.2f}")
    byq = collections.defaultdict(list)
    for s, e, n, q, g in tail:
        byq[q].append((s, e))
    for q, v in byq.items():
        gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
        print(f"   queue {q}: {len(v)} kernels, gap between end and next start on this queue: mean {sum(gaps)/max(len(gaps),1):.1f} us (min {min(gaps):.1f}, max {max(gaps):.1f})")
    for s, e, n, q, g in tail[:12]:
        print(f"     start {(s - t0)/1e3:8.1f} us  end {(e - t0)/1e3:8.1f} us  dur {(e - s)/1e3:7.1f}  queue {q} grid {g}")
