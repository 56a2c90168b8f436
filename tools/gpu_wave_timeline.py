"""Developer tool: read the per-wave stamps a -DPVT_TIMELINE=1 build wrote (PVT_TIMELINE_FILE) and print, per launch,
when workgroups started, when waves ran dry and when they ended (wall clock, 100 MHz)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read()
at = 0
while at + 32 <= len(data):
    magic, launch, grid, n = struct.unpack_from("<4Q", data, at); at += 32
    assert magic == 0xABCD
    w = np.frombuffer(data, dtype=np.uint64, count=grid * 4 * 8, offset=at).reshape(grid * 4, 8).astype(np.int64); at += grid * 4 * 8 * 8
    ok = w[:, 7] == 1
    w = w[ok]
    t0 = w[:, 0].min()
    us = lambda c: (c - t0) / 100.0
    start, staged, first, dry, end, iters = (w[:, k] for k in range(6))
    def pct(x): return " ".join(f"{v:7.1f}" for v in np.percentile(us(x), [0, 5, 25, 50, 75, 95, 100]))
    print(f"launch {launch} grid {grid} n {n}: waves {ok.sum()}  iterations/wave mean {iters.mean():.1f} (min {iters.min()}, max {iters.max()})")
    print("   wave start      us [min p5 p25 p50 p75 p95 max]:", pct(start))
    print("   tables staged   us                             :", pct(staged))
    print("   first step      us                             :", pct(first))
    d = dry[dry > 0]
    if len(d): print("   cursor dry      us                             :", pct(d), f"({len(d)} waves)")
    print("   wave end        us                             :", pct(end))
    life = (end - start) / 100.0
    print(f"   wave lifetime us: mean {life.mean():.1f} p5 {np.percentile(life,5):.1f} p95 {np.percentile(life,95):.1f}; "
          f"us per iteration: {(life.sum() / max(iters.sum(),1)):.2f}; kernel span {us(end).max():.1f} us; "
          f"sum(lifetimes)/(span x waves) = {life.sum() / (us(end).max() * len(life)):.3f}")
