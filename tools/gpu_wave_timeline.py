"""Developer tool: read the per-wave stamps a -DPVT_TIMELINE=1 build wrote (PVT_TIMELINE_FILE): per launch when
workgroups started / ran dry / ended, and over all recorded launches the measured shader clock, the wave-slot
occupancy and the iteration rate -- under whatever overlap the run had (nothing is serialised)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read()
verbose = len(sys.argv) > 2
at = 0
allw = []
while at + 32 <= len(data):
    magic, launch, grid, n = struct.unpack_from("<4Q", data, at); at += 32
    assert magic == 0xABCD
    w = np.frombuffer(data, dtype=np.uint64, count=grid * 4 * 8, offset=at).reshape(grid * 4, 8).astype(np.int64); at += grid * 4 * 8 * 8
    w = w[(w[:, 7] & 1) == 1]
    if len(w) == 0:
        continue
    allw.append(w)
    if verbose:
        t0 = w[:, 0].min()
        us = lambda c: (c - t0) / 100.0
        pct = lambda x: " ".join(f"{v:7.1f}" for v in np.percentile(us(x), [0, 5, 50, 95, 100]))
        mhz = w[:, 6].sum() / ((w[:, 4] - w[:, 0]).sum() / 100.0)
        print(f"launch {launch} grid {grid} n {n}: waves {len(w)} iterations/wave {(w[:,5] & 0xffffffff).mean():.1f}; shader clock {mhz:.0f} MHz; "
              f"first step us [min p5 p50 p95 max] {pct(w[:,2])}; cursor dry {pct(w[w[:,3] > 0][:,3])}; end {pct(w[:,4])}")
        # where the waves that end last ran: the HW_ID register (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
        hw = (w[:, 7] >> 32) & 0xffffffff
        xcc = (w[:, 7] >> 2) & 15
        simd, cu, se = (hw >> 4) & 3, (hw >> 8) & 31, ((hw >> 13) & 7) + 8 * xcc   # (SH bit folded into the CU number; SE x die)
        solo = (w[:, 7] & 2) != 0
        if solo.any():   # the last wave of every workgroup: how many of them share a SIMD with another one?
            key = (se[solo] * 32 + cu[solo]) * 4 + simd[solo]
            uniq, cnt = np.unique(key, return_counts=True)
            cus = len(np.unique(key // 4))
            print(f"   last waves of their workgroups: {solo.sum()} on {cus} CUs, {len(uniq)} SIMDs; waves per occupied SIMD: "
                  f"{dict(zip(*np.unique(cnt, return_counts=True)))}")
            blk = np.arange(len(w))[solo] // 4     # workgroup index (rows are blockIdx * 4 + wave)
            wave = np.arange(len(w))[solo] % 4
            print("   which wave of the workgroup is the last one:", np.bincount(wave, minlength=4), " its SIMD:", np.bincount(simd[solo], minlength=4))
            # do workgroups b, b + 256, ... share a CU?
            cukey = se * 32 + cu
            first = cukey[0::4] if len(cukey) % 4 == 0 else None
            if first is not None and len(first) >= 1024:
                same = sum(len(set(first[b::256])) == 1 for b in range(256))
                print(f"   workgroups b, b+256, b+512, b+768 on ONE CU for {same} of 256 b; distinct CUs seen: {len(set(first))}")
                for b in (0, 1, 2, 3, 257):
                    print(f"      workgroup {b}: CU key {first[b]}, SIMDs of its waves {simd[b * 4:b * 4 + 4].tolist()}")
        late = us(w[:, 4]) > 0.5 * us(w[:, 4]).max()
        key = (se[late] * 16 + cu[late]) * 4 + simd[late]
        uniq, cnt = np.unique(key, return_counts=True)
        print(f"   waves ending in the last half of the launch: {late.sum()} on {len(uniq)} distinct SIMDs (max {cnt.max() if len(cnt) else 0} on one), "
              f"{len(np.unique(key // 4))} distinct CUs; SIMD histogram of ALL waves {np.bincount(simd, minlength=4)}, of the late ones {np.bincount(simd[late], minlength=4)}")
        tail = (w[:, 5] >> 32) > 0   # waves that ran the tail function: word 1 = when it was entered, 5 = its trips << 32 | the loop's
        if tail.any():
            trips, enter, leave = (w[tail, 5] >> 32), us(w[tail, 1]), us(w[tail, 4])
            per = (leave - enter) / np.maximum(trips, 1)
            order = np.argsort(leave)[::-1][:8]
            print(f"   tail function: {tail.sum()} waves; entered us [min p5 p50 p95 max] {' '.join(f'{v:7.1f}' for v in np.percentile(enter, [0, 5, 50, 95, 100]))}; "
                  f"trips [p50 p95 max] {np.percentile(trips, 50):.0f} {np.percentile(trips, 95):.0f} {trips.max()}; us per trip [p5 p50 p95] "
                  f"{' '.join(f'{v:5.2f}' for v in np.percentile(per[trips >= 8], [5, 50, 95]))}")
            print("   the last to return: " + "; ".join(f"in {enter[k]:.0f} out {leave[k]:.0f} trips {trips[k]} ({per[k]:.2f} us each), loop trips {w[tail, 5][k] & 0xffffffff}" for k in order))
        w = w.copy(); w[:, 5] &= 0xffffffff
        end_us = us(w[:, 4])
        for q in (50, 90, 99, 99.9, 100):
            print(f"   end percentile {q}: {np.percentile(end_us, q):.1f} us")
w = np.concatenate(allw)
start, end, iters, cyc = w[:, 0], w[:, 4], w[:, 5] & 0xffffffff, w[:, 6]
life_us = (end - start) / 100.0
span_us = (end.max() - start.min()) / 100.0
clock_mhz = cyc.sum() / life_us.sum()
print(f"recorded launches {len(allw)}, waves {len(w)}, span {span_us:.1f} us")
print(f"measured shader clock (s_memtime cycles per s_memrealtime microsecond, lifetime-weighted): {clock_mhz:.0f} MHz")
print(f"wave-slot occupancy: sum of wave lifetimes / (span x 1024 SIMDs x 4 slots) = {life_us.sum() / (span_us * 4096):.3f}")
print(f"wave-iterations {iters.sum()}  -> {iters.sum() / span_us:.1f} per us; mean {life_us.sum() / iters.sum():.2f} us per iteration per wave")
