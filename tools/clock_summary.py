#!/usr/bin/env python3
"""Aggregate the phase-clock lines of an RT_PHASE_CLOCK build (tools/gpu_clock.sh): python tools/clock_summary.py gpurun_out/<tag>/out.txt
WPC lines (render_kernel): per phase, share of wave time and mean active lanes.  PC lines (render_kernel_sorted): shares per stage."""
import re
import sys
from collections import defaultdict

txt = open(sys.argv[1], errors="replace").read()
wpc = [l for l in txt.splitlines() if l.startswith("WPC")]
if wpc:
    tot = 0
    cyc, lan = defaultdict(int), defaultdict(int)
    for l in wpc:
        t = l.split()
        tot += int(t[2])
        i = 3
        while i + 2 < len(t):  # (a trailing `wall10ns <ticks>` pair is the wave's run time, not a phase)
            name = t[i]
            cyc[name] += int(t[i + 1]); lan[name] += int(t[i + 2])
            i += 3
    print(f"{len(wpc)} waves, total cycles {tot}")
    for k in cyc:
        if cyc[k]:
            print(f"  {k:12s} {100.0 * cyc[k] / tot:6.2f} % of wave time, {lan[k] / cyc[k]:5.1f} lanes")
pc = [l for l in txt.splitlines() if l.startswith("PC ")]
if pc:
    names = ["shade", "cast", "bar1", "write", "bar2", "read", "top"]
    s = defaultdict(int)
    cls = defaultdict(lambda: [0, 0])
    for l in pc:
        m = dict(re.findall(r"(\w+) (\d+)(?= |$)", l))
        for n in names:
            s[n] += int(m.get(n, 0))
        for n in ("lam", "spec", "castl"):
            mm = re.search(n + r" (\d+) (\d+)", l)
            if mm:
                cls[n][0] += int(mm.group(1)); cls[n][1] += int(mm.group(2))
    tot = sum(s.values())
    print(f"{len(pc)} waves")
    for n in names:
        print(f"  {n:6s} {100.0 * s[n] / tot:6.2f} %")
    for n, (c, cl) in cls.items():
        if c:
            print(f"  {n:6s} {100.0 * c / tot:6.2f} % of wave time at {cl / c:5.1f} lanes")
