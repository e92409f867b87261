#!/usr/bin/env python3
"""Summarise the final burst of a rocprofv3 kernel trace (CSV): everything after
the last idle gap >= --gap-ms is aggregated per kernel name and divided by the
number of replays.  Writes a small CSV (suitable for profiles/) and prints the top."""
import argparse
import csv
import re
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\s+", " ", name)
    return name[:190]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--replays", type=int, required=True)
    ap.add_argument("--gap-ms", type=float, default=200.0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--by-grid", action="store_true", help="one row per (kernel, grid size): separates the layers that share a kernel")
    ap.add_argument("--sequence", default=None, help="also write the replay as a SEQUENCE: one row per launch position "
                    "(mean duration and mean gap to the previous kernel over the replays, kernel, grid)")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        rd = csv.DictReader(f)
        for r in rd:
            name = r["Kernel_Name"]
            if a.by_grid:
                gx, gy = r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "1")
                wx = r.get("Workgroup_Size_X", "")
                name = "%s  [grid %sx%s wg %s]" % (short(name)[:130], gx, gy, wx)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] >= a.gap_ms * 1e6:
            cut = i
    burst = rows[cut:]
    span = (burst[-1][1] - burst[0][0]) / 1e6
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in burst:
        agg[short(n)][0] += 1
        agg[short(n)][1] += e - s
    busy = sum(v[1] for v in agg.values()) / 1e6
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# burst of %d kernels after the last %.0f ms idle gap; %d replays; wall %.3f ms/replay; "
                    "kernel-busy %.3f ms/replay" % (len(burst), a.gap_ms, a.replays, span / a.replays, busy / a.replays)])
        w.writerow(["kernel", "launches_per_replay", "avg_us", "us_per_replay", "pct_of_busy"])
        for name, (cnt, ns) in out:
            w.writerow([name, round(cnt / a.replays, 2), round(ns / cnt / 1e3, 2), round(ns / a.replays / 1e3, 2),
                        round(100.0 * ns / (busy * 1e6), 2)])
    if a.sequence and len(burst) % a.replays == 0:
        per = len(burst) // a.replays
        with open(a.sequence, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["# position in the replay, mean over %d replays" % a.replays])
            w.writerow(["pos", "start_us", "dur_us", "gap_before_us", "kernel"])
            for i in range(per):
                d = [burst[r * per + i][1] - burst[r * per + i][0] for r in range(a.replays)]
                g = [burst[r * per + i][0] - burst[r * per + i - 1][1] for r in range(a.replays) if r * per + i > 0]
                st = [burst[r * per + i][0] - burst[r * per][0] for r in range(a.replays)]
                w.writerow([i, round(sum(st) / len(st) / 1e3, 2), round(sum(d) / len(d) / 1e3, 2),
                            round(sum(g) / max(len(g), 1) / 1e3, 2), short(burst[i][2])])
    print("wall %.3f ms/replay, kernel-busy %.3f ms/replay, %d kernels/replay" % (span / a.replays, busy / a.replays, len(burst) // a.replays))
    for name, (cnt, ns) in out[: a.top]:
        print("%7.1f us/replay %6.1f launches avg %8.2f us  %s" % (ns / a.replays / 1e3, cnt / a.replays, ns / cnt / 1e3, name[:110]))


if __name__ == "__main__":
    main()
