#!/usr/bin/env python3
"""Timeline of the last forward pass in a rocprofv3 kernel trace: every launch with its start offset, duration and
the idle gap before it, then totals (busy time, gaps, wall time).

    python tools/trace_timeline.py <kernel_trace.csv> [first kernel of a pass: stage_input]"""
import csv
import sys


def short(name):
    n = name.split("(")[0]
    if "<" in n:
        head, args = n.split("<", 1)
        n = head.split("::")[-1] + "<" + args.split(">")[0].replace(" ", "") + ">"
    else:
        n = n.split("::")[-1]
    return n[:64]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    first = sys.argv[2] if len(sys.argv) > 2 else "stage_input"
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    if not starts:
        sys.exit("no launch of %s in the trace" % first)
    begin = starts[-1]
    # the pass ends with the detector tail; everything after the last launch of the pass's final kernel is dropped
    seq = rows[begin:]
    t0 = int(seq[0]["Start_Timestamp"])
    prev_end = t0
    busy = gaps = 0.0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3
        dur = (e - s) / 1e3
        busy += dur
        gaps += max(gap, 0.0)
        print("%9.1f us  +%6.1f gap  %8.1f us  grid %-8s wg %-4s %s" % ((s - t0) / 1e3, gap, dur, r.get("Grid_Size", "?"),
                                                                    r.get("Workgroup_Size", "?"), short(r["Kernel_Name"])))
        prev_end = max(prev_end, e)
    print("launches %d, busy %.1f us, gaps %.1f us, wall %.1f us" % (len(seq), busy, gaps, (prev_end - t0) / 1e3))


if __name__ == "__main__":
    main()
