#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs of the same bench command).

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [frames]

With `frames` (the number of frames the profiled command pushed through the detector, warm-up included) the file also
holds the traffic of the whole run per frame: every kernel of the library except the weight packing of detector set-up.

Units and corrections (MI355X guide, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide (16 B per lane) coalesced read stream, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Values are averages per launch."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    out = {}
    for k in f:
        if "fd::" not in k:
            continue
        n, fv = f[k]
        wn, wv = w.get(k, [0, 0.0])
        fetch = 2.0 * 1024.0 * fv / n
        write = 1024.0 * wv / max(wn, 1)
        short = k.split("(")[0].replace("void ", "")
        out[short] = {"launches": n, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                      "hbm_bytes_per_launch": round(fetch + write)}
    from facedetection_amd import _build
    whole = None
    if len(sys.argv) > 4:
        frames = int(sys.argv[4])
        total = sum(v["launches"] * v["hbm_bytes_per_launch"] for k, v in out.items() if not k.startswith("fd::pack_"))
        whole = {"frames": frames, "hbm_bytes": int(total), "hbm_bytes_per_frame": int(round(total / frames)),
                 "note": "sum over all kernels of the run (launches x bytes per launch) except the weight packing of "
                         "detector set-up, divided by the frames that went through the detector (warm-up included)"}
    json.dump({"csrc_sha16": _build.source_id(),   # the kernel set these counters belong to (fd_build_id of the library)
               "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes, averaged per launch; "
                       "command: bench.py --frames 16 --steps 1 --warmup 1 " + ("--no-split-record (micro-batch 16, 1080p; the float32 pass alone)" if whole else
                                                                                "--split-streams 1 (micro-batch 16, 1080p; float32 pass, then f16x3 pass)"),
               "whole_run": whole,
               "kernels": out}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    if whole:
        print("whole run: %.1f GB over %d frames = %.2f GB per frame" % (whole["hbm_bytes"] / 1e9, whole["frames"], whole["hbm_bytes_per_frame"] / 1e9))
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print("%-50s %4d launches  %8.1f MB/launch" % (k[:50], v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
