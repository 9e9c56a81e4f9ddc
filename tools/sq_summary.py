#!/usr/bin/env python3
"""Per-kernel reading of one rocprofv3 SQ-counter pass over a short benchmark run.

    python tools/sq_summary.py <counter_collection.csv> [top_n] > profiles/<tag>_sq_counters_f32.md

The pass (counters only, no tracing in the same run):
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY
              SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --frames 16
              --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-split-record

Derived columns, per kernel family summed over its launches:
  matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
  clock            = GRBM_GUI_ACTIVE / 8 / duration (End - Start timestamps of the dispatches)
  wave wait        = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (both in quad-cycles)
  waves per CU     = 4 x SQ_WAVE_CYCLES / (cycles per XCD x 256 CUs): waves resident on a CU, averaged over the launch
"""
import collections
import csv
import re
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("fd::", "")
    return re.sub(r"\s+", "", n)


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r["Dispatch_Id"]
        launches[k].add(did)
        if did not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            seen.add(did)
            dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    rows = []
    for k, v in vals.items():
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        cyc = gui / 8.0                                   # cycles per XCD, summed over the launches
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        wave = v.get("SQ_WAVE_CYCLES", 0.0)
        rows.append(dict(kernel=k, launches=len(launches[k]), cycles=cyc,
                         busy=mf / (1024.0 * cyc),
                         ghz=(cyc / dur[k] * 1e-9) if dur[k] > 0 else float("nan"),
                         wait=(v.get("SQ_WAIT_INST_ANY", 0.0) / wave) if wave else float("nan"),
                         wait_lds=(v.get("SQ_WAIT_INST_LDS", 0.0) / wave) if wave else float("nan"),
                         occ=4.0 * wave / (cyc * 256.0),
                         mfma=v.get("SQ_INSTS_MFMA", 0.0),
                         valu_per_mfma=((v.get("SQ_INSTS_VALU", 0.0) - v.get("SQ_INSTS_MFMA", 0.0)) / v["SQ_INSTS_MFMA"])
                         if v.get("SQ_INSTS_MFMA") else float("nan"),
                         lds_per_mfma=(v.get("SQ_INSTS_LDS", 0.0) / v["SQ_INSTS_MFMA"]) if v.get("SQ_INSTS_MFMA") else float("nan"),
                         ms=dur[k] * 1e3))
    rows.sort(key=lambda r: -r["cycles"])
    total = sum(r["cycles"] for r in rows)
    print("| kernel | launches | share of GPU cycles | ms under the counters | matrix pipe busy | clock GHz | waves resident per CU | "
          "waves waiting | of which on LDS | other VALU per MFMA | LDS instr. per MFMA |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        print("| `%s` | %d | %.1f %% | %.2f | %.0f %% | %.2f | %.1f | %.0f %% | %.0f %% | %.2f | %.2f |" % (
            r["kernel"], r["launches"], 100.0 * r["cycles"] / total, r["ms"], 100.0 * r["busy"], r["ghz"], r["occ"],
            100.0 * r["wait"], 100.0 * r["wait_lds"], r["valu_per_mfma"], r["lds_per_mfma"]))
    busy_all = sum(r["busy"] * r["cycles"] for r in rows) / total
    print()
    print("All kernels of the pass: matrix pipes busy %.0f %% of the GPU-active cycles." % (100.0 * busy_all))


if __name__ == "__main__":
    main()
