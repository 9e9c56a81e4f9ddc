#!/usr/bin/env python3
"""Prints average duration (us) of the kernels whose name contains one of the given substrings, from a
rocprofv3 kernel_stats.csv:  python tools/kstat.py <kernel_stats.csv> substr [substr ...]"""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print("%-70s calls %5s avg %10.1f us  min %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3))
