#!/usr/bin/env python3
"""Prints the busiest kernels of a rocprofv3 --kernel-trace --stats run: python profiles/print_kernel_stats.py <dir> [n]"""
import csv
import glob
import sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for p in glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:n]:
        print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), f"{float(r['AverageNs']) / 1e3:10.2f} us", r["Percentage"].rjust(7))
