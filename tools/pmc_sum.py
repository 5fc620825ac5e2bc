#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_sum.py <csv> [<csv> ...]"""
import csv, collections, sys
for path in sys.argv[1:]:
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    for k in sorted(tot):
        print(path.split("/")[-1], k, "launches", len(calls[k]), {c: v for c, v in tot[k].items()})
