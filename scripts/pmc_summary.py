#!/usr/bin/env python
"""Average the per-dispatch counter values of rocprofv3 --pmc passes for kernels whose name
contains a pattern; prints a markdown table."""
import collections
import csv
import glob
import sys


def main(root, pattern):
    print("| pass | counter | mean per dispatch | dispatches |")
    print("|---|---|---:|---:|")
    for p in sorted(glob.glob(f"{root}/pass*/pmc_counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if pattern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"| {p.split('/')[-2]} | {k} | {sum(v)/len(v):.6g} | {len(v)} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
