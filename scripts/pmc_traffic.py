#!/usr/bin/env python
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of rocprofv3 over bench.py.

    python scripts/pmc_traffic.py <dir with pass*/..pmc_counter_collection.csv> <bench json of the same command> [--append]

profiles/pmc_traffic.json is a LIST of records, the headline shape first; --append adds the record (or replaces the one of the
same shape) instead of starting the list over.

HBM bytes per launch of the dominant kernel = (FETCH_SIZE x 2 [gfx950 tallies a 128-byte request as 64 bytes,
MI355X_MICROARCH.md "HBM"] + WRITE_SIZE) x 1024 [the counters are in KiB], averaged over the dispatches of the kernel
named in the bench line.  The record carries the launch geometry and a digest of the kernel sources; bench.py refuses
it (traffic = null) when either differs from the run it is reporting."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_digest  # noqa: E402


def main(pmc_dir, bench_json, append=False):
    rec = json.load(open(bench_json))
    kernel = rec["roofline"]["kernel"]
    base = kernel.split("<")[0]
    vals = collections.defaultdict(list)
    for p in sorted(glob.glob(os.path.join(pmc_dir, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(p)):
            if base in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not vals["FETCH_SIZE"] or not vals["WRITE_SIZE"]:
        raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE rows for {base} under {pmc_dir}")
    fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
    write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
    launch = rec["config"]["launch"]
    out = {
        "hbm_bytes_per_launch": int(round((2.0 * fetch + write) * 1024)),
        "fetch_size_kb": fetch, "write_size_kb": write,
        "dispatches": [len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"])],
        "rows": rec["roofline"]["rows_per_launch"], "K": rec["config"]["K"], "kernel": kernel,
        "workgroups": launch["workgroups"], "threads": launch["threads"], "chunks_per_wave": launch["chunks_per_wave"],
        "source_sha256": kernel_source_digest(kernel),
        "algorithmic_bytes_per_launch": rec["roofline"]["algorithmic_bytes_per_launch"],
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py, kernel {kernel}: "
                  "(FETCH_SIZE x 2 [gfx950 half-count correction] + WRITE_SIZE) x 1024",
    }
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    recs = []
    if append and os.path.exists(path):
        recs = json.load(open(path))
        if isinstance(recs, dict):
            recs = [recs]
        recs = [r for r in recs if not (r.get("rows") == out["rows"] and r.get("K") == out["K"])]
    recs.append(out)
    json.dump(recs, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], "--append" in sys.argv[3:])
