#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a markdown table (per-kernel calls /
total / avg / min / max in microseconds + launch geometry) — the `--stats` summary for
builds of rocprofv3 whose default output is the SQLite rocpd format."""
import collections
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute(
        "select name, end-start, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels"))
    d = collections.defaultdict(list)
    meta = {}
    for r in rows:
        d[r[0]].append(r[1])
        meta[r[0]] = r[2:]
    tot = sum(sum(v) for v in d.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % | grid | wg | lds B | vgpr | agpr | sgpr | scratch |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        g = meta[n]
        lines.append(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.1f} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | "
                     f"{max(v)/1e3:.2f} | {100*sum(v)/tot:.1f} | {g[0]} | {g[1]} | {g[2]} | {g[3]} | {g[4]} | {g[5]} | {g[6]} |")
    text = "\n".join(lines) + "\n"
    if out:
        with open(out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
