#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max / %.

    python tools/rocpd_stats.py gpurun_out/prof1/r01_results.db [--skip-first N] > profiles/xxx_kernel_stats.txt
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    q = f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    rows = list(db.execute(q))
    stats = {}
    for name, st, en in rows:
        short = name.split("(")[0].replace("gatsspg::", "")
        a = stats.setdefault(short, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(v[1] for v in stats.values())
    print(f"# {path}: {len(rows)} dispatches, {total / 1e6:.3f} ms of kernel time; columns: calls total_us avg_us min_us max_us pct")
    print(f"# columns cols={cols[:6]}...")
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:70]:70s} {v[0]:7d} {v[1] / 1e3:12.1f} {v[1] / v[0] / 1e3:10.2f} {v[2] / 1e3:10.2f} {v[3] / 1e3:10.2f} {100.0 * v[1] / total:6.2f}")


if __name__ == "__main__":
    main()
