#!/usr/bin/env python
"""Per-kernel PMC summary from a rocprofv3 rocpd SQLite file (counter values summed over all
instances/dimensions of a dispatch, averaged over the dispatches of each kernel).

    python tools/rocpd_pmc.py gpurun_out/pmc1/pmc_results.db [--by-grid]

--by-grid keeps launches of one kernel with different grid sizes apart (the SuperPoint convolutions are one template
launched on four resolutions) and prints the full template arguments.
"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tables if t.startswith(p)][0]  # noqa: E731
    pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    names = dict(db.execute(f"select id, name from {ip}"))
    disp = {}
    by_grid = "--by-grid" in sys.argv
    for ev, kname, st, en, gx in db.execute(f"select d.event_id, s.{name_col}, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id"):
        base = kname.split("(")[0].replace("gatsspg::", "")
        disp[ev] = ((base[:96] + f" grid={gx}") if by_grid else base[:48], en - st)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(float)
    seen = set()
    for ev, pmc, val in db.execute(f"select event_id, pmc_id, value from {pe}"):
        if ev not in disp:
            continue
        k, d = disp[ev]
        per[k][names[pmc]] += val
        if ev not in seen:
            seen.add(ev)
            cnt[k] += 1
            dur[k] += d
    ctrs = sorted({c for v in per.values() for c in v})
    print("# kernel calls avg_us " + " ".join(ctrs))
    for k in sorted(per, key=lambda k: -dur[k]):
        n = cnt[k]
        print(f"{k:{110 if by_grid else 50}s} {n:5d} {dur[k] / n / 1e3:9.2f} " + " ".join(f"{per[k][c] / n:14.1f}" for c in ctrs))


if __name__ == "__main__":
    main()
