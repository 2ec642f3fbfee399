#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each in its own run: MI355X_MICROARCH.md, HBM section)
summarised by tools/rocpd_pmc.py:

    python tools/make_pmc_traffic.py fetch.txt write.txt [--git HEAD_SHA] > profiles/pmc_traffic.json

Keeps the entries of the existing file that the passes do not cover (the extractor's).  Records the hash of the sources the passes were taken
on (onepose_amd.build_ext.source_hash): bench.py compares it with the sources it runs and reports a stale file as stale."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onepose_amd import build_ext  # noqa: E402

KEYS = {"mlp0_kernel": "mlp0", "qkv_kv_kernel": "qkv_kv", "mlp3_kernel": "mlp3", "gats_leaf8x4_kernel": "gats", "score_exp_kernel": "score_exp",
        "conf_finalize_kernel": "conf_finalize", "final_proj_norm_kernel": "final_proj_norm", "match_tail_kernel": "match_tail",
        "kv_final_kernel": "kv_final", "stat_final_kernel": "stat_final", "mlp0_sp_kernel": "mlp0_sp", "qkv_kv_sp_kernel": "qkv_kv_sp",
        "mlp3_sp_kernel": "mlp3_sp", "score_exp_sp_kernel": "score_exp_sp"}


def parse(path, counter):
    out = {}
    with open(path) as f:
        lines = [l.split() for l in f if l.strip()]
    hdr = next(l for l in lines if l[0] == "#")
    col = hdr.index(counter) - 1   # "# kernel calls avg_us C1 C2": the data lines have no "#"
    for l in lines:
        if l[0] == "#":
            continue
        for kname, key in KEYS.items():     # names are mangled and truncated (_ZN7gatsspg11mlp0_kernelINS_8GemmTile...): match the kernel's own name
            if (f"{len(kname)}{kname}" in l[0] or l[0].split("<")[0] == kname) and key not in out:
                if kname == "gats_leaf8x4_kernel" and "kernelILb0ELb0" not in l[0] and "<" not in l[0]:
                    continue                 # the plain layer kernel, not the first layer's fused state load
                out[key] = float(l[col])
    return out


def main():
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    git = sys.argv[sys.argv.index("--git") + 1] if "--git" in sys.argv else None
    path = sys.argv[sys.argv.index("--file") + 1] if "--file" in sys.argv else os.path.join(ROOT, "profiles", "pmc_traffic.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    if "--config" in sys.argv:
        # a pass taken on another BASELINE config (python bench.py --config NAME): filed under _configs[NAME], the rest of the file is kept
        name = sys.argv[sys.argv.index("--config") + 1]
        entry = {"_source": {"csrc_sha": build_ext.source_hash(), "git_head": git, "passes": [os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2])],
                             "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --config {name} --steps 4 --warmup 2 --reps 1 "
                                        "--min-timed-seconds 0 --streams 1 (tools/r06_collect.sh)"}}
        for k in sorted(set(fetch) & set(write)):
            entry[k] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k], "bytes": int(round((2 * fetch[k] + write[k]) * 1024))}
        old.setdefault("_configs", {})[name] = entry
        json.dump(old, sys.stdout, indent=1)
        return
    out = {"_comment": "HBM-side traffic per launch from rocprofv3 PMC passes: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, values in KB per "
                       "dispatch; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, so fetch bytes = 2 * FETCH_SIZE * 1024 "
                       "(MI355X_MICROARCH.md, HBM section); write bytes = WRITE_SIZE * 1024.  bench.py quotes these as static numbers and checks "
                       "_source.csrc_sha against the sources it runs (roofline.traffic_source).",
           "_source": {"csrc_sha": build_ext.source_hash(), "git_head": git, "passes": [os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2])],
                       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 6 --warmup 2 --reps 1 --min-timed-seconds 0 --streams 1 (tools/r06_collect.sh)"}}
    for k in sorted(set(fetch) & set(write)):
        out[k] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k], "bytes": int(round((2 * fetch[k] + write[k]) * 1024))}
    for k, v in old.items():
        if k not in out and not k.startswith("_"):
            out[k] = dict(v, carried_over_from="the previous file (not covered by these passes)")
    if "_configs" in old:
        out["_configs"] = old["_configs"]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
