#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; values in KB).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads, so
traffic = 2*FETCH + WRITE (an upper bound for 8-byte-per-lane reads).  Usage: pmc_traffic.py <fetch.db> <write.db> <out-prefix> <title> [csrc-hash]"""
import collections
import json
import sqlite3
import sys


def load(db, ctr):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (ctr,)).fetchall()
    out = collections.defaultdict(list)
    for kn, v in rows:
        out[kn.split("(")[0].replace("void ", "")].append(v)
    return out


def main(fdb, wdb, prefix, title):
    f, w = load(fdb, "FETCH_SIZE"), load(wdb, "WRITE_SIZE")
    rows, js = [], {}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * sum(f.get(k, [0])) + sum(w.get(k, [0])))):
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        mb = (2 * fa + wa) * 1024 / 1e6
        rows.append(f"| `{k[:60]}` | {len(fv)} | {fa:.1f} | {max(fv):.1f} | {wa:.1f} | {max(wv):.1f} | {mb:.2f} |")
        js[k] = dict(launches=len(fv), fetch_KB_avg=fa, write_KB_avg=wa, traffic_MB_per_launch=mb,
                     traffic_MB_max_launch=(2 * max(fv) + max(wv)) * 1024 / 1e6)
    with open(prefix + ".md", "w") as fh:
        fh.write(f"# {title}\n\nseparate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (`--kernel-trace` only) of "
                 "`python bench.py --steps 4 --warmup 2 --no-cpu-baseline` (cfg 2; launch mix: 3 layers, training + the secondary "
                 "eval/predict loops of bench.py); values in KB per launch; traffic = 2*FETCH + WRITE (gfx950 correction).\n\n"
                 "| kernel | launches | FETCH avg | FETCH max | WRITE avg | WRITE max | traffic/launch (MB, avg) |\n|---|---|---|---|---|---|---|\n")
        fh.write("\n".join(rows) + "\n")
    if len(sys.argv) > 5:
        js["csrc_sha256_16"] = sys.argv[5]          # bench.csrc_hash() of the build the counters were taken from
    with open(prefix + ".json", "w") as fh:
        json.dump(js, fh, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:5])
