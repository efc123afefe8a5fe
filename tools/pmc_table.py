#!/usr/bin/env python
"""SQ counters per launch shape from a rocprofv3 --pmc run (rocpd sqlite).  Usage: pmc_table.py <db> [name-substring ...]"""
import collections
import sqlite3
import sys


def main(db, *subs):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    q = f"select kernel_name, {gx or 0}, dispatch_id, counter_name, value from counters_collection"
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for kn, g, did, cn, v in cur.execute(q):
        if subs and not any(s in kn for s in subs):
            continue
        acc[(kn.split("(")[0].replace("void ", "")[:48], g, did)][cn] += v
    groups = collections.defaultdict(list)
    for (kn, g, did), c in acc.items():
        groups[(kn, g)].append(c)
    names = sorted({n for cs in groups.values() for c in cs for n in c})
    print("| kernel | grid | n | " + " | ".join(names) + " |")
    print("|---|---|---|" + "---|" * len(names))
    for (kn, g), cs in sorted(groups.items()):
        print(f"| `{kn}` | {g} | {len(cs)} | " + " | ".join(f"{sum(c.get(n, 0.0) for c in cs) / len(cs):.4g}" for n in names) + " |")


if __name__ == "__main__":
    main(*sys.argv[1:])
