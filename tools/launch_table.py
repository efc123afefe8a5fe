#!/usr/bin/env python
"""Per-launch-shape durations from a rocprofv3 kernel trace (rocpd sqlite): launches of one kernel grouped by grid size
(one group per layer / launch shape).  Usage: launch_table.py <db> [name-substring ...]"""
import sqlite3
import sys


def main(db, *subs):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                       "max(lds_size), max(vgpr_count) from kernels group by name, grid_x, grid_y order by name, grid_x").fetchall()
    print("| kernel | grid (threads) | wg | launches | avg_us | min_us | max_us | lds | vgpr |\n|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        if subs and not any(s in r[0] for s in subs):
            continue
        print(f"| `{r[0][:70]}` | {r[1]}x{r[2]} | {r[3]} | {r[4]} | {r[5]:.1f} | {r[6]:.1f} | {r[7]:.1f} | {r[8]} | {r[9]} |")


if __name__ == "__main__":
    main(*sys.argv[1:])
