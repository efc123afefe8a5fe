#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats rocpd database (gpurun_out/.../*_results.db) into a per-kernel table
(the committed evidence under profiles/)."""
import sqlite3
import sys


def main(db_path, out_path, title):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels "
                       "group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write(f"# {title}\n\nsource: rocprofv3 --kernel-trace --stats (rocpd sqlite), durations in microseconds\n\n")
        f.write("| kernel | calls | total_us | avg_us | min_us | max_us | pct | vgpr | agpr | lds |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{r[0][:90]}` | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {100 * r[2] / tot:.2f} | {r[6]} | {r[7]} | {r[8]} |\n")
        f.write(f"\ntotal kernel time: {tot:.1f} us\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
