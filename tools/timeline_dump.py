#!/usr/bin/env python
"""One steady-state training step of a rocprofv3 kernel trace (rocpd sqlite) as a table: start / end relative to the step's first
kernel, duration, queue, kernel name.  Usage: timeline_dump.py <db> [anchor-kernel-substring] [steps-back]"""
import sqlite3
import sys


def main(db, anchor="k_adam", back="3"):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    b = int(back)
    lo, hi = idx[-b - 1], idx[-b]
    seg = rows[lo + 1:hi + 1]
    t0 = seg[0][1]
    print(f"step span {(seg[-1][2] - t0) / 1e3:.1f} us, {len(seg)} kernels")
    for name, s, e, qu in seg:
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{qu}  {name[:90]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
