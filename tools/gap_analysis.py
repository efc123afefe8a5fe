#!/usr/bin/env python
"""Timeline view of a rocprofv3 kernel trace (rocpd sqlite): GPU busy time vs wall span over the steady-state steps, and the
largest idle gaps grouped by (previous kernel -> next kernel).  Usage: gap_analysis.py <db> [anchor-kernel-substring]"""
import collections
import sqlite3
import sys


def main(db, anchor="k_adam"):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    # steady state = between the 10th-last and the last anchor kernel
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 12:
        print("not enough steps")
        return
    lo, hi = idx[-11], idx[-1]
    seg = rows[lo + 1:hi + 1]
    steps = 10
    span = (seg[-1][2] - seg[0][1]) / 1e3
    busy, cur_end = 0.0, seg[0][1]
    gaps = collections.defaultdict(lambda: [0.0, 0])
    prev_name = None
    for name, s, e in seg:
        if s > cur_end:
            if prev_name is not None:
                g = gaps[(prev_name[:40], name[:40])]
                g[0] += (s - cur_end) / 1e3
                g[1] += 1
            busy += (e - s) / 1e3
            cur_end = e
            prev_name = name
        else:
            if e > cur_end:
                busy += (e - cur_end) / 1e3
                cur_end = e
                prev_name = name
    print(f"steps {steps}: span {span / steps:.1f} us/step, busy {busy / steps:.1f} us/step, idle {(span - busy) / steps:.1f} us/step, "
          f"kernels/step {len(seg) / steps:.1f}")
    print("largest idle gaps (us/step, count/step):")
    for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {t / steps:7.2f} {c / steps:5.1f}  {a}  ->  {b}")
    per = collections.defaultdict(float)
    for name, s, e in seg:
        per[name[:60]] += (e - s) / 1e3
    print("kernel time per step (us):")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:30]:
        print(f"  {v / steps:8.2f}  {k}")


if __name__ == "__main__":
    main(*sys.argv[1:])
