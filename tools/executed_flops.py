#!/usr/bin/env python
"""EXECUTED fp64 MFMA flops per training step of a BASELINE config shape, from a rocprofv3 PMC pass.

  run   : rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d <dir> -o p -- python tools/executed_flops.py run <cfg> <nsteps>
          (3 warm-up steps, a marker launch, <nsteps> steps, a marker launch; serial schedule via DSDGP_NO_OVERLAP=1 is not needed:
          instruction counts do not depend on the schedule)
  parse : python tools/executed_flops.py parse <out.json> <csrc-hash> cfg1=<db> cfg2=<db> ...
          sums the counter over the dispatches between the two markers: flops = MOPS x 512 (one v_mfma_f64_16x16x4_f64 = 4 MOPS
          = 2048 flops), per step and per kernel; writes profiles/r05_executed_flops.json (+ .md beside it)

The marker is dsdgp_randn with count = 2: a one-workgroup k_randn launch no model step issues."""
import collections
import ctypes as C
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def run(cfg_id, nsteps):
    import torch
    import bench_configs as BC
    from doubly_stochastic_dgp import _lib
    model, step = BC.build(cfg_id)
    ctx = model.engine().ctx
    scratch = ctx.empty(1, 16)

    def marker():
        torch.cuda.synchronize()
        _lib.check(ctx.lib.dsdgp_randn(ctx.handle, 12345, 0, 2, C.c_void_p(scratch.data_ptr())))
        torch.cuda.synchronize()
    for _ in range(3):
        step()
    marker()
    for _ in range(nsteps):
        step()
    marker()
    print(json.dumps(dict(cfg=cfg_id, steps=nsteps)))


def parse_db(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    rows = cur.execute(f"select dispatch_id, kernel_name, {gx or 0}, value from counters_collection where counter_name = "
                       "'SQ_INSTS_VALU_MFMA_MOPS_F64' order by dispatch_id").fetchall()
    disp = collections.OrderedDict()
    for did, kn, g, v in rows:
        d = disp.setdefault(did, [kn, g, 0.0])
        d[2] += v
    ids = list(disp)
    marks = [i for i, did in enumerate(ids) if disp[did][0].startswith("k_randn") and disp[did][1] <= 256]
    if len(marks) < 2:
        raise SystemExit(f"{db}: {len(marks)} marker launches found")
    lo, hi = marks[-2], marks[-1]
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    for did in ids[lo + 1:hi]:
        kn, g, v = disp[did]
        k = kn.split("(")[0].replace("void ", "")[:40]
        per_kernel[k][0] += 1
        per_kernel[k][1] += v * 512.0
    return per_kernel, hi - lo - 1


def parse(out_json, csrc_hash, *pairs):
    out = dict(csrc_sha256_16=csrc_hash, counter="SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flops", configs={})
    md = ["# round 5 — fp64 MFMA flops EXECUTED per training step (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64, x 512), by config shape", ""]
    for pr in pairs:
        key, spec = pr.split("=", 1)
        db, nsteps = spec.rsplit(":", 1)
        pk, ndisp = parse_db(db)
        n = int(nsteps)
        tot = sum(v[1] for v in pk.values())
        out["configs"][key] = dict(executed_gflop_per_step=round(tot / n / 1e9, 3), launches_per_step=round(ndisp / n, 1), steps=n,
                                   per_kernel_gflop_per_step={k: round(v[1] / n / 1e9, 3) for k, v in sorted(pk.items(), key=lambda kv: -kv[1][1]) if v[1] > 0})
        md += [f"## {key}: {tot / n / 1e9:.2f} GFLOP executed per step, {ndisp / n:.1f} launches per step ({n} steps)", "",
               "| kernel | launches / step | GFLOP / step |", "|---|---|---|"]
        md += [f"| `{k}` | {v[0] / n:.1f} | {v[1] / n / 1e9:.3f} |" for k, v in sorted(pk.items(), key=lambda kv: -kv[1][1]) if v[1] > 0]
        md.append("")
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.splitext(out_json)[0] + ".md", "w") as f:
        f.write("\n".join(md) + "\n")
    print(json.dumps({k: v["executed_gflop_per_step"] for k, v in out["configs"].items()}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]))
    else:
        parse(*sys.argv[2:])
