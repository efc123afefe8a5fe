#!/usr/bin/env python
"""N training steps of the config-2 model on a minibatch of <rows> rows (the per-rank shard of a strong-scaling run), for a kernel
timeline under rocprofv3:  rocprofv3 --kernel-trace -d /tmp/tl -o t -- python tools/shard_timeline.py 125"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    sys.path.insert(0, p)
import bench as B  # noqa: E402


def main(rows, steps=40):
    import torch
    model, X, Y, Z = B.build_model(dict(B.CFG), 0, 1, rows)
    for _ in range(steps):
        model.train_step(0.01)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 40)
