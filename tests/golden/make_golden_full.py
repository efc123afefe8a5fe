"""Generates tests/golden/golden_full_<case>.npz: the CPU oracle at the FULL sizes of BASELINE.json configs[1..4] (minutes per
case on 8 cores — run once in the build container, never at test time and never on the GPU box).

    python -m tests.golden.make_golden_full [case ...]

Stored per case (tests/golden/full_cases.py `summarise`): ELBO, data term, per-layer KL, per-layer Fmean / Fvar summaries, every
gradient block of -ELBO's negative (the oracle's d ELBO / d theta) as norm + projection + corners, and for config 5 the last
layer after one NatGradOptimizer(0.1) step.  source = "oracle" — these pin the HIP path to the oracle at the sizes where the
multi-round chains, the d-split hand-over and the multi-round split-K engage; they do not pin the oracle to the reference."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import dgp_oracle as O  # noqa: E402
from oracle import model as OM  # noqa: E402
from tests.golden import full_cases as FC  # noqa: E402


def oracle_case(name):
    """(spec, state, X, Y, zs, c) without the device model (the generator runs where no GPU exists)"""
    c, Xall, Yall, Z, specs, zs = FC.inputs(name)
    rng = np.random.RandomState(c["seed"])
    lds = O.init_layers_linear(Xall, Yall, Z, specs, white=False, jitter=1e-6, num_outputs=c.get("classes"))
    for l in lds:                                   # tests/helpers.make_case(randomize=True), same draws in the same order
        l["q_mu"] = 0.3 * rng.randn(*l["q_mu"].shape)
        D, M = l["q_sqrt"].shape[0], l["q_sqrt"].shape[1]
        l["q_sqrt"] = l["q_sqrt"] * 0.7 + 0.05 * np.tril(rng.randn(D, M, M))
    likname = "multiclass" if c.get("classes") else "gaussian"
    sl, state = OM.state_from_layers(lds, lik_variance=c["lik"] or 1.0, likelihood=likname)
    spec = dict(jitter=1e-6, white=False, likelihood=likname, layers=sl, num_classes=c.get("classes"))
    return spec, state, Xall[:c["N"]], Yall[:c["N"]], zs, c


def outputs(name):
    spec, state, X, Y, zs, c = oracle_case(name)
    S, L = c["S"], c["L"]
    t0 = time.time()
    _, Fm, Fv = OM.propagate(spec, state, X, zs, S)
    om = OM.build(O.NP, spec, state, S, c["num_data"])
    kls = np.array([float(l.KL(O.NP)) for l in om.layers])
    elbo, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=c["num_data"])
    out = dict(elbo=np.array(elbo), kls=kls, source=np.array("oracle"))
    for l in range(L):
        out.update(FC.summarise(f"Fmean{l}", Fm[l]))
        out.update(FC.summarise(f"Fvar{l}", Fv[l]))
    for k, v in g.items():
        out.update(FC.summarise("grad." + k, v))
    if c.get("natgrad"):
        k = f"l{L - 1}"
        mu, sq = O.natgrad_step(np.asarray(state[k + ".q_mu"]), np.asarray(state[k + ".q_sqrt"]), -np.asarray(g[k + ".q_mu"]),
                                -np.asarray(g[k + ".q_sqrt"]), c["natgrad"])
        out.update(FC.summarise("ng.q_mu", mu))
        out.update(FC.summarise("ng.q_sqrt", sq))
    out["oracle_seconds"] = np.array(time.time() - t0)
    return out


def main():
    names = sys.argv[1:] or list(FC.FULL)
    for name in names:
        out = outputs(name)
        path = os.path.join(HERE, f"golden_full_{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "elbo", float(out["elbo"]), f"{float(out['oracle_seconds']):.0f} s", os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
