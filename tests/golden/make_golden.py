"""Generates tests/golden/golden_<case>.npz from the CPU oracle (oracle/).  The reference itself cannot be imported here
(gpflow 1.1.1 / TF 1.8 unavailable), so these vectors pin the ORACLE + the HIP path against each other and against
regressions; the oracle in turn is pinned relationally by tests/test_oracle_identities.py.

    python -m tests.golden.make_golden
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import dgp_oracle as O  # noqa: E402
from oracle import model as OM  # noqa: E402
from tests.golden import cases, extras  # noqa: E402


def outputs(name):
    spec, state, _, X, Y, zs, c = cases.build(name)
    S = c["S"]
    Fs, Fm, Fv = OM.propagate(spec, state, X, zs, S)
    om = OM.build(O.NP, spec, state, S, c["num_data"])
    kls = np.array([float(l.KL(O.NP)) for l in om.layers])
    elbo, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=c["num_data"])
    out = dict(elbo=np.array(elbo), kls=kls, source=np.array("oracle"))     # NOT the reference: see DESIGN.md section 3
    for l in range(c["L"]):
        out[f"Fmean{l}"], out[f"Fvar{l}"], out[f"F{l}"] = Fm[l], Fv[l], Fs[l]
    for k, v in g.items():
        if v.size <= 4096:
            out["grad." + k] = v
        else:                     # large q_sqrt gradients: Frobenius norm + leading 16x16 block of every output
            out["gradnorm." + k] = np.array(np.linalg.norm(v))
            out["gradblock." + k] = v[:, :16, :16].copy()
    out.update(extras.oracle_extras(spec, state, X, Y, zs, c))      # predict_y / density, full_cov, 3 Adam steps, natgrad step
    return out


def main():
    for name in cases.CASES:
        out = outputs(name)
        path = os.path.join(HERE, f"golden_{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "elbo", float(out["elbo"]), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
