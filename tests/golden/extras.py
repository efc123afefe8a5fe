"""The WIDER set of golden quantities: beyond one ELBO + gradient per case, the fixtures also pin
    x.predy_mean / x.predy_var   predict_y at the case's X with the case's z draws (dgp.py:117-119)
    x.preddens                   predict_density at (X, Y) (dgp.py:121-126)
    x.fc_Fmean / x.fc_Fvar       last layer of predict_all_layers_full_cov (dgp.py:113-114, layers.py:66-69 / utils.py:33-45)
    x.adam3_elbo / x.adam3_q_mu  ELBO and last-layer q_mu after three tf.train.AdamOptimizer(0.01) steps on -ELBO at FIXED z draws
    x.ng_q_mu / x.ng_q_sqrt      last layer after ONE gpflow NatGradOptimizer(gamma = 0.1) step (Gaussian-likelihood cases;
                                 demos/demo_regression_UCI.ipynb:360-366)
`oracle_extras` computes them with the CPU oracle (make_golden.py stores them, tests/test_golden.py compares oracle and HIP path with
them); make_golden_from_reference.py computes the same keys with the reference itself where gpflow 1.1.1 / TF 1.8 exist."""
import numpy as np

from oracle import dgp_oracle as O
from oracle import model as OM

ADAM_STEPS, ADAM_LR, NG_GAMMA = 3, 0.01, 0.1


def _is_gaussian(c):
    return not c.get("classes") and not c.get("bernoulli")


def oracle_extras(spec, state, X, Y, zs, c):
    S, L = c["S"], c["L"]
    out = {}
    om = OM.build(O.NP, spec, state, S, c["num_data"])
    _, Fm, Fv = om.propagate(O.NP, np.asarray(X, float), zs, full_cov=False, S=S)
    pm, pv = om.likelihood.predict_mean_and_var(O.NP, Fm[-1], Fv[-1])
    out["x.predy_mean"], out["x.predy_var"] = np.asarray(pm), np.asarray(pv)
    out["x.preddens"] = np.asarray(om.predict_density(O.NP, np.asarray(X, float), np.asarray(Y, float), zs, S))
    _, Fmc, Fvc = om.propagate(O.NP, np.asarray(X, float), zs, full_cov=True, S=S)
    out["x.fc_Fmean"], out["x.fc_Fvar"] = np.asarray(Fmc[-1]), np.asarray(Fvc[-1])
    # three Adam steps on loss = -ELBO, the same z draws every step
    st = {k: np.array(v, dtype=np.float64, copy=True) for k, v in state.items()}
    m1 = {k: np.zeros_like(v) for k, v in st.items()}
    m2 = {k: np.zeros_like(v) for k, v in st.items()}
    for t in range(1, ADAM_STEPS + 1):
        _, g = OM.elbo_and_grad(spec, st, X, Y, zs, S, num_data=c["num_data"])
        for k in g:
            O.adam_step(st[k], -np.asarray(g[k]), m1[k], m2[k], t, lr=ADAM_LR)
    out["x.adam3_elbo"] = np.array(OM.elbo(spec, st, X, Y, zs, S, num_data=c["num_data"]))
    out["x.adam3_q_mu"] = st[f"l{L - 1}.q_mu"].copy()
    if _is_gaussian(c):
        _, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=c["num_data"])
        k = f"l{L - 1}"
        mu, sq = O.natgrad_step(np.asarray(state[k + ".q_mu"]), np.asarray(state[k + ".q_sqrt"]), -np.asarray(g[k + ".q_mu"]),
                                -np.asarray(g[k + ".q_sqrt"]), NG_GAMMA)
        out["x.ng_q_mu"] = mu
        out.update(pack_q_sqrt("x.ng_q_sqrt", sq))
    return out


def pack_q_sqrt(key, sq):
    """small factors whole, large ones (M >= 128) as Frobenius norm + leading 16 x 16 block of every output (as the q_sqrt gradients)"""
    sq = np.asarray(sq)
    if sq.size <= 4096:
        return {key: sq}
    return {key + "_norm": np.array(np.linalg.norm(sq)), key + "_block": sq[:, :16, :16].copy()}
