"""Regenerates tests/golden/golden_<case>.npz FROM THE REFERENCE ITSELF (UCL-SML/Doubly-Stochastic-DGP on gpflow==1.1.1 +
tensorflow==1.8) — the step that turns "parity unpinned" into "pinned" (VERDICT r01 item 1).

    python -m tests.golden.make_golden_from_reference [--reference /root/reference] [--check-only]

Runs only where `import gpflow, tensorflow` succeeds with those versions (a py3.6 environment with the wheels; NOT this
build container — TF 1.8 has no cp310 wheel — and never the GPU box: /root/reference does not travel).  Without them it prints
why and exits 0, leaving the oracle-generated fixtures of make_golden.py in place.

What it does per case of tests/golden/cases.py:
  * takes the case's inputs (X, Y, Z, kernels, explicit z draws) and PARAMETER VALUES (q_mu, q_sqrt, hyper-parameters) from the
    same recipe the oracle fixtures use, builds the reference `DGP` with them (dgp.py:184-192), injects the z draws through
    `propagate(..., zs=...)` (dgp.py:62,68 — the hook DGP_Quad uses) and evaluates predict_all_layers, every layer's KL(),
    compute_log_likelihood() and tf.gradients of it w.r.t. GPflow's unconstrained variables;
  * reports the largest deviation from the committed (oracle) vectors, and unless --check-only rewrites the .npz with the
    reference's numbers plus `source = "reference"`.
Only data (inputs' recipe, outputs) is stored; no reference source text.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _import_reference(ref_root):
    """(gpflow, tf, reference dgp module) or a string saying what is missing."""
    try:
        import tensorflow as tf
    except Exception as e:  # noqa: BLE001
        return f"tensorflow is not importable ({type(e).__name__}: {e})"
    try:
        import gpflow
    except Exception as e:  # noqa: BLE001
        return f"gpflow is not importable ({type(e).__name__}: {e})"
    if not str(getattr(gpflow, "__version__", "")).startswith("1.1"):
        return f"gpflow {gpflow.__version__} found, the reference needs 1.1.1 (README.md:4: >= 1.2 breaks the feature API)"
    if not os.path.isdir(os.path.join(ref_root, "doubly_stochastic_dgp")):
        return f"no reference checkout at {ref_root}"
    # the product package shares the reference's package name: make sure the REFERENCE wins inside this process
    sys.path = [p for p in sys.path if not p.rstrip("/").endswith("doubly-stochastic-dgp_amd")]
    sys.path.insert(0, ref_root)
    for k in [k for k in sys.modules if k == "doubly_stochastic_dgp" or k.startswith("doubly_stochastic_dgp.")]:
        del sys.modules[k]
    import doubly_stochastic_dgp.dgp as ref_dgp
    if not os.path.abspath(ref_dgp.__file__).startswith(os.path.abspath(ref_root)):
        return f"imported {ref_dgp.__file__}, not the reference"
    return gpflow, tf, ref_dgp


def _tril_vec_to_dense(vec, M):
    """GPflow's LowerTriangular free state (D, M(M+1)/2) -> dense (D, M, M) lower-triangular (np.tril_indices order)."""
    D = vec.shape[0]
    out = np.zeros((D, M, M))
    ii, jj = np.tril_indices(M)
    out[:, ii, jj] = vec
    return out


def reference_outputs(name, gpflow, tf, ref_dgp):
    from oracle import dgp_oracle as O
    from tests.golden import cases
    c, X, Y, Z, specs, zs = cases.inputs(name)
    spec, state, _, X, Y, zs, c = cases.build(name)     # the product model is None without a GPU; only values are used
    S, L = c["S"], c["L"]
    cfg = gpflow.settings.get_settings()
    cfg.numerics.jitter_level = c["jitter"]
    with gpflow.settings.temp_settings(cfg), gpflow.session_manager.get_session().as_default() as sess:
        K = gpflow.kernels
        kerns = []
        for i, ks in enumerate(specs):
            cls = {"rbf": K.RBF, "matern52": K.Matern52}[ks["kind"]]
            var = float(O.positive_forward(O.NP, state[f"l{i}.kern_variance_raw"]))
            ls = np.asarray(O.positive_forward(O.NP, state[f"l{i}.kern_lengthscales_raw"]))
            k = cls(ks["input_dim"], variance=var, lengthscales=(ls if ks["ARD"] else float(ls)), ARD=ks["ARD"])
            if ks.get("white_variance") is not None:
                k = k + K.White(ks["input_dim"], variance=float(O.positive_forward(O.NP, state[f"l{i}.white_variance_raw"])))
            kerns.append(k)
        if c.get("classes"):
            lik = gpflow.likelihoods.MultiClass(c["classes"])
        elif c.get("bernoulli"):
            lik = gpflow.likelihoods.Bernoulli()
        else:
            lik = gpflow.likelihoods.Gaussian()
            lik.variance = float(O.positive_forward(O.NP, state["lik_variance_raw"]))
        z_const = [tf.constant(z, dtype=gpflow.settings.float_type) for z in zs]

        class DGPWithZ(ref_dgp.DGP):
            """explicit N(0,1) draws through the reference's own `zs` hook (dgp.py:62,68)"""

            def propagate(self, Xp, full_cov=False, S=1, zs=None):
                return ref_dgp.DGP.propagate(self, Xp, full_cov=full_cov, S=S, zs=zs or z_const)

        model = DGPWithZ(X, Y, Z, kerns, lik, white=c["white"], num_samples=S, num_outputs=c.get("classes"),
                         num_data=c["num_data"])
        for i, layer in enumerate(model.layers):
            layer.feature.Z = state[f"l{i}.Z"]
            layer.q_mu = state[f"l{i}.q_mu"]
            layer.q_sqrt = np.tril(state[f"l{i}.q_sqrt"])
        model.compile()
        out = dict(source=np.array("reference"))
        Fs, Fm, Fv = model.predict_all_layers(X, S)
        for l in range(L):
            out[f"Fmean{l}"], out[f"Fvar{l}"], out[f"F{l}"] = np.asarray(Fm[l]), np.asarray(Fv[l]), np.asarray(Fs[l])
        out["elbo"] = np.array(model.compute_log_likelihood())
        with gpflow.params_as_tensors_for(*model.layers):
            out["kls"] = np.array([sess.run(layer.KL()) for layer in model.layers])
        # gradients w.r.t. GPflow's unconstrained variables == the oracle's state entries
        wanted = {}
        for i, layer in enumerate(model.layers):
            stat = layer.kern.kernels[0] if hasattr(layer.kern, "kernels") else layer.kern
            wanted[f"l{i}.Z"] = layer.feature.Z
            wanted[f"l{i}.q_mu"] = layer.q_mu
            wanted[f"l{i}.q_sqrt"] = layer.q_sqrt
            wanted[f"l{i}.kern_variance_raw"] = stat.variance
            wanted[f"l{i}.kern_lengthscales_raw"] = stat.lengthscales
            if hasattr(layer.kern, "kernels"):
                wanted[f"l{i}.white_variance_raw"] = layer.kern.kernels[1].variance
        if not c.get("classes"):
            wanted["lik_variance_raw"] = model.likelihood.likelihood.variance
        names = sorted(wanted)
        grads = sess.run(tf.gradients(model.likelihood_tensor, [wanted[k].unconstrained_tensor for k in names]),
                         feed_dict=model.initializable_feeds)
        for k, g in zip(names, grads):
            g = np.asarray(g)
            if k.endswith(".q_sqrt"):
                g = _tril_vec_to_dense(g.reshape(state[k].shape[0], -1), state[k].shape[1])
            g = g.reshape(state[k].shape)
            if g.size <= 4096:
                out["grad." + k] = g
            else:
                out["gradnorm." + k] = np.array(np.linalg.norm(g))
                out["gradblock." + k] = g[:, :16, :16].copy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--check-only", action="store_true", help="report deviations, do not rewrite the fixtures")
    args = ap.parse_args()
    got = _import_reference(args.reference)
    if isinstance(got, str):
        print("make_golden_from_reference: cannot run here —", got)
        print("The committed fixtures stay oracle-generated (tests/golden/make_golden.py); parity remains UNPINNED until this "
              "script is run in an environment with gpflow==1.1.1 and tensorflow==1.8.")
        return 0
    gpflow, tf, ref_dgp = got
    from tests.golden import cases
    worst = 0.0
    for name in cases.CASES:
        out = reference_outputs(name, gpflow, tf, ref_dgp)
        path = os.path.join(HERE, f"golden_{name}.npz")
        if os.path.exists(path):
            old = np.load(path)
            for k in old.files:
                if k in out and k != "source":
                    dev = float(np.max(np.abs(np.asarray(out[k], dtype=np.float64) - old[k])) / (np.max(np.abs(old[k])) + 1e-300))
                    worst = max(worst, dev)
                    if dev > 1e-7:
                        print(f"  {name}:{k} deviates from the committed vector by {dev:.3e} (relative to its largest entry)")
        if not args.check_only:
            np.savez_compressed(path, **out)
        print(f"{name}: elbo {float(out['elbo']):.12g} ({'checked' if args.check_only else 'written'})")
    print(f"largest relative deviation reference vs committed fixtures: {worst:.3e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
