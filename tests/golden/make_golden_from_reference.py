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
  * the wider set of tests/golden/extras.py with the reference's own entry points: predict_y and predict_density (dgp.py:117-126),
    predict_all_layers_full_cov (dgp.py:113-114), three gpflow.train.AdamOptimizer(0.01) steps on a fresh model, and one
    gpflow.training.NatGradOptimizer(gamma = 0.1) step on the last layer's (q_mu, q_sqrt) of the Gaussian-likelihood cases
    (demos/demo_regression_UCI.ipynb:360-366) — all with the same injected z draws;
  * reports the largest deviation from the committed (oracle) vectors, and unless --check-only rewrites the .npz with the
    reference's numbers plus `source = "reference"`.
Only data (inputs' recipe, outputs) is stored; no reference source text.

Environment recipe (the versions README.md:4 of the reference names; none of this exists in the build container or on the GPU box):

    conda create -n dsdgp-ref python=3.6 && conda activate dsdgp-ref
    pip install tensorflow==1.8.0 gpflow==1.1.1 numpy==1.14.5 scipy==1.1.0
    git clone https://github.com/UCL-SML/Doubly-Stochastic-DGP reference        # the checkout SURVEY.md describes
    cd <this repository> && python -m tests.golden.make_golden_from_reference --reference ../reference --check-only
    # deviations <= 1e-7 expected; then without --check-only to rewrite tests/golden/*.npz, and commit them:
    # DESIGN.md section 3 / README then say "pinned by reference-generated fixtures" instead of "parity unpinned"

`--dry-run` walks the same comparison / writing code with the ORACLE standing in for the reference and a scratch output directory
(not exercised by the test suite since round 5: this script cannot run in the build container): it proves the plumbing, not parity.
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

        def fresh_model():
            mdl = DGPWithZ(X, Y, Z, kerns, lik, white=c["white"], num_samples=S, num_outputs=c.get("classes"),
                           num_data=c["num_data"])
            for i, layer in enumerate(mdl.layers):
                layer.feature.Z = state[f"l{i}.Z"]
                layer.q_mu = state[f"l{i}.q_mu"]
                layer.q_sqrt = np.tril(state[f"l{i}.q_sqrt"])
            return mdl

        model = fresh_model()
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
        # ---- the wider set (tests/golden/extras.py)
        from tests.golden import extras
        pm, pv = model.predict_y(X, S)                                                  # dgp.py:117-119
        out["x.predy_mean"], out["x.predy_var"] = np.asarray(pm), np.asarray(pv)
        out["x.preddens"] = np.asarray(model.predict_density(X, Y, S))                  # dgp.py:121-126
        _, Fmc, Fvc = model.predict_all_layers_full_cov(X, S)                           # dgp.py:113-114
        out["x.fc_Fmean"], out["x.fc_Fvar"] = np.asarray(Fmc[-1]), np.asarray(Fvc[-1])
        if extras._is_gaussian(c):
            from gpflow.training import NatGradOptimizer
            m_ng = fresh_model()
            last = m_ng.layers[-1]
            last.q_mu.set_trainable(False)
            last.q_sqrt.set_trainable(False)
            m_ng.compile()
            NatGradOptimizer(gamma=extras.NG_GAMMA).minimize(m_ng, var_list=[(last.q_mu, last.q_sqrt)], maxiter=1)
            out["x.ng_q_mu"] = np.asarray(last.q_mu.read_value())
            out.update(extras.pack_q_sqrt("x.ng_q_sqrt", np.asarray(last.q_sqrt.read_value())))
        m_ad = fresh_model()
        m_ad.compile()
        gpflow.train.AdamOptimizer(extras.ADAM_LR).minimize(m_ad, maxiter=extras.ADAM_STEPS)
        out["x.adam3_elbo"] = np.array(m_ad.compute_log_likelihood())
        out["x.adam3_q_mu"] = np.asarray(m_ad.layers[-1].q_mu.read_value())
    return out


def compare_and_write(name, out, out_dir, check_only, committed_dir=HERE):
    """largest relative deviation of `out` from the committed fixture of `name`; (re)writes <out_dir>/golden_<name>.npz unless check_only"""
    worst = 0.0
    old_path = os.path.join(committed_dir, f"golden_{name}.npz")
    if os.path.exists(old_path):
        old = np.load(old_path)
        missing = [k for k in old.files if k not in out and k != "source"]
        if missing:
            print(f"  {name}: keys of the committed fixture that the generator did not produce: {missing}")
            worst = float("inf")
        for k in old.files:
            if k in out and k != "source":
                dev = float(np.max(np.abs(np.asarray(out[k], dtype=np.float64) - old[k])) / (np.max(np.abs(old[k])) + 1e-300))
                worst = max(worst, dev)
                if dev > 1e-7:
                    print(f"  {name}:{k} deviates from the committed vector by {dev:.3e} (relative to its largest entry)")
    if not check_only:
        np.savez_compressed(os.path.join(out_dir, f"golden_{name}.npz"), **out)
    return worst


def oracle_stand_in(name):
    """--dry-run: the oracle's numbers in the reference's place (plumbing check only; says so in `source`)"""
    from tests.golden import make_golden
    out = make_golden.outputs(name)
    out["source"] = np.array("oracle (dry run)")
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--check-only", action="store_true", help="report deviations, do not rewrite the fixtures")
    ap.add_argument("--dry-run", action="store_true", help="oracle in the reference's place, output into --out-dir (plumbing check)")
    ap.add_argument("--out-dir", default=HERE)
    ap.add_argument("--cases", default="", help="comma-separated subset of tests/golden/cases.py (default: all)")
    args = ap.parse_args(argv)
    from tests.golden import cases
    names = [n for n in args.cases.split(",") if n] or list(cases.CASES)
    if args.dry_run:
        if os.path.abspath(args.out_dir) == os.path.abspath(HERE) and not args.check_only:
            print("--dry-run refuses to overwrite the committed fixtures: give --out-dir or --check-only")
            return 2
        compute = oracle_stand_in
    else:
        got = _import_reference(args.reference)
        if isinstance(got, str):
            print("make_golden_from_reference: cannot run here —", got)
            print("The committed fixtures stay oracle-generated (tests/golden/make_golden.py); parity remains UNPINNED until this "
                  "script is run in an environment with gpflow==1.1.1 and tensorflow==1.8 (recipe: this file's docstring).")
            return 0
        gpflow, tf, ref_dgp = got
        compute = lambda name: reference_outputs(name, gpflow, tf, ref_dgp)      # noqa: E731
    worst = 0.0
    for name in names:
        out = compute(name)
        worst = max(worst, compare_and_write(name, out, args.out_dir, args.check_only))
        print(f"{name}: elbo {float(out['elbo']):.12g} ({'checked' if args.check_only else 'written'}; source = {out['source']})")
    print(f"largest relative deviation vs committed fixtures: {worst:.3e}")
    return 0 if worst <= 1e-6 else 1


if __name__ == "__main__":
    sys.exit(main())
