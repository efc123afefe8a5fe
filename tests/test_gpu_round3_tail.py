"""-m gpu: the fused tail of the reverse pass (k_asm_rows + k_tail, Adam inside the tail launch: dsdgp_model_train_step) against the
separate kernels (DSDGP_FORCE=tail=0) and against the oracle's Adam trajectory."""
import numpy as np
import pytest

from oracle import dgp_oracle as O
from oracle import model as OM
from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu


def _case(monkeypatch, force, ard, white_noise, M=40, seed=3):
    monkeypatch.setenv("DSDGP_FORCE", force)
    rng = np.random.RandomState(seed)
    N, D, S = 90, 4, 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = rng.randn(M, D)
    ls = (0.7 + rng.rand(D)) if ard else 1.1
    specs = [kern_spec("rbf", D, 1.2, ls, ARD=ard, white_variance=0.05 if white_noise else None),
             kern_spec("matern52", D, 0.8, 0.9), kern_spec("rbf", D, 1.0, 1.3)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=400)
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 2)]
    return X, Y, spec, state, model, zs, S


@pytest.mark.parametrize("ard,white_noise", [(False, False), (True, True)])
def test_fused_tail_gradient_matches_oracle_and_separate_kernels(monkeypatch, ard, white_noise):
    X, Y, spec, state, model, zs, S = _case(monkeypatch, "tail=1", ard, white_noise)
    ref, gref = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=400)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert abs(got - ref) <= 1e-9 * abs(ref)
    g1 = model.engine().gradient_dict()
    for k in gref:
        assert np.max(np.abs(-gref[k] - g1[k])) <= 1e-7 * (np.max(np.abs(gref[k])) + 1e-12), k
    _, _, _, _, model0, _, _ = _case(monkeypatch, "tail=0", ard, white_noise)
    got0 = model0._build_likelihood(X, Y, zs=zs, with_grad=True)
    g0 = model0.engine().gradient_dict()
    assert got0 == got
    for k in g0:
        assert np.max(np.abs(g0[k] - g1[k])) <= 1e-12 * (np.max(np.abs(g0[k])) + 1e-300), k


def test_fused_adam_matches_separate_adam(monkeypatch):
    """dsdgp_model_train_step (Adam inside k_tail) == dsdgp_model_elbo + dsdgp_model_adam_step on the same gradient kernels, bit for
    bit, over several steps; set_trainable(False) entries stay put."""
    outs = []
    for fused in (True, False):
        X, Y, spec, state, model, zs, S = _case(monkeypatch, "tail=1", True, True, M=33, seed=8)
        model.layers[1].feature.Z.trainable = False
        model.likelihood.likelihood.variance.trainable = False
        eng = model.engine()
        eng._upload_if_needed()
        before = eng.theta.cpu().numpy().copy()
        for t in range(4):
            if fused:
                eng.train_step(X, Y, S, zs=zs, seed=0, data_scale=400 / X.shape[0], lr=0.02)
            else:
                eng.elbo(X, Y, S, zs=zs, seed=0, data_scale=400 / X.shape[0], with_grad=True, sync=False)
                eng.adam_step(0.02)
        eng.ctx.sync()
        after = eng.theta.cpu().numpy().copy()
        moved = 0
        for p, off, cnt, kind in eng.entries:
            if p is model.layers[1].feature.Z or p is model.likelihood.likelihood.variance:
                assert np.array_equal(after[off:off + cnt], before[off:off + cnt])
            else:
                moved += int(np.any(after[off:off + cnt] != before[off:off + cnt]))
        assert moved >= len(eng.entries) - 2
        outs.append((after, eng.out4.cpu().numpy().copy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_minibatch_step_in_one_call_matches_gather_then_step(monkeypatch):
    """dsdgp_model_train_step_minibatch (gather inside the head launch) == dsdgp_gather_rows2 + dsdgp_model_train_step on the same
    index stream, bit for bit; also with the separate head kernels (DSDGP_FORCE=head=0: the gather falls back to its own launch)."""
    for force in ("head=1", "head=0"):
        outs = []
        for fused in (True, False):
            monkeypatch.setenv("DSDGP_FORCE", force)
            rng = np.random.RandomState(21)
            N, D, M, S = 330, 3, 24, 4
            X, Y = rng.randn(N, D), rng.randn(N, 2)
            Z = X[:M] + 0.05 * rng.randn(M, D)
            _, _, model = make_case(X, Y, Z, [kern_spec("rbf", D), kern_spec("rbf", D)], S=S, num_data=N, minibatch_size=100)
            for _ in range(9):                       # 330 / 100: minibatches straddle the epoch boundary
                if fused:
                    model.train_step(0.01)
                else:
                    Xb, Yb = model.next_minibatch()
                    model.train_step(0.01, X=Xb, Y=Yb)
            eng = model.engine()
            eng.ctx.sync()
            outs.append((eng.theta.cpu().numpy().copy(), eng.out4.cpu().numpy().copy()))
        assert np.array_equal(outs[0][0], outs[1][0]), force
        assert np.array_equal(outs[0][1], outs[1][1]), force
