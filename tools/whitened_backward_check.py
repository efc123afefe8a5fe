#!/usr/bin/env python
"""CPU check (torch fp64 autograd) of the reverse pass of one NON-white SVGP layer written in WHITENED coordinates — the plan of
DESIGN.md section 9 for the training step (round 3 uses the whitened form for forward-only evaluations only).

With Lu = chol(Ku), V_d = Lu^-1 q_sqrt_d, nL = Lu^-1 q_mu and, per data row, a1 = Lu^-1 k:
    mean_d = a1^T nL_d,      var_d = kdiag - a1^T a1 + |V_d^T a1|^2                      (layers.py:184-217 in these coordinates)
the chain kernels would need  a1  only (no a = Lu^-T a1: one triangular product less per row block in BOTH passes, and
k_bar = Lu^-T a1_bar — triangular — instead of the dense Ku^-1 abar).  Everything below the row level is M x M algebra:
    P_d   = sum_r vbar_rd a1_r a1_r^T            (the split-K product of today, with a1 in the place of a)
    nLbar = sum_r a1_r mbar_r^T,   Vbar_d = 2 P_d V_d
    sum_r a1bar_r a1_r^T = nL nLbar^T + sum_d 2 V_d V_d^T P_d - 2 sum_d P_d           (no E A^T product: the alg_g identity again)
    Lbar  = -tril( Lu^-T [sum_r a1bar_r a1_r^T] )  - tril( qmubar nL^T ) - tril( sum_d Tbar'_d V_d^T )
    qmubar = Lu^-T nLbar,          Tbar'_d = Lu^-T Vbar_d,     q_sqrt_bar_d = tril(Tbar'_d)
    Kubar = sym( Lu^-T Phi(Lu^T Lbar) Lu^-1 ),   Phi = tril with the diagonal halved          (the Cholesky adjoint of the white=True tail)
(+ dl/dk_r = Lu^-T a1bar_r for the Gram adjoint, unchanged in form).  This script draws a random layer and random upstream adjoints
and compares every one of these with autograd of the plain (non-white) formulas.  Run: python tools/whitened_backward_check.py"""
import torch

torch.manual_seed(0)
dt = torch.float64
M, D, R = 12, 3, 40


def phi(X):
    return torch.tril(X) - 0.5 * torch.diag(torch.diagonal(X))


def main():
    B = torch.randn(M, M, dtype=dt)
    Ku = (B @ B.T + M * torch.eye(M, dtype=dt)).requires_grad_(True)
    K = torch.randn(M, R, dtype=dt).requires_grad_(True)                       # Kuf columns
    q_mu = torch.randn(M, D, dtype=dt).requires_grad_(True)
    T = torch.tril(torch.randn(D, M, M, dtype=dt) * 0.3 + torch.eye(M, dtype=dt)).requires_grad_(True)
    mbar, vbar = torch.randn(R, D, dtype=dt), torch.randn(R, D, dtype=dt)      # upstream adjoints of mean / var
    kdiag = 1.7

    # ---- plain non-white form (reference op sequence), autograd
    Lu = torch.linalg.cholesky(Ku)
    A = torch.linalg.solve_triangular(Lu.T, torch.linalg.solve_triangular(Lu, K, upper=False), upper=True)    # Ku^-1 Kuf
    mean = A.T @ q_mu
    Tl = torch.tril(T)
    SK = Tl @ Tl.transpose(1, 2) - Ku                                                                        # layers.py:195
    var = kdiag + torch.einsum("mr,dmn,nr->rd", A, SK, A)
    loss = (mbar * mean).sum() + (vbar * var).sum()
    gKu, gK, gmu, gT = torch.autograd.grad(loss, [Ku, K, q_mu, T])
    gKu = 0.5 * (gKu + gKu.T)

    # ---- whitened coordinates, by hand
    with torch.no_grad():
        Linv = torch.linalg.inv(Lu)
        V = Linv @ Tl                                    # (D, M, M), lower-triangular
        nL = Linv @ q_mu
        a1 = Linv @ K                                    # (M, R)
        c = torch.einsum("dmn,mr->dnr", V, a1)           # c_d = V_d^T a1
        mean_w = a1.T @ nL
        var_w = kdiag - (a1 * a1).sum(0)[:, None] + (c * c).sum(1).T
        assert torch.allclose(mean_w, mean, rtol=1e-11, atol=1e-12) and torch.allclose(var_w, var, rtol=1e-10, atol=1e-11)
        g = vbar.sum(1)                                                                        # (R,)
        a1bar = nL @ mbar.T + 2.0 * torch.einsum("dmn,dnr,rd->mr", V, c, vbar) - 2.0 * a1 * g[None, :]
        kbar = Linv.T @ a1bar
        P = torch.einsum("rd,mr,nr->dmn", vbar, a1, a1)                                        # P_d
        nLbar = a1 @ mbar
        Vbar = 2.0 * P @ V
        S1 = nL @ nLbar.T + 2.0 * (V @ V.transpose(1, 2) @ P).sum(0) - 2.0 * P.sum(0)          # sum_r a1bar_r a1_r^T without E A^T
        assert torch.allclose(S1, a1bar @ a1.T, rtol=1e-10, atol=1e-10)
        qmubar = Linv.T @ nLbar
        Tbp = Linv.T @ Vbar                                                                     # (D, M, M)
        Lbar = -torch.tril(Linv.T @ S1) - torch.tril(qmubar @ nL.T) - torch.tril((Tbp @ V.transpose(1, 2)).sum(0))
        Kubar = Linv.T @ phi(Lu.T @ Lbar) @ Linv
        Kubar = 0.5 * (Kubar + Kubar.T)
        checks = dict(dl_dKuf=(kbar, gK), dl_dq_mu=(qmubar, gmu), dl_dq_sqrt=(torch.tril(Tbp), torch.tril(gT)), dl_dKu=(Kubar, gKu))
        worst = 0.0
        for name, (got, ref) in checks.items():
            err = float((got - ref).abs().max() / ref.abs().max())
            worst = max(worst, err)
            print(f"{name:12s} rel. deviation from autograd {err:.2e}")
        assert worst < 1e-9
        print("whitened reverse pass == autograd of the plain form")


if __name__ == "__main__":
    main()
