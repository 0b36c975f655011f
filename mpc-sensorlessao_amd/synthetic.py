"""Seeded synthetic inputs for tests and bench.  There is no network and the
reference's data files (Zs.mat, SNR_10.mat) are not shipped, so the VAR(2) Zernike-coefficient
model and the turbulence realisations are generated here with the sizes and weights of the
reference notebook (README.md:333-356: Q = 1.5e4 I, Qf = Q, R = I, u in [-28, 28] rad,
x_min/x_max = -/+100 used only for the cold start, README.md:538-540)."""
from __future__ import annotations

import numpy as np

MODEL_SEED = 20211001


def radial_order(noll_j):
    """Radial order of Noll mode j (j = 1 is piston)."""
    return int(np.ceil((-3.0 + np.sqrt(9.0 + 8.0 * (noll_j - 1))) / 2.0))


def make_model(n=27, m=144, T=30, seed=MODEL_SEED, var_order=2):
    """Shared model: A1, A2 (per-mode AR(2) poles + weak coupling), B, weights, bounds."""
    rng = np.random.default_rng(seed)
    rho = rng.uniform(0.90, 0.995, n)
    theta = rng.uniform(0.0, 0.15, n)
    A1 = np.diag(2 * rho * np.cos(theta)) + 0.01 * rng.standard_normal((n, n)) / np.sqrt(n)
    A2 = np.diag(-rho ** 2) + 0.01 * rng.standard_normal((n, n)) / np.sqrt(n)
    B = 0.05 * rng.standard_normal((n, m))
    if var_order == 1:
        A1 = np.diag(rho) + 0.01 * rng.standard_normal((n, n)) / np.sqrt(n)
        A2 = np.zeros((n, n))
    comp = np.block([[A1, A2], [np.eye(n), np.zeros((n, n))]])
    sr = np.max(np.abs(np.linalg.eigvals(comp)))
    if sr >= 0.999:
        c = 0.998 / sr
        A1, A2 = A1 * c, A2 * c * c
    return dict(n=n, m=m, T=T, var_order=var_order, A1=A1, A2=A2, B=B,
                Q=1.5e4 * np.eye(n), Qf=1.5e4 * np.eye(n), R=np.eye(m),
                u_min=-28.0 * np.ones(m), u_max=28.0 * np.ones(m),
                x_min=-100.0 * np.ones(n), x_max=100.0 * np.ones(n))


def make_realisation(model, r=0, steps=2000, burn_in=200, mean_norm=3.0):
    """Coefficient series a[k] of realisation r (seed 1000+r): VAR(2) driven by innovations whose
    std falls with radial order as n_rad^(-11/6), scaled so that mean ||a[k]||_2 = mean_norm."""
    n = model["n"]
    rng = np.random.default_rng(1000 + r)
    sig = np.array([radial_order(j + 2) ** (-11.0 / 6.0) for j in range(n)])   # piston removed
    total = burn_in + steps + 1
    e = rng.standard_normal((total, n)) * sig[None, :]
    a = np.zeros((total, n))
    A1, A2 = model["A1"], model["A2"]
    for k in range(2, total):
        a[k] = A1 @ a[k - 1] + A2 @ a[k - 2] + e[k]
    a = a[burn_in:]
    a *= mean_norm / np.mean(np.linalg.norm(a[1:], axis=1))
    return a                                           # steps+1 rows: a[0] is the "previous" of a[1]


def make_replay_batch(model, r=0, steps=2000, with_nu0=True):
    """Replay batch: problem k has x0 = a[k], x0_pre = a[k-1], w = 0 (T*n).  nu0 ~ U(0,1), seed
    5000+r (stands in for `nu = rand(length(b),1)`, inf_newton_solver.m:2)."""
    a = make_realisation(model, r, steps)
    x0 = np.ascontiguousarray(a[1:steps + 1])
    x0_pre = np.ascontiguousarray(a[0:steps])
    out = dict(x0=x0, x0_pre=x0_pre, w=None)
    if with_nu0:
        out["nu0"] = np.random.default_rng(5000 + r).random((steps, model["T"] * model["n"]))
    return out


def make_test_problem(n=8, m=5, T=10, seed=0, umax=2.0, xf=False, var_order=2, batch=1):
    """The reference demo's configuration (Fast_MPC/VAR_2/test_fast_mpc.m:8-37): Q = I, R = I,
    Qf = 50 I, Xmax = 10, Umax = 2, random A with spectral radius 1, random B, w, x0 in U(0,1)."""
    rng = np.random.default_rng(seed)
    A1 = rng.random((n, n))
    A1 /= np.max(np.abs(np.linalg.eigvals(A1)))
    A2 = 0.3 * rng.standard_normal((n, n)) / np.sqrt(n) if var_order == 2 else np.zeros((n, n))
    B = rng.random((n, m))
    model = dict(n=n, m=m, T=T, var_order=var_order, A1=A1, A2=A2, B=B, Q=np.eye(n), R=np.eye(m),
                 Qf=50.0 * np.eye(n), u_min=-umax * np.ones(m), u_max=umax * np.ones(m),
                 x_min=-10.0 * np.ones(n), x_max=10.0 * np.ones(n),
                 xf=np.ones(n) if xf else None)
    nb = T + (1 if xf else 0)
    data = dict(x0=rng.random((batch, n)), x0_pre=rng.random((batch, n)),
                w=rng.random((batch, T * n)), nu0=rng.random((batch, nb * n)))
    return model, data
