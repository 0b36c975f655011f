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


# ---------------------------------------------------------------------------------------------------------------------------
# Synthetic optics for the estimator (README.md:456-480): the reference loads its Zernike modes (Zs.mat) and its linearised
# image model (model_approx.mat: A_s, b_s).  Zs.mat is not shipped; model_approx.mat IS (tests/golden/model_approx_As_bs.npz holds it,
# tests/test_golden_model_approx.py pins the estimator's linear half to it) but belongs to the reference's own optics, which cannot be
# rebuilt without Zs.mat -- so the image-formation tests use these self-consistent stand-ins, which follow the README's pixel arrays
# (README.md:236-250, 366-396): OSA/ANSI-indexed, Noll-normalised Zernike modes on the len x len grid x = (-N:2:N)/N, the
# pin-hole pupil, and y = b_s + A_s alpha linearised at alpha = 0 from the same image formation (README.md:406).
def _zernike_radial(nr, ma, rho):
    import math
    R = np.zeros_like(rho)
    for s_ in range((nr - ma) // 2 + 1):
        c_ = (-1) ** s_ * math.factorial(nr - s_) / (math.factorial(s_) * math.factorial((nr + ma) // 2 - s_) * math.factorial((nr - ma) // 2 - s_))
        R += c_ * rho ** (nr - 2 * s_)
    return R


def zernike_modes(length, n_modes=28):
    """(n_modes, len, len): OSA/ANSI index j = 0 (piston), 1, 2 (tilts), 3, 4 (defocus: the reference's idx2 = 5, 1-based), ...;
    unit RMS over the disk, zero outside it."""
    N = length - 1
    x = np.arange(-N, N + 1, 2) / N
    X, Y = np.meshgrid(x, x)
    rho, th = np.hypot(X, Y), np.arctan2(Y, X)
    inside = rho <= np.max(np.abs(x))
    Z = np.zeros((n_modes, length, length))
    j = 0
    nr = 0
    while j < n_modes:
        for mz in range(-nr, nr + 1, 2):
            if j >= n_modes:
                break
            R = _zernike_radial(nr, abs(mz), rho)
            norm = np.sqrt((2 if mz else 1) * (nr + 1))
            Z[j] = norm * R * (np.sin(-mz * th) if mz < 0 else np.cos(mz * th)) * inside
            j += 1
        nr += 1
    return Z


def estimator_optics(length=512, dx=6.5e-6, n_modes=28, zd_dist=3.0, AU=1e12):
    """Everything the estimator is built from: pupil, diversity mode W (defocus), zd_list, the window, and the linearised model
    A_s (p x (n_modes - 1)), b_s (p) at zero aberration (piston removed, README.md:331)."""
    Z = zernike_modes(length, n_modes)
    df = 1.0 / (length * dx)
    fx = np.arange(-length // 2, length // 2) * df
    FX, FY = np.meshgrid(fx, -fx)
    pupil = (np.sqrt(FX ** 2 + FY ** 2) <= (length / 2 - 1) * df).astype(np.float64)
    W = Z[4]
    zd_list = np.array([-zd_dist, 0.0, zd_dist])
    xaxis = np.arange(-length // 2, length // 2) * dx
    rmin = int(np.nonzero(np.abs(xaxis + 1.0e-4) < 3e-6)[0][0]); rmax = int(np.nonzero(np.abs(xaxis - 1.0e-4) < 3e-6)[0][-1])
    d = rmax - rmin + 1
    nx = n_modes - 1
    b_s = np.zeros(len(zd_list) * d * d); A_s = np.zeros((len(zd_list) * d * d, nx))
    ft = lambda P_: np.fft.fftshift(np.fft.fft2(np.fft.fftshift(P_))) * dx ** 2
    for k_, zd in enumerate(zd_list):
        P0 = pupil * np.exp(1j * zd * W)
        I0 = ft(P0)[rmin:rmax + 1, rmin:rmax + 1]
        b_s[k_ * d * d:(k_ + 1) * d * d] = (np.abs(I0) ** 2 * AU).reshape(-1, order="F")
        for j in range(nx):                       # d|I|^2 / d alpha_j = 2 Re(conj(I) FT(i Z_j P0))
            dI = ft(1j * Z[j + 1] * P0)[rmin:rmax + 1, rmin:rmax + 1]
            A_s[k_ * d * d:(k_ + 1) * d * d, j] = (2.0 * np.real(np.conj(I0) * dI) * AU).reshape(-1, order="F")
    return dict(len=length, dx=dx, AU=AU, Z=Z, pupil=pupil, W=W, zd_list=zd_list, range_min=rmin, range_max=rmax, d=d, A_s=A_s, b_s=b_s, nx=nx)
