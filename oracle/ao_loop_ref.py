"""TEST INFRASTRUCTURE -- numpy restatement of the reference's simulation loop WITH its estimator (README.md:444-626), for one
realisation.  PARITY UNPINNED (MATLAB-only reference; Zs.mat and SNR_10.mat are not shipped: synthetic optics; the reference's model_approx.mat
(A_s, b_s) IS shipped and pins the estimator's linear half in tests/test_golden_model_approx.py).

    phase_res(:,:,k) = phase_valid(:,:,k) [+ phase_cor(:,:,k-1)]                          README.md:446-454
    Y_M (three PSF windows) ; ad_est = lsqminnorm(A_s'*A_s, A_s'*(Y_M - b_s))            README.md:456-480  (estimator_ref)
    x0 = ad_est ; x0_pre = previous ad_est (zeros at the first step)                     README.md:482-488
    b_ref = 0 ; -M1*B*U(k-1) ; -M1*B*U(k-1) - M2*B*U(k-2)                                README.md:490-497
    z = Fast_MPC2(..., b_ref, [], []).mpc_fixed_log_newton(n_fix, k_fix) ; u_prev = U(1:nu)     README.md:547-555, 589
    ad_cor = B*u_prev ; phase_cor(:,:,k) = sum_j ad_cor(j) .* Zs(j+1,:,:)                README.md:590-601

Checker only: imported by tests/, never by the package."""
import numpy as np

from . import estimator_ref as er
from .banded_ref import BandedFastMPC
from .closed_loop_ref import design_matrices


def ao_loop(model, op, phase_valid, n_newton=1, k=1e-2, noise=None):
    A1, A2, B, T = model["A1"], model["A2"], model["B"], model["T"]
    n, m = B.shape
    solver = BandedFastMPC(A1, A2, B, model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], T)
    M1, M2 = design_matrices(A1, A2, T)
    steps = phase_valid.shape[0]
    X0 = np.zeros((steps, n)); U0 = np.zeros((steps, m))
    u1 = np.zeros(m); u2 = np.zeros(m); x0_last = np.zeros(n)
    phase_cor = np.zeros_like(phase_valid[0])
    for s in range(steps):
        scrn = phase_valid[s] + (phase_cor if s >= 1 else 0.0)
        ad_est, _ = er.estimator_step(scrn, op["pupil"], op["W"], op["zd_list"], op["dx"], op["A_s"], op["b_s"],
                                      None if noise is None else noise[s], AU=op["AU"])
        x0 = ad_est
        x0_pre = x0_last if s >= 1 else np.zeros(n)
        w = np.zeros(T * n)
        if s >= 1:
            w -= M1 @ (B @ u1)
        if s >= 2:
            w -= M2 @ (B @ u2)
        z, _, _, _ = solver.solve(x0, x0_pre, w, n_newton, k)
        u0 = z[:m].copy()
        X0[s], U0[s] = x0, u0
        phase_cor = np.tensordot(B @ u0, op["Z"][1:], axes=1)
        u2, u1, x0_last = u1, u0, x0
    return {"ad_est": X0, "u0": U0}
