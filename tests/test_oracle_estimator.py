"""CPU checks of the estimator's numpy restatement (oracle/estimator_ref.py, README.md:456-480) and of the synthetic optics it
is exercised on: the window of the FFT equals the partial DFT written out, the linearised model is the derivative of the image
formation, the window indices are the README's."""
import importlib

import numpy as np

from oracle import estimator_ref as er

pkg = importlib.import_module("mpc-sensorlessao_amd")


def test_window_indices_of_the_readme():
    # README.md:370-380 with len = 512, dx = 6.5e-6: samples -15 .. +15 around the centre (MATLAB 242 .. 272)
    assert er.window_range(512, 6.5e-6) == (241, 271)
    assert er.window_range(128, 6.5e-6) == (49, 79)


def test_fft_window_equals_the_partial_dft():
    op = pkg.synthetic.estimator_optics(64)
    rng = np.random.default_rng(3)
    scr = 0.4 * rng.standard_normal((64, 64))
    Y = er.measurements(scr, op["pupil"], op["W"], op["zd_list"], op["dx"], AU=op["AU"])
    L, d, first = 64, op["d"], op["range_min"]
    yy = np.arange(L) - L // 2
    F = np.exp(-2j * np.pi * np.outer(yy, first + np.arange(d) - L // 2) / L)            # F[y][j]
    for k, zd in enumerate(op["zd_list"]):
        P = op["pupil"] * np.exp(1j * (scr + zd * op["W"]))
        out = F.T @ P @ F                                                                 # out[u][v]
        im = np.abs(out * op["dx"] ** 2) ** 2 * op["AU"]
        assert np.allclose(im.reshape(-1, order="F"), Y[k * d * d:(k + 1) * d * d], rtol=1e-11, atol=0)


def test_linearised_model_is_the_derivative_of_the_image_formation():
    op = pkg.synthetic.estimator_optics(64)
    Y0 = er.measurements(np.zeros((64, 64)), op["pupil"], op["W"], op["zd_list"], op["dx"], AU=op["AU"])
    assert np.allclose(Y0, op["b_s"], rtol=1e-12)
    h = 1e-5
    for j in (0, 3, 11, 26):
        Yp = er.measurements(+h * op["Z"][j + 1], op["pupil"], op["W"], op["zd_list"], op["dx"], AU=op["AU"])
        Ym = er.measurements(-h * op["Z"][j + 1], op["pupil"], op["W"], op["zd_list"], op["dx"], AU=op["AU"])
        fd = (Yp - Ym) / (2 * h)
        assert np.linalg.norm(fd - op["A_s"][:, j]) <= 1e-6 * np.linalg.norm(op["A_s"][:, j])


def test_minimum_norm_solution_on_a_rank_deficient_model():
    rng = np.random.default_rng(5)
    A = rng.standard_normal((40, 6)); A[:, 5] = A[:, 4]                                   # A'A singular
    y = rng.standard_normal(40); b = rng.standard_normal(40)
    x = er.estimate(A, b, y)
    assert np.allclose(x, np.linalg.pinv(A.T @ A) @ (A.T @ (y - b)), atol=1e-10) and abs(x[4] - x[5]) <= 1e-10
