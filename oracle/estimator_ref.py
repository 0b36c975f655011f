"""TEST INFRASTRUCTURE -- numpy restatement of the reference's phase-diversity estimator (README.md:456-480 with the pixel
arrays of README.md:236-240, 366-396), op for op:
    for k = 1:numel(zd_list)
        kW = zd_list(k) .* squeeze(Zs(idx2,:,:));              P_defocus = pupil .* exp(1i*(scrn + kW));
        I_defocus = fftshift(fft2(fftshift(P_defocus), res, res)) * dx^2;     im = abs(I_defocus).^2;
        v_im(:,:,k) = im(range_min:range_max, range_min:range_max) * AU;      Y_M = [Y_M; reshape(v_im(:,:,k), [], 1)];
    Y_M = Y_M + Y_M_noise;       ad_est = lsqminnorm(A_s'*A_s, A_s'*(Y_M - b_s));
PARITY UNPINNED for the image formation: the reference is MATLAB only and ships neither Zs.mat (the mode maps) nor SNR_10.mat, so
those checks run on synthetic Zernike modes and a model linearised from this same image formation.  The LINEAR half is pinned: the
reference does ship model_approx.mat (A_s, b_s; MATLAB v7.3), read by tests/golden/read_mat73.py into tests/golden/model_approx_As_bs.npz
and used by tests/test_golden_model_approx.py (`estimate` below against independent routes on the real model).  Checker only: nothing outside
tests/, smoke() and bench.py's CPU leg may import this module."""
import numpy as np


def window_range(length, dx, mag=1):
    """README.md:370-380: range_min / range_max of the +-1e-4 m window, 0-based inclusive."""
    res = length * mag
    xaxis_res = np.arange(-res // 2, res // 2) * dx / mag
    lo = np.nonzero(np.abs(xaxis_res - (-1.0e-4)) < 3e-6)[0]
    hi = np.nonzero(np.abs(xaxis_res - (+1.0e-4)) < 3e-6)[0]
    return int(lo[0]), int(hi[-1])


def pupil_mask(length, dx):
    """README.md:238, 383-391: the pin-hole pupil on the frequency grid of the len x len array."""
    df = 1.0 / (length * dx)
    fxaxis = np.arange(-length // 2, length // 2) * df
    FX, FY = np.meshgrid(fxaxis, -fxaxis)
    freq_rad = np.sqrt(FX ** 2 + FY ** 2)
    maxfreq = (length / 2 - 1) * df
    return (freq_rad <= 1.0 * maxfreq).astype(np.float64)


def measurements(scrn, pupil, W, zd_list, dx, AU=1e12, mag=1):
    """Y_M of README.md:461-472 (without the noise): the window of the three PSFs, column-major per diversity."""
    length = scrn.shape[0]
    res = length * mag
    rmin, rmax = window_range(length, dx, mag)
    Y = []
    for zd in zd_list:
        P = pupil * np.exp(1j * (scrn + zd * W))
        I = np.fft.fftshift(np.fft.fft2(np.fft.fftshift(P), s=(res, res))) * dx ** 2
        im = np.abs(I) ** 2
        v = im[rmin:rmax + 1, rmin:rmax + 1] * AU
        Y.append(v.reshape(-1, order="F"))
    return np.concatenate(Y)


def estimate(A_s, b_s, Y_M):
    """ad_est = lsqminnorm(A_s'*A_s, A_s'*(Y_M - b_s)) (README.md:478): minimum-norm least squares of the normal equations."""
    return np.linalg.lstsq(A_s.T @ A_s, A_s.T @ (Y_M - b_s), rcond=None)[0]


def estimator_step(scrn, pupil, W, zd_list, dx, A_s, b_s, noise=None, AU=1e12):
    Y = measurements(scrn, pupil, W, zd_list, dx, AU)
    if noise is not None:
        Y = Y + noise
    return estimate(A_s, b_s, Y), Y
