"""oracle/train_oracle.py held to what CAN be pinned without pytorch_msssim (absent: the SSIM of
train.py:58-63 stays "parity unpinned"): analytic known answers of the SSIM definition the reference's constructor names
(model_gaussian.py:57: SSIM(data_range=1.0, size_average=True, channel=3) - Wang et al. with an 11-tap Gaussian window,
sigma 1.5, K = (0.01, 0.03), 'valid' filtering), and an INDEPENDENT restatement with scipy's separable correlation in
float64 - a second implementation by a different route, so that a slip in the oracle's filtering, window or crop shows."""
import numpy as np
import torch
from scipy.ndimage import correlate1d

from oracle import train_oracle as TO


def _ssim_scipy(x, y, size=11, sigma=1.5, data_range=1.0):
    """mean SSIM of two [H, W, C] float64 arrays: Gaussian-weighted local moments by scipy.ndimage.correlate1d, cropped to
    the positions where the whole window lies inside the image ('valid')."""
    c = np.arange(size, dtype=np.float64) - size // 2
    g = np.exp(-(c ** 2) / (2.0 * sigma ** 2))
    g /= g.sum()
    h = size // 2

    def f(a):
        a = correlate1d(a, g, axis=0, mode="constant")
        a = correlate1d(a, g, axis=1, mode="constant")
        return a[h:a.shape[0] - h, h:a.shape[1] - h]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mx, my = f(x), f(y)
    sx, sy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
    smap = ((2 * mx * my + c1) / (mx * mx + my * my + c1)) * ((2 * sxy + c2) / (sx + sy + c2))
    return smap.reshape(-1, smap.shape[2]).mean(axis=0).mean()


def test_window_is_the_normalised_gaussian():
    w = TO.gauss_window(dtype=torch.float64)
    assert w.numel() == 11 and abs(float(w.sum()) - 1.0) < 1e-15 and torch.equal(w, w.flip(0))
    assert abs(float(w[5] / w[4]) - np.exp(1.0 / (2 * 1.5 ** 2))) < 1e-12


def test_ssim_known_answers():
    g = torch.Generator().manual_seed(1)
    x = torch.rand(40, 52, 3, generator=g, dtype=torch.float64)
    assert abs(float(TO.ssim(x, x)) - 1.0) < 1e-12                              # identical images
    # constant images a, b: every local variance and covariance is 0 -> SSIM = (2ab + C1) / (a^2 + b^2 + C1)
    for a, b in ((0.2, 0.7), (0.0, 1.0), (0.5, 0.5)):
        xa, xb = torch.full((30, 33, 3), a, dtype=torch.float64), torch.full((30, 33, 3), b, dtype=torch.float64)
        want = (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)
        assert abs(float(TO.ssim(xa, xb)) - want) < 1e-12
    # a constant against a zero-mean pattern of variance v around the same mean m: luminance term 1, structure term
    # C2 / (v_local + C2) <= 1, so SSIM < 1 and symmetric in its arguments
    y = x.clone()
    y[..., 0] = 1.0 - y[..., 0]
    assert abs(float(TO.ssim(x, y)) - float(TO.ssim(y, x))) < 1e-14 and float(TO.ssim(x, y)) < 1.0


def test_ssim_equals_an_independent_scipy_restatement():
    g = torch.Generator().manual_seed(2)
    for h, w in ((23, 31), (64, 48), (11, 11)):
        x = torch.rand(h, w, 3, generator=g, dtype=torch.float64)
        y = (x + 0.2 * torch.randn(h, w, 3, generator=g, dtype=torch.float64)).clamp(0, 1)
        got = float(TO.ssim(x, y))
        want = float(_ssim_scipy(x.numpy(), y.numpy()))
        assert abs(got - want) < 1e-12, (h, w, got, want)


def test_photometric_loss_is_the_weighted_sum_of_train_py():
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(20, 24, 3, generator=g, dtype=torch.float64), torch.rand(20, 24, 3, generator=g, dtype=torch.float64)
    loss, l1, s = TO.photometric_loss(x, y, 0.2)                                 # train.py:58-63
    assert abs(float(l1) - float((x - y).abs().mean())) < 1e-15
    assert abs(float(loss) - (0.8 * float(l1) + 0.2 * (1.0 - float(s)))) < 1e-15
