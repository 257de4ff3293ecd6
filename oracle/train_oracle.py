"""CPU oracle for the training-step ops.  TEST INFRASTRUCTURE ONLY.

    *** PARITY UNPINNED for SSIM ***  tinysplat builds its SSIM with the third-party package
    pytorch_msssim (model_gaussian.py:57: SSIM(data_range=1.0, size_average=True, channel=3)), which
    is absent from /root/reference and from this image.  ``ssim`` below restates that package's
    published algorithm (11-tap Gaussian window with sigma 1.5, separable "valid" filtering with a
    grouped conv2d, K = (0.01, 0.03), compensation 1.0, mean over the map then over channels).

Adam is pinned: the reference uses torch.optim.Adam itself (scripts/train.py:26), which is
importable here, so the tests compare against it directly.
"""
import torch
import torch.nn.functional as F


def gauss_window(size: int = 11, sigma: float = 1.5, dtype=torch.float32) -> torch.Tensor:
    coords = torch.arange(size, dtype=dtype) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    c = x.shape[1]
    x = F.conv2d(x, win.view(1, 1, -1, 1).repeat(c, 1, 1, 1), groups=c)
    return F.conv2d(x, win.view(1, 1, 1, -1).repeat(c, 1, 1, 1), groups=c)


def ssim(img_hwc: torch.Tensor, tgt_hwc: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Mean SSIM of two [H,W,3] images, arranged as train.py:60-62 does (permute(2,0,1).unsqueeze(0))."""
    X, Y = img_hwc.permute(2, 0, 1)[None], tgt_hwc.permute(2, 0, 1)[None]
    win = gauss_window(dtype=X.dtype)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _filter(X, win), _filter(Y, win)
    s1 = _filter(X * X, win) - mu1 * mu1
    s2 = _filter(Y * Y, win) - mu2 * mu2
    s12 = _filter(X * Y, win) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    smap = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs
    return smap.flatten(2).mean(-1).mean()


def photometric_loss(img_hwc, tgt_hwc, lambda_dssim: float = 0.2):
    l1 = (img_hwc - tgt_hwc).abs().mean()
    s = ssim(img_hwc, tgt_hwc)
    return (1 - lambda_dssim) * l1 + lambda_dssim * (1 - s), l1, s
