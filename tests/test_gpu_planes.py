"""The RGB + depth frame as two planes (ABI 5: ts_raster_fwd_planes / ts_raster_bwd_planes, TS_FRAME_PLANES):
what the adapter hands out as ``rgb`` and ``extras["depth"]`` (rasterize.py:45, :51) comes out of the 4-channel
compositing pass as two contiguous images, and their gradients go back in as two planes.  Only the layout of the
image changes, so everything must be bitwise what the interleaved 4-channel frame gives."""
import pytest
import torch

from helpers import scene_args

DEV = "cuda:0"

pytestmark = pytest.mark.gpu

CASES = [(60000, 3, 800, 450, 2.0, None),        # one wave per tile, wide lists
         (20000, 1, 333, 211, 5.0, None),        # image not a multiple of the tile, split mapping
         (50000, 2, 1000, 520, 3.0, (7, 19)),    # a tile-row stripe
         (3000, 0, 40, 24, 8.0, None)]


def _grads(md, xys):
    return [xys.grad] + [p.grad for p in md.parameters()]


@pytest.mark.parametrize("n,sh,w,h,mult,rows", CASES)
def test_planes_are_the_interleaved_frame_bit_for_bit(n, sh, w, h, mult, rows):
    from tinysplat_amd import frame
    from tinysplat_amd.rasterizer import camera_on_device
    dev = torch.device(DEV)
    model, cam = scene_args(n, sh, w, h, seed=91 + sh, scale_mult=mult)
    g = torch.Generator().manual_seed(92)
    model.opacities = torch.empty(n, 1).uniform_(-5.0, 8.0, generator=g)
    rows_px = h if rows is None else min(h, 16 * rows[1]) - 16 * rows[0]
    wr, wd = torch.rand(rows_px, w, 3, generator=g).to(DEV), torch.rand(rows_px, w, generator=g).to(DEV)
    view, projview, origin = camera_on_device(cam, dev)
    args = (view[:3, :], projview, origin, cam.f_x, cam.f_y, w, h)

    # which outputs take part in the loss: both, RGB only, depth only
    for use_rgb, use_depth in ((True, True), (True, False), (False, True)):
        md = model.to(DEV).requires_grad_(True)
        img, xys, radii = frame.render_frame(md, *args, True, tile_rows=rows)
        loss = 0.0
        if use_rgb:
            loss = loss + (img[..., :3] * wr).sum()
        if use_depth:
            loss = loss + (img[..., 3] * wd).sum()
        loss.backward()
        ref = _grads(md, xys)

        mp = model.to(DEV).requires_grad_(True)
        rgb, depth, xys_p, radii_p = frame.render_frame_planes(mp, *args, tile_rows=rows)
        assert rgb.is_contiguous() and depth.is_contiguous()
        assert rgb.shape == (rows_px, w, 3) and depth.shape == (rows_px, w)
        assert torch.equal(rgb, img[..., :3]) and torch.equal(depth, img[..., 3])
        assert torch.equal(radii_p, radii) and torch.equal(xys_p, xys)
        loss = 0.0
        if use_rgb:
            loss = loss + (rgb * wr).sum()
        if use_depth:
            loss = loss + (depth * wd).sum()
        loss.backward()
        for a, b in zip(ref, _grads(mp, xys_p)):
            assert torch.equal(a, b), (use_rgb, use_depth)

    with torch.no_grad():                      # the viewer's frame, same layout
        v_rgb, v_depth, _, _ = frame.render_view(mp, *args, True, tile_rows=rows, planes=True)
    assert torch.equal(v_rgb, rgb.detach()) and torch.equal(v_depth, depth.detach())


def test_adapter_hands_out_contiguous_images():
    """GaussianRasterizer.__call__ (rasterize.py:26-62) on the fused path: rgb and extras['depth'] are separate
    contiguous tensors, with and without autograd, and equal the three-op recipe's outputs."""
    from tinysplat_amd.rasterizer import GaussianRasterizer
    model, cam = scene_args(30000, 2, 640, 360, seed=95, scale_mult=3.0)
    md = model.to(DEV).requires_grad_(True)
    fused = GaussianRasterizer(md, None, device=torch.device(DEV))
    rgb, extras = fused(cam)
    assert rgb.is_contiguous() and extras["depth"].is_contiguous() and rgb.requires_grad and extras["depth"].requires_grad
    (rgb.sum() + extras["depth"].sum()).backward()
    g_fused = [p.grad.clone() for p in md.parameters()]
    with torch.no_grad():
        rgb_v, extras_v = fused(cam)
    assert rgb_v.is_contiguous() and extras_v["depth"].is_contiguous()
    assert torch.equal(rgb_v, rgb.detach()) and torch.equal(extras_v["depth"], extras["depth"].detach())

    for p in md.parameters():
        p.grad = None
    plain = GaussianRasterizer(md, None, device=torch.device(DEV), fused_colors=False)
    rgb2, extras2 = plain(cam)
    assert (rgb2 - rgb).abs().max().item() <= 1e-5
    assert ((extras2["depth"] - extras["depth"]).abs().max().item()
            <= 1e-5 * max(1.0, float(extras2["depth"].detach().abs().max())))
    (rgb2.sum() + extras2["depth"].sum()).backward()
    for a, p in zip(g_fused, md.parameters()):
        if a.numel():
            assert (a - p.grad).abs().max().item() <= 1e-5 * max(1.0, float(p.grad.abs().max()))
