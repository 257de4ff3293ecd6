"""Forward-only (viewer) path, SURVEY.md 8(f) F4: render_view under no_grad gives the bits of the
training-mode forward, matches the oracle frame, and ViewRenderer follows viewer.py:79-98."""
import numpy as np
import pytest
import torch

from helpers import assert_close_masked, oracle_frame, scene_args
from tinysplat_amd.rasterizer import GaussianRasterizer
from tinysplat_amd.synthetic import PinholeCamera
from tinysplat_amd.viewer import ViewRenderer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_no_grad_frame_is_bitwise_the_training_forward_and_matches_oracle():
    n, sh, w, h = 30000, 2, 400, 300
    model, cam = scene_args(n, sh, w, h, seed=8, scale_mult=3.0)
    md = model.to(DEV).requires_grad_(True)
    r = GaussianRasterizer(md, None, device=torch.device(DEV))
    rgb_t, ex_t = r(cam, (w, h), sh)
    with torch.no_grad():
        rgb_v, ex_v = r(cam, (w, h), sh)
    assert not rgb_v.requires_grad and rgb_t.requires_grad
    assert torch.equal(rgb_v, rgb_t.detach()) and torch.equal(ex_v["depth"], ex_t["depth"].detach())
    assert torch.equal(ex_v["radii"], ex_t["radii"]) and torch.equal(ex_v["xys"], ex_t["xys"].detach())
    f = oracle_frame(model, cam, (w, h), depth=True)
    stable = f["aux"]["margin"] > 1e-4
    assert_close_masked(rgb_v, f["rgb"], 1e-5, stable, what="rgb")
    assert_close_masked(ex_v["depth"], f["depth"], 1e-5, stable, what="depth", scale_by_value=True)


def test_view_renderer_follows_the_reference_request_handler():
    n, sh, w, h = 20000, 1, 320, 200
    model, _ = scene_args(n, sh, w, h, seed=9, scale_mult=3.0)
    model.background = torch.tensor([0.3, 0.3, 0.3])          # the viewer overrides it with zeros (:91)
    template = PinholeCamera.look_at_origin_plus_z(w, h)
    vr = ViewRenderer(model.to(DEV), template, DEV)
    pos, quat = [0.2, -0.1, -0.5], [0.9914449, 0.0, 0.1305262, 0.0]
    img = vr.render(pos, quat)
    assert img.shape == (h, w, 3) and img.dtype == np.float32
    # the same request done by hand through the oracle
    cam = PinholeCamera.look_at_origin_plus_z(w, h)
    cam.update_view_matrix(np.asarray(pos, dtype=np.float32), np.asarray(quat, dtype=np.float32))
    assert torch.equal(cam.view_matrix, vr.camera.view_matrix)
    model.background = torch.zeros(3)
    f = oracle_frame(model, cam, (w, h), depth=False)
    stable = (f["aux"]["margin"] > 1e-4).numpy()
    assert np.abs(img - f["rgb"].numpy() * 255)[stable].max() <= 1e-5 * 255
    u8 = vr.render(pos, quat, as_uint8=True)
    assert u8.dtype == np.uint8 and np.abs(u8.astype(np.float32) - np.clip(img, 0, 255)).max() <= 0.5 + 1e-3
    # a second pose re-uses the pinned buffer and really moves the camera
    img2 = vr.render([0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]).copy()
    assert np.abs(img2 - vr.render(pos, quat)).max() > 1.0


def test_two_host_threads_render_on_one_device():
    """A viewer thread beside the training loop (the reference runs its viewer server next to train.py):
    the frame path shares one pinned count word per device, guarded by a lock - frames issued from two
    threads must not pick up each other's intersection counts (a wrong count would mis-size the list
    buffers).  Each thread re-renders its own scene and must see the same image every time."""
    import threading

    from tinysplat_amd import frame
    from tinysplat_amd.rasterizer import camera_on_device
    dev = torch.device(DEV)
    res = {}

    def work(tag, n, w, h):
        model, cam = scene_args(n, 1, w, h, seed=tag)
        md = model.to(dev)
        view, projview, origin = camera_on_device(cam, dev)
        ref = None
        for it in range(60):
            with torch.no_grad():
                img, _, _ = frame.render_view(md, view[:3, :], projview, origin, cam.f_x, cam.f_y, w, h, True)
            if ref is None:
                ref = img.clone()
            elif not torch.equal(ref, img):
                res[tag] = f"frame {it} differs"
                return
        res[tag] = "ok"

    threads = [threading.Thread(target=work, args=(1, 30000, 640, 360)),
               threading.Thread(target=work, args=(2, 90000, 800, 450))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert res == {1: "ok", 2: "ok"}, res
