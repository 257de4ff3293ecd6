"""Forward-only viewer frame.

Mirrors the render half of /root/reference/tinysplat/viewer.py:79-98 (``process_async_queue``):
take a pose from a client message, update the client's camera, render under ``torch.no_grad()``
with a black background, bring the image to the host and scale it to 0..255.  The websocket
transport and the JPEG encoder around it (viewer.py:16-56, cv2 / websockets) are out of scope.

What is MI355X-specific: the frame runs through ``frame.render_view`` - the render kernels with
every backward-only output dropped - and the x255 scaling is done on the GPU into a pinned host
buffer, so a request costs the kernels plus one 24.9 MB (1080p float32) or 6.2 MB (uint8) copy.
"""
from __future__ import annotations

import copy
from typing import Optional

import numpy as np
import torch

from .scene import Scene


class ViewRenderer:
    def __init__(self, model, camera, device="cuda:0"):
        """``camera``: the template every client camera is copied from (viewer.py:61-62 copies
        ``scene.cameras[0]``)."""
        self.device = torch.device(device)
        self.model = model
        self.camera = copy.copy(camera)
        self.scene = Scene([self.camera], model, device=self.device)      # viewer.py:61-62, :92
        self.rasterizer = self.scene.rasterizer
        self._pinned = {}

    def _host_buffer(self, shape, dtype):
        """Two pinned buffers per (shape, dtype), used alternately: the array handed out by one
        request stays valid while the next request renders (an encoder / websocket send may still
        hold it, viewer.py:44-56) and is overwritten by the request after that."""
        key = (tuple(shape), dtype)
        pair = self._pinned.get(key)
        if pair is None:
            pair = [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(2)]
            self._pinned = {key: pair}
        pair.reverse()
        return pair[0]

    def render(self, position, quat, as_uint8: bool = False) -> np.ndarray:
        """One ``renderRequest`` (viewer.py:82-95) -> image [H, W, 3] scaled to 0..255:
        float32 exactly as the reference hands it to its encoder (``img * 255``), or rounded
        uint8 when ``as_uint8`` (what a JPEG encoder consumes).  The returned array is a view of a
        pinned host buffer that stays untouched until the SECOND following call (double-buffered);
        copy it to keep it longer."""
        self.camera.update_view_matrix(np.asarray(position, dtype=np.float32),
                                       np.asarray(quat, dtype=np.float32))                 # :84-87
        with torch.no_grad():                                                               # :90
            self.model.background = torch.zeros(3, device=self.device)                     # :91
            img, _extras = self.scene.render(self.camera)                                   # :92
            img = img * 255                                                                 # :94
            if as_uint8:
                img = img.clamp_(0, 255).round_().to(torch.uint8)
            host = self._host_buffer(img.shape, img.dtype)
            host.copy_(img, non_blocking=True)                                              # :93
        torch.cuda.current_stream(self.device).synchronize()
        return host.numpy()
