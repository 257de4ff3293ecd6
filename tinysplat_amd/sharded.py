"""Gaussian-sharded multi-GPU frame (the reference is single-device, rasterize.py:17).

One process per GPU.  Rank r OWNS a contiguous range of the Gaussians - their six parameter tensors, and
whatever the caller keeps per Gaussian (Adam moments, the densification accumulator) - and RENDERS one
stripe of tile rows.  A frame on rank r:

  owner stage    project + colour stage over the OWNED Gaussians only (N / G), with the full-frame camera;
                 ``ts_route_count / ts_route_pack`` group a 64-byte export record of every visible Gaussian
                 by the rank(s) whose stripe its tile box reaches
  exchange       all_to_all of the records (RCCL over xGMI with backend "nccl"): a rank receives what its
                 stripe lists - ~N / G records plus the Gaussians straddling a stripe boundary
  stripe stage   ``ts_import_records`` -> scan -> ``ts_import_pack`` -> bin / sort / composite the stripe with
                 the very kernels a single GPU runs (tight lists, wide lists, split blocks as frame.py)
  backward       composite back to front, one 48-byte gradient row per imported record
                 (``ts_reduce_partials_rows``), reverse all_to_all, ``ts_route_accumulate`` sums a Gaussian's
                 rows in ascending stripe order, then the colour stage's and the projection's backward run on
                 the owned Gaussians: every rank ends with the gradients of ITS parameters.

Nothing is replicated and no stage of a rank is O(N): per-Gaussian work is N / G, per-record work ~1.15 N / G,
compositing 1 / G of the pixels, and 64 + 48 bytes per (Gaussian, stripe) pair cross the links - at 1 M
Gaussians on 8 ranks ~13 MB per rank and frame, against the dense (36 + 4 ch) N-byte all-reduce (40 MB on every
rank) plus ~0.25 ms of replicated per-Gaussian stages of the replicated-parameter design (sharding.py, kept).

The stripes tile the single-GPU image bit for bit (records reach a rank ordered by global Gaussian index, so
equal depths tie-break as on one GPU); gradients agree to rounding (a Gaussian's rows are summed per stripe,
then over stripes).  ``Exchange`` abstracts the two collectives so that the same frame code runs under
torch.distributed (``DistExchange``) and against recorded remote records (``ReplayExchange``: bench.py's
single-GPU estimate of one rank's step); ``simulate_frame`` runs the stages of ALL ranks one after the other in
one process on one GPU (full-size functional checks and the per-stage table of a rank on 1-GPU boxes).
"""
from __future__ import annotations

import ctypes
import os
import weakref

import numpy as np
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import _comm
from ._comm import collective_timer
from . import _lib
from . import frame as _frame
from ._lib import TsFrame, TsStripes
from .ops import _call, _camera, kernel_timer, _f32c, _need_hip, _stream, _stripe_rows, _tile_bounds, deg_from_sh
from .rasterizer import camera_on_device
from .sharding import stripe_rows
from .synthetic import SplatModel

RECORD_FLOATS = 16       # TS_EXPORT_RECORD_FLOATS
ROW_FLOATS = 12          # TS_PARTIAL_ROW_FLOATS

# Gaussians up to which a rank's owner stage runs as its small-N fusion (csrc/shard.hip: FUSED OWNER FORWARD; the
# library reads the same variable: 0 = never)
SMALL_N_FUSED = int(os.environ.get("TS_SMALL_N_FUSED", "262144"))
# PADDED EXCHANGE (an option: TS_PADDED_EXCHANGE=1 switches it on): how many records a rank sends to every other rank
# is known only after its owner stage has run, and sizing the all_to_all from it costs a host read in the middle of
# the frame - where the GPU is what bounds a rank's step, it idles while the host waits for the counts, allocates and
# enqueues the rest.  With the option, from the second frame of a layout on, every (source, destination) group gets the
# CAPACITY the previous frame's count matrix suggests (+ 12.5 % + 192 rows, multiples of 64): the send buffer is zeroed
# (an all-zero record lists nothing: radius 0), ts_route_count_padded fixes the groups' bases, the all_to_all runs with
# the capacities as split sizes, and the whole forward pass is enqueued without looking at a count.  The counts travel
# anyway - one all_gather of `world` ints per rank gives EVERY rank the whole matrix - and are read where the frame
# already waits for the stripe's pair count: if any group outgrew its capacity, every rank sees it in the same matrix
# and all of them run the forward pass again with exact sizes (no extra collective to agree on that).
# Why it is not the default: on config 3 at 8 ranks the emulated rank step is bound by the HOST (~0.44 ms of Python per
# step against 0.36 ms of kernels: tools/host_split_rank.py), the GPU is already waiting when the read happens, and
# the option's own bookkeeping (zeroed buffer, pinned copy, event, the check) adds ~40 us of Python: 0.46 -> 0.56 ms
# on the slowest box, 0.43 -> 0.44 ms on the fastest.  On config 5 (GPU-bound, 0.95 ms) it is neutral.  What the
# blocking read costs on eight real devices cannot be measured here.
PADDED_EXCHANGE = os.environ.get("TS_PADDED_EXCHANGE", "0") == "1"
_route_caps = {}         # layout key -> capacity matrix [world][world] derived from the previous frame's counts
_count_slots = {}        # (device index, world) -> pinned int32[world * world] the gathered counts are copied into
padded_frames = [0, 0]   # frames that ran padded / padded frames that had to run again (tests, tools)


def _capacity(c):
    """counts (numpy int array) -> capacities"""
    c = np.asarray(c, dtype=np.int64)
    return ((c + (c >> 3) + 192) // 64 + 1) * 64


# --------------------------------------------------------------------------------------------------
# layout: who owns which Gaussians, who renders which tile rows
# --------------------------------------------------------------------------------------------------
def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced index range of rank's Gaussians (the first n % world ranks own one more)."""
    q, r = divmod(int(n_total), int(world))
    i0 = rank * q + min(rank, r)
    return i0, i0 + q + (1 if rank < r else 0)


class ShardLayout:
    """Static description of a sharded frame: ownership ranges and tile-row stripes of all ranks."""

    def __init__(self, n_total: int, world: int, rank: int, dims: Tuple[int, int],
                 stripes: Optional[Sequence[int]] = None):
        if not (1 <= world <= _lib.MAX_RANKS) or not (0 <= rank < world):
            raise ValueError(f"world size must be 1..{_lib.MAX_RANKS} and 0 <= rank < world")
        self.n_total, self.world, self.rank, self.dims = int(n_total), int(world), int(rank), (int(dims[0]), int(dims[1]))
        self.bounds = [shard_range(n_total, world, k)[0] for k in range(world)] + [int(n_total)]
        tby = _tile_bounds(self.dims[1], self.dims[0])[1]
        if stripes is None:
            stripes = [stripe_rows(tby, world, k)[0] for k in range(world)] + [tby]
        stripes = [int(v) for v in stripes]
        if len(stripes) != world + 1 or any(a > b for a, b in zip(stripes, stripes[1:])) or stripes[0] < 0 \
                or stripes[-1] > tby:
            raise ValueError("stripes must be world + 1 ascending tile rows inside the frame")
        self.stripes = stripes
        self.key = (self.world, self.rank, self.dims, self.n_total, tuple(stripes))
        st = TsStripes()
        st.num = world
        for k, v in enumerate(stripes):
            st.row[k] = v
        self.c_stripes = st

    @property
    def owned(self) -> Tuple[int, int]:
        return self.bounds[self.rank], self.bounds[self.rank + 1]

    @property
    def tile_rows(self) -> Tuple[int, int]:
        return self.stripes[self.rank], self.stripes[self.rank + 1]

    def for_rank(self, rank: int) -> "ShardLayout":
        return ShardLayout(self.n_total, self.world, rank, self.dims, self.stripes)


# --------------------------------------------------------------------------------------------------
# stripes balanced by WORK, not by rows ("optionally balanced by per-row intersection counts from
# the previous frame")
# --------------------------------------------------------------------------------------------------
TILE_COST_PAIRS = float(os.environ.get("TS_TILE_COST_PAIRS", "48"))     # fixed cost of a tile, in listed pairs


def row_work(tile_bins: Tensor, tile_bounds_x: int, row0: int = 0) -> List[float]:
    """Work of every tile row of a launch from its lists: listed pairs + TILE_COST_PAIRS per tile (the per-tile
    prologue / epilogue of the compositing kernels, pixel state, image stores).  ``tile_bins``: [tiles, 2] of 16x16
    lists, row-major.  -> one float per tile row of the launch (the caller knows they start at ``row0``)."""
    lens = (tile_bins[:, 1] - tile_bins[:, 0]).to(torch.float64)
    rows = lens.numel() // int(tile_bounds_x)
    per_row = lens[:rows * tile_bounds_x].view(rows, tile_bounds_x).sum(dim=1) + TILE_COST_PAIRS * tile_bounds_x
    return per_row.cpu().tolist()


def balanced_stripes(work: Sequence[float], world: int) -> List[int]:
    """Cuts the tile rows 0 .. len(work) into ``world`` CONTIGUOUS stripes such that the largest stripe's work is
    minimal and, among the cuts that reach that bottleneck, the loads are as even as possible (smallest sum of squares:
    no rank is left idle - or handed everything that remains - just because one heavy row fixes the bottleneck; ADVICE
    r5: [10, 1, 1] over three ranks is [0, 1, 2, 3], not [0, 1, 3, 3]) -> world + 1 ascending row indices (ShardLayout's
    ``stripes``; stripes are empty only when there are fewer rows than ranks).  Exact: two dynamic programmes over
    (row, stripe), O(rows^2 world) on a few dozen rows."""
    rows = len(work)
    pre = [0.0]
    for v in work:
        pre.append(pre[-1] + float(v))
    INF = float("inf")
    best = [[INF] * (rows + 1) for _ in range(world + 1)]     # best[k][r]: rows 0..r in k stripes, minimal bottleneck
    best[0][0] = 0.0
    for k in range(1, world + 1):
        for r in range(rows + 1):
            for q in range(r + 1):                            # the last stripe = rows q .. r
                if best[k - 1][q] == INF:
                    continue
                v = max(best[k - 1][q], pre[r] - pre[q])
                if v < best[k][r]:
                    best[k][r] = v
    cap = best[world][rows] * (1.0 + 1e-12)
    # second pass: smallest sum of squared loads (then fewest empty stripes) with no stripe above the bottleneck
    even = [[(INF, 0)] * (rows + 1) for _ in range(world + 1)]
    cut = [[0] * (rows + 1) for _ in range(world + 1)]
    even[0][0] = (0.0, 0)
    for k in range(1, world + 1):
        for r in range(rows + 1):
            for q in range(r + 1):
                load = pre[r] - pre[q]
                if even[k - 1][q][0] == INF or load > cap:
                    continue
                v = (even[k - 1][q][0] + load * load, even[k - 1][q][1] + (1 if q == r else 0))
                if v <= even[k][r]:                           # (ties: the later cut - empty stripes go to the LAST ranks)
                    even[k][r], cut[k][r] = v, q
    out, r = [rows], rows
    for k in range(world, 0, -1):
        r = cut[k][r]
        out.append(r)
    return out[::-1]


class StripeBalancer:
    """Keeps a sharded frame's stripes balanced by the previous frame's work.  After a frame, every rank hands in the
    work of ITS tile rows (``row_work`` of its stripe's lists); ``update`` all_gathers them (a few dozen floats per
    rank; every ``every`` frames) and all ranks derive the same new stripes - a new ShardLayout for the next frame.
    Stripes only move when the bottleneck would shrink by more than ``hysteresis`` (re-cutting costs the capacity
    estimates of the padded exchange one exact frame)."""

    def __init__(self, layout: ShardLayout, every: int = 1, hysteresis: float = 0.03):
        self.layout, self.every, self.hysteresis, self.frames = layout, max(1, int(every)), float(hysteresis), 0
        self.tby = _tile_bounds(layout.dims[1], layout.dims[0])[1]

    def stripes_from(self, work_all: Sequence[float]) -> List[int]:
        lay = self.layout
        new = balanced_stripes(work_all, lay.world)
        pre = [0.0]
        for v in work_all:
            pre.append(pre[-1] + float(v))
        cost = lambda st: max(pre[b] - pre[a] for a, b in zip(st, st[1:]))
        return new if cost(new) < (1.0 - self.hysteresis) * cost(lay.stripes) else list(lay.stripes)

    def update(self, my_rows: Sequence[float], group=None) -> ShardLayout:
        """my_rows: work of this rank's tile rows (len = its stripe's rows) -> the layout for the next frame"""
        lay = self.layout
        self.frames += 1
        if self.frames % self.every:
            return lay
        r0, r1 = lay.tile_rows
        if len(my_rows) != r1 - r0:
            raise ValueError("one work figure per tile row of this rank's stripe")
        mine = torch.zeros((self.tby,), dtype=torch.float64)
        mine[r0:r1] = torch.tensor(list(my_rows), dtype=torch.float64)
        if lay.world > 1:
            if dist.get_backend(group) == "gloo":
                dist.all_reduce(mine, group=group)
            else:
                dev = torch.device("cuda", torch.cuda.current_device())
                t = mine.to(dev)
                dist.all_reduce(t, group=group)
                mine = t.cpu()
        st = self.stripes_from(mine.tolist())
        if st != list(lay.stripes):
            self.layout = ShardLayout(lay.n_total, lay.world, lay.rank, lay.dims, st)
        return self.layout


def shard_model(model, world: int, rank: int) -> SplatModel:
    """Rank's rows of the six parameter tensors as a model of its own (fresh leaf tensors)."""
    i0, i1 = shard_range(model.means.shape[0], world, rank)
    ps = [p.detach()[i0:i1].clone() for p in model.parameters()]
    return SplatModel(*ps, active_sh_degree=model.active_sh_degree, background=model.background)


# --------------------------------------------------------------------------------------------------
# the two collectives of a frame
# --------------------------------------------------------------------------------------------------
class Exchange:
    """counts(): device int32[world] records per destination -> (send_counts, recv_counts) as Python lists.
    rows(): send[sum(send_counts), F] grouped by destination -> recv[sum(recv_counts), F] grouped by source.
    ``backward=True`` marks the gradient return (same shapes with the roles of the counts swapped)."""

    can_gather = False       # gather() exists: the padded exchange (module docstring of PADDED_EXCHANGE) may be used

    def counts(self, counts_dev: Tensor) -> Tuple[List[int], List[int]]:
        raise NotImplementedError

    def gather(self, counts_dev: Tensor) -> Tensor:
        """device int32[world] records per destination -> device int32[world, world]: row s = the counts of rank s.
        No host read."""
        raise NotImplementedError

    def rows(self, send: Tensor, send_counts: List[int], recv_counts: List[int], backward: bool = False,
             out: Optional[Tensor] = None) -> Tensor:
        """``out``: a [sum(recv_counts), F] buffer to receive into instead of a fresh tensor (the rank executor's)"""
        raise NotImplementedError


class DistExchange(Exchange):
    """torch.distributed: ``all_to_all_single`` with split sizes (RCCL over xGMI for backend "nccl").  gloo
    (functional tests with all ranks on one GPU) moves CUDA tensors through the host."""

    can_gather = True

    def __init__(self, group=None):
        self.group = group
        self.via_host = dist.get_backend(group) == "gloo"
        self.world = dist.get_world_size(group)

    def gather(self, counts_dev):
        if self.via_host:
            mine = counts_dev.cpu()
            got = [torch.empty_like(mine) for _ in range(self.world)]
            with collective_timer.span(on_device=False, label="all_gather(counts)", nbytes=mine.numel() * 4):
                dist.all_gather(got, mine, group=self.group)
            return torch.stack(got).to(counts_dev.device)
        out = torch.empty((self.world, counts_dev.shape[0]), dtype=counts_dev.dtype, device=counts_dev.device)
        with collective_timer.span(label="all_gather(counts)", nbytes=counts_dev.numel() * 4):
            dist.all_gather_into_tensor(out.view(-1), counts_dev, group=self.group)
        return out

    def counts(self, counts_dev):
        if self.via_host:
            mine = counts_dev.cpu()
            got = torch.empty_like(mine)
            with collective_timer.span(on_device=False, label="all_to_all(counts)", nbytes=mine.numel() * 4):
                dist.all_to_all_single(got, mine, group=self.group)
            return mine.tolist(), got.tolist()
        got = torch.empty_like(counts_dev)
        with collective_timer.span(label="all_to_all(counts)", nbytes=counts_dev.numel() * 4):
            dist.all_to_all_single(got, counts_dev, group=self.group)
        both = torch.stack([counts_dev, got]).cpu()              # the frame's host read of the record counts
        return both[0].tolist(), both[1].tolist()

    def rows(self, send, send_counts, recv_counts, backward=False, out=None):
        m = int(sum(recv_counts))
        if self.via_host:
            src = send.cpu()
            got = torch.empty((m, send.shape[1]), dtype=send.dtype)
            with collective_timer.span(on_device=False, label="all_to_all(gradient rows)" if backward else "all_to_all(records)",
                                       nbytes=int(sum(send_counts)) * send.shape[1] * send.element_size()):
                dist.all_to_all_single(got, src, output_split_sizes=list(recv_counts),
                                       input_split_sizes=list(send_counts), group=self.group)
            if out is not None:
                out.copy_(got)
                return out
            return got.to(send.device)
        got = send.new_empty((m, send.shape[1])) if out is None else out
        with collective_timer.span(label="all_to_all(gradient rows)" if backward else "all_to_all(records)",
                                   nbytes=int(sum(send_counts)) * send.shape[1] * send.element_size()):
            dist.all_to_all_single(got, send, output_split_sizes=list(recv_counts),
                                   input_split_sizes=list(send_counts), group=self.group)
        return got


class ReplayExchange(Exchange):
    """bench.py --emulate-ranks: one rank's step on one GPU.  The records the other ranks would send were
    produced once (``remote_records``); every frame the rank's own group is copied into its place, and in
    backward only the rows of the rank's own records come back (the others would travel to their owners).
    No byte crosses a link: what is timed is the rank's compute, NOT a multi-GPU measurement."""

    can_gather = True

    def __init__(self, rank: int, recv_counts: List[int], recv_template: Tensor):
        self.rank, self.recv_counts, self.template = rank, list(recv_counts), recv_template
        self.recv_off = [0]
        for c in self.recv_counts:
            self.recv_off.append(self.recv_off[-1] + c)
        self.send_counts = None
        # the count matrix as far as this rank can know it: column `rank` = what the others send here
        world = len(self.recv_counts)
        mat = torch.zeros((world, world), dtype=torch.int32)
        mat[:, rank] = torch.tensor(self.recv_counts, dtype=torch.int32)
        self.mat = mat.to(recv_template.device)
        self._padded = (None, None)              # (split sizes, the template laid out for them)

    def counts(self, counts_dev):
        self.send_counts = counts_dev.cpu().tolist()
        if self.send_counts[self.rank] != self.recv_counts[self.rank]:
            raise RuntimeError("the replayed records do not belong to this scene / camera")
        return self.send_counts, self.recv_counts

    def gather(self, counts_dev):
        mat = self.mat.clone()
        mat[self.rank] = counts_dev
        return mat

    def _template_for(self, splits):
        """the remote groups at the offsets the (padded) split sizes give, zero rows in between"""
        splits = tuple(int(c) for c in splits)
        if splits == tuple(self.recv_counts):
            return self.template
        if self._padded[0] != splits:
            t = self.template.new_zeros((sum(splits), self.template.shape[1]))
            off = 0
            for k, c in enumerate(splits):
                n_k = min(self.recv_counts[k], c)     # (a group that outgrew its capacity arrives cut off, as from
                #                                        ts_route_pack: that frame is dropped and run again)
                t[off:off + n_k] = self.template[self.recv_off[k]:self.recv_off[k] + n_k]
                off += c
            self._padded = (splits, t)
        return self._padded[1]

    def rows(self, send, send_counts, recv_counts, backward=False, out=None):
        so = sum(send_counts[:self.rank])
        ro = sum(recv_counts[:self.rank])
        k = send_counts[self.rank]
        if backward:        # send = gradient rows of the imported records; own rows return, the rest is remote
            got = send.new_zeros((sum(recv_counts), send.shape[1])) if out is None else out.zero_()
            got[ro:ro + k] = send[so:so + k]
            return got
        got = self._template_for(recv_counts).clone() if out is None else out.copy_(self._template_for(recv_counts))
        got[ro:ro + k] = send[so:so + k]
        return got


# --------------------------------------------------------------------------------------------------
# the frame
# --------------------------------------------------------------------------------------------------
class _Owner:
    __slots__ = ("n", "nb", "ch", "cam", "fr", "ws", "xys", "radii", "send_counts", "recv_counts", "inputs",
                 "p_route_ws", "caps", "key", "count_host", "count_event", "counts")


class _Stripe:
    __slots__ = ("m", "cam", "fr", "ws", "split", "segs", "mode", "bucket", "total", "num_tiles", "offs",
                 "bg", "records")


class _LazyBinning:
    """frame.last_binning entry of a stripe: the views into the stripe's workspace (scene statistics for bench.py /
    tools) are made when somebody asks for them, not on every frame of a training loop."""
    __slots__ = ("cam", "n", "num_tiles", "num_intersects", "_ws", "_offs", "_bucket", "_cap")

    @property
    def tile_bins(self):
        nt = max(self.num_tiles, 1)
        return _view(self._ws, self._offs[8], torch.int32, 2 * nt, (nt, 2))[:self.num_tiles]

    @property
    def gaussian_ids_sorted(self):
        return self._bucket[self._cap:self._cap + self.num_intersects]

    @property
    def cum_tiles_hit(self):
        return _view(self._ws, self._offs[4], torch.int32, self.n, (self.n,))

    @property
    def num_tiles_hit(self):
        return _view(self._ws, self._offs[3], torch.int32, self.n, (self.n,))


def _carve(dev, sizes):
    """ONE allocation for a list of byte sizes (256-byte aligned sections) -> (tensor, section pointers, offsets)."""
    offs, off = [], 0
    for sz in sizes:
        offs.append(off)
        off += (int(sz) + 255) & ~255
    ws = torch.empty((max(off, 256),), dtype=torch.uint8, device=dev)
    base = ws.data_ptr()
    return ws, [base + o for o in offs], offs


def _view(ws, off, dtype, count, shape):
    return ws[off:off + count * dtype.itemsize].view(dtype).view(shape)


def _layout_key(dev, layout: ShardLayout):
    return (dev.index, layout.world, layout.rank, layout.dims, layout.n_total, tuple(layout.stripes))


def _owner_stage(lib, s, dev, layout: ShardLayout, exchange: Exchange, means, scales, quats, opacities,
                 colors_dc, colors_rest, view34, projview, origin, fx, fy, sh_degree, ch, keep: bool,
                 padded: bool = True):
    w, h = layout.dims
    n = means.shape[0]
    nb = colors_rest.shape[1] + 1
    if sh_degree < 0 or sh_degree > deg_from_sh(nb):
        raise ValueError("sh_degree exceeds the stored coefficients")
    O = _Owner()
    O.n, O.nb, O.ch = n, nb, ch
    O.cam = _camera(fx, fy, w / 2, h / 2, h, w, _tile_bounds(h, w), 1.0)
    m = max(n, 1)
    n_route = int(lib.ts_route_ws_ints(n, layout.world))
    #   xys | depths | radii | conics | nth | splats | sh_mask | route_ws | counts
    O.ws, ptr, offs = _carve(dev, [8 * m, 4 * m, 4 * m, 12 * m, 4 * m, 48 * m, m, 4 * n_route, 4 * layout.world])
    O.xys = _view(O.ws, offs[0], torch.float32, 2 * n, (n, 2))
    O.radii = _view(O.ws, offs[2], torch.int32, n, (n,))
    counts = _view(O.ws, offs[8], torch.int32, layout.world, (layout.world,))
    O.counts = counts
    O.p_route_ws = ptr[7]
    fr = TsFrame()
    fr.n, fr.num_bases, fr.sh_degree, fr.channels, fr.flags = n, nb, int(sh_degree), ch, 0
    fr.cam, fr.capacity = O.cam, -1
    fr.means, fr.scales, fr.quats, fr.opacities = means.data_ptr(), scales.data_ptr(), quats.data_ptr(), opacities.data_ptr()
    fr.colors_dc, fr.colors_rest = colors_dc.data_ptr(), colors_rest.data_ptr()
    fr.view34, fr.projview, fr.origin = view34.data_ptr(), projview.data_ptr(), origin.data_ptr()
    fr.xys, fr.depths, fr.radii, fr.conics, fr.num_tiles_hit, fr.splats = ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5]
    fr.sh_mask = ptr[6] if keep else None
    O.fr = fr
    # padded exchange (see PADDED_EXCHANGE): capacities from the previous frame of this layout, if there was one
    O.key = _layout_key(dev, layout)
    use_gather = padded and PADDED_EXCHANGE and exchange.can_gather and _comm.gather_usable
    O.caps = _route_caps.get(O.key) if use_gather else None
    O.count_host = O.count_event = None
    me, world = layout.rank, layout.world
    gb = None
    if O.caps is not None:
        base = [0]
        for c in O.caps[me].tolist():
            base.append(base[-1] + c)
        gb = (ctypes.c_int32 * (world + 1))(*base)
    if kernel_timer.enabled and 0 < n <= SMALL_N_FUSED:      # (what ts_shard_owner_fwd_padded issues for a small shard)
        _call("ts_owner_fwd_fused", lib.ts_shard_owner_fwd_fused, n, int(sh_degree), nb, fr.means, fr.scales, fr.quats,
              fr.view34, fr.projview, O.cam, 3, fr.origin, fr.colors_dc, fr.colors_rest if nb > 1 else None, fr.opacities,
              ch, 1, fr.xys, fr.depths, fr.radii, fr.conics, fr.num_tiles_hit, fr.sh_mask, fr.splats, layout.c_stripes,
              gb, O.p_route_ws, counts.data_ptr(), s)
    elif kernel_timer.enabled:
        _call("ts_project_fwd", lib.ts_project_fwd, n, fr.means, fr.scales, fr.quats, fr.view34, fr.projview, O.cam, 3,
              fr.xys, fr.depths, fr.radii, fr.conics, fr.num_tiles_hit, None, s)
        # colour stage + packed records of the owned Gaussians (slot fields are rewritten by the importing rank)
        _call("ts_colors_pack_fwd", lib.ts_colors_pack_fwd, n, int(sh_degree), nb, fr.means, fr.origin, fr.colors_dc,
              fr.colors_rest if nb > 1 else None, fr.sh_mask, None, ch, 1, fr.xys, fr.radii, fr.conics,
              fr.opacities, fr.num_tiles_hit, O.cam, fr.depths if ch == 4 else None, fr.splats, s)
        _call("ts_route_count", lib.ts_route_count_padded, n, fr.xys, fr.radii, O.cam, layout.c_stripes, gb, O.p_route_ws,
              counts.data_ptr(), s)
    else:
        _lib.check(lib.ts_shard_owner_fwd_padded(ctypes.byref(fr), layout.c_stripes, gb, O.p_route_ws,
                                                 counts.data_ptr(), s), "ts_shard_owner_fwd")
    if O.caps is not None:
        # nothing of this frame's counts is looked at here: they are gathered, copied to the host behind the launches
        # already enqueued, and checked where the frame waits for the stripe's pair count (_ShardedFrame.forward)
        mat = exchange.gather(counts)
        slot = _count_slots.get((dev.index, world))
        if slot is None:
            slot = _count_slots[(dev.index, world)] = torch.empty((world * world,), dtype=torch.int32).pin_memory()
        slot.copy_(mat.view(-1), non_blocking=True)
        O.count_host = slot
        O.count_event = torch.cuda.Event()
        O.count_event.record(torch.cuda.current_stream(dev))
        O.send_counts = O.caps[me].tolist()
        O.recv_counts = O.caps[:, me].tolist()
        total = sum(O.send_counts)
        send = torch.zeros((max(total, 1), RECORD_FLOATS), dtype=torch.float32, device=dev)[:total]
    else:
        if use_gather:
            host = exchange.gather(counts).cpu()                 # the frame's host read of the record counts
            O.send_counts, O.recv_counts = host[me].tolist(), host[:, me].tolist()
            _route_caps[O.key] = _capacity(host.numpy())
        else:
            O.send_counts, O.recv_counts = exchange.counts(counts)
        total = sum(O.send_counts)
        send = torch.empty((total, RECORD_FLOATS), dtype=torch.float32, device=dev) if total > 0 else \
            torch.empty((1, RECORD_FLOATS), dtype=torch.float32, device=dev)[:0]
    _call("ts_route_pack", lib.ts_route_pack, n, layout.owned[0], fr.xys, fr.radii, fr.depths, fr.splats,
          O.cam, layout.c_stripes, O.p_route_ws, send.data_ptr(), s)
    return O, send


def _stripe_stage(lib, s, dev, layout: ShardLayout, records: Tensor, background: Tensor, fx, fy, ch, keep: bool):
    w, h = layout.dims
    S = _Stripe()
    m = records.shape[0]
    S.m, S.records = m, records
    cam = _camera(fx, fy, w / 2, h / 2, h, w, _tile_bounds(h, w), 1.0, tile_rows=layout.tile_rows)
    S.mode = _frame._list_mode(dev.index, cam.tile_rows * cam.tile_bounds_x)
    cam.wide_tiles = 1 if S.mode else 0
    S.cam = cam
    S.split = 0 < cam.tile_rows * cam.tile_bounds_x <= _frame.SPLIT_BLOCKS_BELOW
    S.segs, w16 = _frame._list_segments(cam.tile_rows * cam.tile_bounds_x, S.mode, S.split) if keep else (1, 0)
    _frame.set_launch_hints(cam, S.segs, w16, S.mode, S.split)
    fin_floats = int(lib.ts_final_floats(ctypes.byref(cam), ch))
    num_tiles = int(lib.ts_num_tiles(ctypes.byref(cam)))
    S.num_tiles = num_tiles
    rows = _stripe_rows(cam)
    i32 = dict(dtype=torch.int32, device=dev)
    mm = max(m, 1)
    px = rows * w
    nscan, nbin = int(lib.ts_scan_ws_ints(m)), int(lib.ts_bin_ws_ints(m, num_tiles))
    #   xys | depths | radii | nth | cum | splats | scan_ws | bin_ws | tile_bins | final_Ts | final_index | clamp_mask
    S.ws, ptr, offs = _carve(dev, [8 * mm, 4 * mm, 4 * mm, 4 * mm, 4 * mm, 48 * mm, 4 * nscan, 4 * nbin,
                                   8 * max(num_tiles, 1)] + ([4 * fin_floats, 4 * px, px] if keep else []))
    S.offs = offs
    out_img = torch.empty((rows, w, ch), dtype=torch.float32, device=dev)
    if ch == 4:            # channel 3 is composited over background[0], as the reference's depth pass (:86)
        S.bg = _f32c(torch.cat([background, background[:1]]))
    else:
        S.bg = _f32c(background)
    host, event, lock = _frame._total_slot(dev)
    fr = TsFrame()
    fr.n, fr.num_bases, fr.sh_degree, fr.channels = m, 1, 0, ch
    fr.flags = ((1 if _frame.TIGHT_BINNING else 0) | (2 if S.split else 0) | (8 if S.mode == 2 else 0)
                | (0 if _frame.TWO_HOP_SCATTER else 32))
    fr.cam, fr.capacity = cam, -1
    fr.background = S.bg.data_ptr()
    fr.xys, fr.depths, fr.radii, fr.num_tiles_hit, fr.cum_tiles_hit, fr.splats = ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5]
    fr.scan_ws, fr.bin_ws, fr.tile_bins = ptr[6], ptr[7], ptr[8]
    fr.total_host = host.data_ptr()
    fr.out_img = out_img.data_ptr()
    if keep:
        fr.final_Ts, fr.final_index, fr.clamp_mask = ptr[9], ptr[10], ptr[11]
    S.fr = fr
    timed = kernel_timer.enabled
    tight = fr.splats if _frame.TIGHT_BINNING else None
    cap_key = (dev.index, "stripe", layout.world, w, h, cam.tile_row0, cam.tile_rows)
    est = _frame._capacity.get(cap_key) if (_frame.CAPACITY_ALLOC and m > 0) else None

    def lists(size):
        cap = (max(int(size), 1) + 63) & ~63
        S.bucket = torch.empty((2 * cap,), **i32)
        fr.bucket_ids, fr.gaussian_ids_sorted = S.bucket.data_ptr(), S.bucket.data_ptr() + 4 * cap
        return cap

    def stage_import():
        if not timed:
            _lib.check(lib.ts_shard_stripe_fwd_import(ctypes.byref(fr), records.data_ptr() if m > 0 else None, s),
                       "ts_shard_stripe_fwd_import")
            return
        if m > 0:
            _call("ts_import_records", lib.ts_import_records, m, records.data_ptr(), cam, fr.xys, fr.depths, fr.radii,
                  fr.num_tiles_hit, s)
            _call("ts_scan_tiles", lib.ts_scan_tiles, m, fr.num_tiles_hit, fr.cum_tiles_hit, fr.scan_ws, None, s)
            host.copy_(_view(S.ws, offs[4], torch.int32, m, (m,))[m - 1:m], non_blocking=True)
            _call("ts_import_pack", lib.ts_import_pack, m, records.data_ptr(), fr.cum_tiles_hit, cam, fr.splats, s)
        _call("ts_bin_count", lib.ts_bin_count, m, fr.xys, fr.radii, tight, cam, fr.bin_ws, s)
        _call("ts_tile_offsets", lib.ts_tile_offsets, m, num_tiles, fr.bin_ws, fr.tile_bins, fr.cum_tiles_hit,
              fr.capacity, s)

    def composite():
        if timed:
            _frame._steps_composite(lib, fr, s)
        else:
            _lib.check(lib.ts_frame_fwd_composite(ctypes.byref(fr), s), "ts_frame_fwd_composite")

    cap = None
    spin = _frame.COUNT_WAIT == "spin" and m > 0 and not timed
    with lock:
        if spin:
            word = ctypes.c_int32.from_address(host.data_ptr())
            word.value = -(1 << 31)
        if est is not None:          # sized by the previous frame's count: everything is enqueued before the read
            cap = lists(est)
            fr.capacity, fr.num_intersects = cap, cap
        stage_import()
        event.record(torch.cuda.current_stream(dev))
        if est is not None:
            composite()
        total = 0
        if m > 0:                                        # the stripe's intersection count (sizes the lists)
            if spin:
                k = 0
                while word.value == -(1 << 31):
                    k += 1
                    if k > _frame._SPIN_LIMIT:
                        event.synchronize()
                        break
                total = int(word.value)
            else:
                event.synchronize()
                total = int(host[0])
    if total < 0:
        raise OverflowError("more than 2^31-1 tile intersections in one stripe")
    S.total = total
    _frame._pairs_per_tile[dev.index] = total / max(1, cam.tile_rows * cam.tile_bounds_x)
    if m > 0 and _frame.CAPACITY_ALLOC:
        _frame._capacity[cap_key] = int(total * _frame.CAPACITY_GROWTH) + 4096
    if cap is None or total > cap:
        redo = cap is not None
        cap = lists(total)
        fr.capacity, fr.num_intersects = -1, total
        S.segs = _frame.segments_for_count(fr.cam, S.segs, total)      # short lists: no boundary records are kept
        if redo:
            stage_import()
        composite()
    else:
        fr.num_intersects = total
    b = _LazyBinning()                       # scene statistics of the most recent frame (bench.py / tools)
    b.cam, b.n, b.num_tiles, b.num_intersects = cam, m, num_tiles, total
    b._ws, b._offs, b._bucket, b._cap = S.ws, offs, S.bucket, cap
    _frame.last_binning[dev.index] = b
    return S, out_img


def _stripe_backward(lib, s, dev, S: _Stripe, ch: int, v_img: Tensor) -> Tensor:
    """Compositing backward of the stripe -> one gradient row per imported record [m, 12]."""
    f32 = dict(dtype=torch.float32, device=dev)
    m, fr = S.m, S.fr
    S.segs = _frame.backward_segments(fr.cam, S.segs, S.total, dev.index)
    S.cam.hints = fr.cam.hints
    bwd_split = S.split and S.segs <= 1              # list segments replace the split blocks in this pass
    rows_n = max(S.total, 1) * (4 if bwd_split else 1)
    partials = torch.empty((rows_n, _lib.PARTIAL_ROW_FLOATS), **f32)
    row_flags, fr.flag_gen = _frame.row_flags_for(dev, rows_n)
    grad_rows = torch.empty((m, ROW_FLOATS), **f32) if m > 0 else torch.empty((1, ROW_FLOATS), **f32)[:0]
    fr.v_out_img, fr.partials, fr.row_flags = v_img.data_ptr(), partials.data_ptr(), row_flags.data_ptr()
    if kernel_timer.enabled:
        gen = (fr.flag_gen & 0xff) << 8
        rflags = (4 if bwd_split else 0) | (8 if S.mode == 2 else 0) | gen
        _call("ts_raster_bwd", lib.ts_raster_bwd, ch, rflags, S.total, S.cam, fr.tile_bins, fr.gaussian_ids_sorted,
              fr.splats, fr.background, fr.final_Ts, fr.final_index, fr.v_out_img, None, fr.clamp_mask, fr.partials,
              fr.row_flags, s)
        _call("ts_reduce_partials_rows", lib.ts_reduce_partials_rows, m, ch, (4 if bwd_split else 0) | gen, fr.num_tiles_hit,
              fr.cum_tiles_hit, fr.partials, fr.row_flags, fr.splats, grad_rows.data_ptr(), s)
    else:
        _lib.check(lib.ts_shard_stripe_bwd(ctypes.byref(fr), grad_rows.data_ptr(), s), "ts_shard_stripe_bwd")
    S.records = None
    return grad_rows


def _owner_backward(lib, s, dev, layout: ShardLayout, O: _Owner, back: Tensor, sh_degree: int, opacity_shape,
                    rest_shape):
    """Rows returned by the destinations -> gradients of the six OWNED parameter tensors and v_xy."""
    f32 = dict(dtype=torch.float32, device=dev)
    n, ch, fr = O.n, O.ch, O.fr
    nn = max(n, 1)
    def rows_of(*tail):               # [n, *tail] (a slice of one row when there are none: the pointers stay valid)
        return torch.empty((n,) + tail, **f32) if n > 0 else torch.empty((1,) + tail, **f32)[:0]
    v_xy = rows_of(2)
    v_opac = rows_of(*opacity_shape[1:])
    v_means = rows_of(3)
    v_scales = rows_of(3)
    v_quats = rows_of(4)
    v_dc = rows_of(3)
    v_rest = rows_of(*rest_shape[1:])
    tmp = torch.empty((nn * 7,), **f32)                      # v_conic | v_colors | v_depth
    fr.v_xy, fr.v_opacity = v_xy.data_ptr(), v_opac.data_ptr()
    fr.v_conic, fr.v_colors, fr.v_depth = tmp.data_ptr(), tmp.data_ptr() + 12 * nn, tmp.data_ptr() + 24 * nn
    fr.v_means, fr.v_scales, fr.v_quats = v_means.data_ptr(), v_scales.data_ptr(), v_quats.data_ptr()
    fr.v_colors_dc, fr.v_colors_rest = v_dc.data_ptr(), v_rest.data_ptr()
    if kernel_timer.enabled and 0 < n <= SMALL_N_FUSED:      # (what ts_shard_owner_bwd issues for a small shard)
        _call("ts_owner_bwd_fused", lib.ts_shard_owner_bwd_fused, n, ch, sh_degree, O.nb, fr.means, fr.scales, fr.quats,
              fr.view34, fr.projview, fr.origin, fr.xys, fr.radii, fr.splats, fr.sh_mask, O.cam, layout.c_stripes,
              O.p_route_ws, back.data_ptr(), fr.v_xy, fr.v_conic, fr.v_colors, fr.v_depth if ch == 4 else None,
              fr.v_opacity, fr.v_colors_dc, fr.v_colors_rest if O.nb > 1 else None, fr.v_means, fr.v_scales, fr.v_quats, s)
    elif kernel_timer.enabled:
        _call("ts_route_accumulate", lib.ts_route_accumulate, n, ch, fr.xys, fr.radii, fr.splats, fr.sh_mask, O.cam,
              layout.c_stripes, O.p_route_ws, back.data_ptr(), fr.v_xy, fr.v_conic, fr.v_colors,
              fr.v_depth if ch == 4 else None, fr.v_opacity, s)
        _call("ts_sh_colors_bwd", lib.ts_sh_colors_bwd, n, sh_degree, O.nb, fr.means, fr.origin, None, fr.v_colors,
              fr.v_colors_dc, fr.v_colors_rest if O.nb > 1 else None, s)
        _call("ts_project_bwd", lib.ts_project_bwd, n, fr.means, fr.scales, fr.quats, fr.view34, fr.projview, O.cam, 3,
              fr.radii, fr.v_xy, fr.v_depth if ch == 4 else None, fr.v_conic, None, fr.v_means, fr.v_scales,
              fr.v_quats, s)
    else:
        _lib.check(lib.ts_shard_owner_bwd(ctypes.byref(fr), layout.c_stripes, O.p_route_ws, back.data_ptr(), s),
                   "ts_shard_owner_bwd")
    return (v_means, v_scales, v_quats, v_opac, v_dc, v_rest), v_xy


def _inputs(model, view, projview, origin):
    return tuple(_f32c(t) for t in (model.means, model.scales, model.quats, model.opacities, model.colors_dc,
                                    model.colors_rest, view[:3, :], projview, origin))


# --------------------------------------------------------------------------------------------------
# RANK EXECUTOR: the step of a rank with nothing rebuilt per frame
# --------------------------------------------------------------------------------------------------
# The stage functions above describe a frame from scratch every time: two cameras, two ts_frame structs, two carved
# workspaces, a dozen tensors, two host reads (the record counts in the middle of the forward pass, the stripe's pair
# count) - ~0.25 ms of Python in forward and ~0.19 ms in backward for 0.37 ms of kernels on one rank of 8 of config 3
# (tools/host_profile_rank.py).  From the SECOND frame of a layout on, the step runs from a _RankState instead:
#   * everything whose size follows the layout lives across frames (owner workspace, structs, cameras, route buffers);
#   * what follows the DATA is sized by CAPACITIES derived from the previous frame - the groups of the record exchange
#     (the padded exchange above: zeroed send buffer, ts_route_count_padded, capacities as split sizes, the counts
#     all_gathered and looked at once the whole forward pass is enqueued) and the stripe's lists (ts_tile_offsets'
#     capacity guard) - so no count is waited for in the middle of a frame;
#   * the launches are issued by four native calls split only at the two collectives (ts_shard_rank_fwd_a / _fwd_b /
#     _bwd_a / _bwd_b, csrc/frame.hip).
# A frame whose counts outgrow a capacity (every rank sees the same count matrix) or whose lists outgrow theirs is run
# again through the stage functions with exact sizes, which also renews the capacities; so is the first frame of a
# layout, a frame that starts while the previous one still waits for its backward pass, and every frame while
# ops.kernel_timer records.  Same kernels, same order, same buffers' contents: results are bitwise those of the stage
# functions (tests/test_gpu_sharded.py).  TS_RANK_EXECUTOR=0 switches it off.
RANK_EXECUTOR = os.environ.get("TS_RANK_EXECUTOR", "1") != "0"
_rank_states = {}        # (layout key, n, nb, ch, dims) -> _RankState
rank_executor_frames = [0, 0, 0]   # frames run from the state / of those repeated with exact sizes / frames via the stage functions


class _RankState:
    __slots__ = ("key", "dev", "layout", "n", "nb", "ch", "o_cam", "owner_ws", "o_offs", "fo", "R", "xys", "radii",
                 "counts", "caps", "send_caps", "recv_caps", "send", "recv", "grad_rows", "back", "list_cap",
                 "s_cam", "hints0", "split", "segs", "mode", "num_tiles", "rows", "fin_floats", "stripe_ws", "s_offs", "fs",
                 "bucket", "partials", "busy", "count_host", "count_event", "total", "bg", "bg_key", "m", "tiles16", "fxy",
                 "frames", "count_np", "frame_id", "token", "last_used")

    def __init__(self, lib, dev, layout: ShardLayout, n: int, nb: int, ch: int, fx, fy):
        w, h = layout.dims
        self.dev, self.layout, self.n, self.nb, self.ch = dev, layout, n, nb, ch
        self.busy, self.caps, self.list_cap, self.partials, self.bg_key, self.frames = False, None, 0, None, None, 0
        self.frame_id, self.token, self.last_used = 0, None, 0
        world = layout.world
        # ---- owner side (sizes follow the layout only)
        self.o_cam = _camera(fx, fy, w / 2, h / 2, h, w, _tile_bounds(h, w), 1.0)
        m = max(n, 1)
        n_route = int(lib.ts_route_ws_ints(n, world))
        self.owner_ws, ptr, offs = _carve(dev, [8 * m, 4 * m, 4 * m, 12 * m, 4 * m, 48 * m, m, 4 * n_route, 4 * world])
        self.o_offs = offs
        self.xys = _view(self.owner_ws, offs[0], torch.float32, 2 * n, (n, 2))
        self.radii = _view(self.owner_ws, offs[2], torch.int32, n, (n,))
        self.counts = _view(self.owner_ws, offs[8], torch.int32, world, (world,))
        fo = TsFrame()
        fo.n, fo.num_bases, fo.channels, fo.flags = n, nb, ch, 0
        fo.cam, fo.capacity = self.o_cam, -1
        fo.xys, fo.depths, fo.radii, fo.conics, fo.num_tiles_hit, fo.splats = ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5]
        fo.sh_mask = ptr[6]
        self.fo = fo
        R = _lib.TsRankStep()
        R.stripes = layout.c_stripes
        R.gid_base = layout.owned[0]
        R.route_ws, R.counts = ptr[7], ptr[8]
        self.R = R
        # ---- stripe side: made by size() (its buffers follow the data, its list shape the previous frame)
        self.tiles16 = (layout.tile_rows[1] - layout.tile_rows[0]) * _tile_bounds(h, w)[0]
        self.fxy = (fx, fy)
        self.mode = None
        self.count_host = torch.empty((world * world,), dtype=torch.int32).pin_memory()
        self.count_np = self.count_host.numpy().reshape(world, world)        # (a view of the pinned words)
        self.count_event = torch.cuda.Event()

    def size(self, lib, caps, list_cap: int):
        """buffers for the capacities ``caps`` (the [world, world] matrix every rank derives alike) and ``list_cap``"""
        dev, lay, ch = self.dev, self.layout, self.ch
        me, world = lay.rank, lay.world
        w, h = lay.dims
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        # the stripe's list shape and launch mapping, decided as _stripe_stage decides them
        self.mode = _frame._list_mode(dev.index, self.tiles16)
        cam = _camera(self.fxy[0], self.fxy[1], w / 2, h / 2, h, w, _tile_bounds(h, w), 1.0, tile_rows=lay.tile_rows)
        cam.wide_tiles = 1 if self.mode else 0
        self.split = 0 < cam.tile_rows * cam.tile_bounds_x <= _frame.SPLIT_BLOCKS_BELOW
        self.segs, w16 = _frame._list_segments(cam.tile_rows * cam.tile_bounds_x, self.mode, self.split)
        _frame.set_launch_hints(cam, self.segs, w16, self.mode, self.split)
        self.hints0 = cam.hints
        self.s_cam = cam
        self.fin_floats = int(lib.ts_final_floats(ctypes.byref(cam), ch))
        self.num_tiles = int(lib.ts_num_tiles(ctypes.byref(cam)))
        self.rows = _stripe_rows(cam)
        self.bg_key = None
        self.caps = caps
        self.send_caps, self.recv_caps = [int(c) for c in caps[me]], [int(c) for c in caps[:, me]]
        send_rows, m = sum(self.send_caps), sum(self.recv_caps)
        self.m = m
        self.send = torch.empty((send_rows, RECORD_FLOATS), **f32)
        self.recv = torch.empty((m, RECORD_FLOATS), **f32)
        self.grad_rows = torch.empty((m, ROW_FLOATS), **f32)
        self.back = torch.empty((send_rows, ROW_FLOATS), **f32)
        R = self.R
        base = 0
        for d in range(world):
            R.group_base[d] = base
            base += self.send_caps[d]
        R.group_base[world] = base
        R.send_rows, R.recv_rows = send_rows, m
        R.send, R.recv, R.grad_rows, R.back = (self.send.data_ptr(), self.recv.data_ptr(), self.grad_rows.data_ptr(),
                                               self.back.data_ptr())
        # the stripe's workspace, as _stripe_stage lays it out (m = the capacity: padding rows are all-zero records)
        px = self.rows * w
        nscan, nbin = int(lib.ts_scan_ws_ints(m)), int(lib.ts_bin_ws_ints(m, self.num_tiles))
        self.stripe_ws, ptr, offs = _carve(dev, [8 * m, 4 * m, 4 * m, 4 * m, 4 * m, 48 * m, 4 * nscan, 4 * nbin,
                                                 8 * max(self.num_tiles, 1), 4 * self.fin_floats, 4 * px, px])
        self.s_offs = offs
        self.list_cap = (max(int(list_cap), 1) + 63) & ~63
        self.bucket = torch.empty((2 * self.list_cap,), **i32)
        fs = TsFrame()
        fs.n, fs.num_bases, fs.sh_degree, fs.channels = m, 1, 0, ch
        fs.flags = ((1 if _frame.TIGHT_BINNING else 0) | (2 if self.split else 0) | (8 if self.mode == 2 else 0)
                    | (0 if _frame.TWO_HOP_SCATTER else 32))
        fs.cam = self.s_cam
        fs.xys, fs.depths, fs.radii, fs.num_tiles_hit, fs.cum_tiles_hit, fs.splats = ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5]
        fs.scan_ws, fs.bin_ws, fs.tile_bins = ptr[6], ptr[7], ptr[8]
        fs.final_Ts, fs.final_index, fs.clamp_mask = ptr[9], ptr[10], ptr[11]
        fs.total_host = _frame._total_slot(dev)[0].data_ptr()
        fs.bucket_ids, fs.gaussian_ids_sorted = self.bucket.data_ptr(), self.bucket.data_ptr() + 4 * self.list_cap
        fs.capacity, fs.num_intersects = self.list_cap, self.list_cap
        self.fs = fs
        self.partials = None

    def renew(self, lib, mat, total: int):
        """Capacities for the next frame from this frame's counts, kept while they still fit with room to spare (a
        change re-allocates the buffers).  The groups' capacities are a function of the count matrix and the previous
        capacities alone - identical on every rank -, the lists' capacity and their shape are this rank's own."""
        caps = self.caps
        self.frames += 1
        shrink = self.frames % 32 == 0                # (looked at now and then: capacities that have become far too large)
        if caps is None or int((caps - mat).min()) < 0 or (shrink and not bool((_capacity(mat) * 2 >= caps).all())):
            caps = _capacity(mat)
        need = int(total * _frame.CAPACITY_GROWTH) + 4096
        list_cap = self.list_cap if (total <= self.list_cap and (not shrink or self.list_cap <= 2 * need)) else need
        if caps is not self.caps or list_cap != self.list_cap or self.mode != _frame._list_mode(self.dev.index, self.tiles16):
            self.size(lib, caps, list_cap)


class _FrameToken:
    """Lives on the autograd ctx of the frame that holds a _RankState: when the graph is freed - with or without a
    backward pass - the token dies with it, and the state sees that through its weak reference (`busy` was
    cleared by backward only, so a grad-enabled forward whose backward never came left the state busy for good)."""
    __slots__ = ("__weakref__",)


_rank_clock = [0]        # frames started through _rank_state (any layout): the age of a state = clock - last_used
RANK_STATE_MAX_AGE = 64  # a busy state that no frame has touched for this long is dropped by the purge with the others


def _state_in_use(st: "_RankState") -> bool:
    """busy AND its frame's graph is still alive; a state whose frame was dropped without a backward pass is reclaimed"""
    if st.busy and (st.token is None or st.token() is None):
        st.busy, st.token = False, None
    return st.busy


def _rank_state(lib, dev, layout: ShardLayout, n, nb, ch, fx, fy, stream_handle) -> Optional["_RankState"]:
    key = (layout.key, dev.index, n, nb, ch, float(fx), float(fy), stream_handle)
    _rank_clock[0] += 1
    st = _rank_states.get(key)
    if st is None:
        if len(_rank_states) > 16:                    # layouts come and go with densification: no unbounded growth
            for k in [k for k, v in _rank_states.items()
                      if not _state_in_use(v) or _rank_clock[0] - v.last_used > RANK_STATE_MAX_AGE]:
                # (a dropped state that a retained graph still points at keeps its buffers alive through that graph;
                # its backward pass still finds them untouched - no later frame can reach the state any more)
                del _rank_states[k]
        st = _rank_states[key] = _RankState(lib, dev, layout, n, nb, ch, fx, fy)
        st.key = key
    st.last_used = _rank_clock[0]
    return st


def _fast_forward(lib, cur, dev, st: "_RankState", exchange: Exchange, inputs, background, sh_degree: int):
    """The forward pass from a sized _RankState -> (out image, total) or None (a capacity was outgrown: the caller
    runs the frame again through the stage functions, which renews the capacities)."""
    _frame._mark("rank fwd: enter")
    s = cur.cuda_stream                               # (cur: torch's current stream on dev, looked up once per pass)
    lay, ch = st.layout, st.ch
    world, me = lay.world, lay.rank
    fo, fs, R = st.fo, st.fs, st.R
    means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin = inputs
    fo.sh_degree = int(sh_degree)
    fo.means, fo.scales, fo.quats, fo.opacities = means.data_ptr(), scales.data_ptr(), quats.data_ptr(), opacities.data_ptr()
    fo.colors_dc, fo.colors_rest = colors_dc.data_ptr(), colors_rest.data_ptr()
    fo.view34, fo.projview, fo.origin = view34.data_ptr(), projview.data_ptr(), origin.data_ptr()
    bkey = (background.data_ptr(), background._version)
    if st.bg_key != bkey:
        st.bg = _f32c(torch.cat([background, background[:1]])) if ch == 4 else _f32c(background)
        st.bg_key = bkey
        fs.background = st.bg.data_ptr()
    fs.cam.hints = st.hints0                          # (the backward pass may have cleared the segment bits)
    out_img = torch.empty((st.rows, lay.dims[0], ch), dtype=torch.float32, device=dev)
    fs.out_img = out_img.data_ptr()
    # xys / radii are handed to the caller (extras['xys'], whose .grad the backward pass sets): a frame's own tensors
    n = st.n
    xr = torch.empty((max(3 * n, 1),), dtype=torch.float32, device=dev)
    st.xys, st.radii = xr[:2 * n].view(n, 2), xr[2 * n:3 * n].view(torch.int32)
    fo.xys, fo.radii = xr.data_ptr(), xr.data_ptr() + 8 * n
    host, event, lock = _frame._total_slot(dev)
    with lock:
        word = ctypes.c_int32.from_address(host.data_ptr())
        word.value = -(1 << 31)
        _frame._mark("rank fwd: structs set, image allocated")
        _lib.check(lib.ts_shard_rank_fwd_a(ctypes.byref(fo), ctypes.byref(R), s), "ts_shard_rank_fwd_a")
        _frame._mark("rank fwd: fwd_a issued")
        mat_dev = exchange.gather(st.counts)
        st.count_host.copy_(mat_dev.view(-1), non_blocking=True)
        st.count_event.record(cur)
        exchange.rows(st.send, st.send_caps, st.recv_caps, out=st.recv)
        _frame._mark("rank fwd: counts gathered, records exchanged")
        _lib.check(lib.ts_shard_rank_fwd_b(ctypes.byref(fs), ctypes.byref(R), s), "ts_shard_rank_fwd_b")
        _frame._mark("rank fwd: fwd_b issued")
        # the frame's one look at its counts, with the whole forward pass enqueued
        total = 0
        if st.m > 0:
            event.record(cur)
            k = 0
            while word.value == -(1 << 31):
                k += 1
                if k > _frame._SPIN_LIMIT:
                    event.synchronize()
                    break
            total = int(word.value)
        st.count_event.synchronize()
    _frame._mark("rank fwd: waited for the counts (GPU time)")
    if total < 0:
        raise OverflowError("more than 2^31-1 tile intersections in one stripe")
    mat = st.count_np.astype(np.int64)
    st.total = total
    if int((st.caps - mat).min()) < 0:
        return None, mat, total                       # a group outgrew its capacity: EVERY rank sees that and repeats
    if total > st.list_cap:
        # this rank's lists outgrew theirs (the device left them empty): its own affair - larger lists, the stripe
        # stage again on the records it already holds, no collective
        st.list_cap = (int(total * _frame.CAPACITY_GROWTH) + 4096 + 63) & ~63
        st.bucket = torch.empty((2 * st.list_cap,), dtype=torch.int32, device=dev)
        fs.bucket_ids, fs.gaussian_ids_sorted = st.bucket.data_ptr(), st.bucket.data_ptr() + 4 * st.list_cap
        fs.capacity, fs.num_intersects = st.list_cap, st.list_cap
        st.partials = None
        _lib.check(lib.ts_shard_rank_fwd_b(ctypes.byref(fs), ctypes.byref(R), s), "ts_shard_rank_fwd_b")
    _frame._pairs_per_tile[dev.index] = total / max(1, st.s_cam.tile_rows * st.s_cam.tile_bounds_x)
    b = _LazyBinning()
    b.cam, b.n, b.num_tiles, b.num_intersects = st.s_cam, st.m, st.num_tiles, total
    b._ws, b._offs, b._bucket, b._cap = st.stripe_ws, st.s_offs, st.bucket, st.list_cap
    _frame.last_binning[dev.index] = b
    _frame._mark("rank fwd: exit")
    return out_img, mat, total


def _fast_backward(lib, s, dev, st: "_RankState", exchange: Exchange, v_img: Tensor, sh_degree: int, opacity_shape,
                   rest_shape):
    _frame._mark("rank bwd: enter")
    f32 = dict(dtype=torch.float32, device=dev)
    fo, fs, R, n, ch = st.fo, st.fs, st.R, st.n, st.ch
    segs = _frame.backward_segments(fs.cam, st.segs, st.total, dev.index)
    bwd_split = st.split and segs <= 1
    rows_n = st.list_cap * (4 if bwd_split else 1)
    if st.partials is None or st.partials.shape[0] < rows_n:
        st.partials = torch.empty((rows_n, _lib.PARTIAL_ROW_FLOATS), **f32)
    row_flags, fs.flag_gen = _frame.row_flags_for(dev, rows_n)
    fs.v_out_img, fs.partials, fs.row_flags = v_img.data_ptr(), st.partials.data_ptr(), row_flags.data_ptr()
    fs.num_intersects = st.total
    _frame._mark("rank bwd: rows / flags ready")
    _lib.check(lib.ts_shard_rank_bwd_a(ctypes.byref(fs), ctypes.byref(R), s), "ts_shard_rank_bwd_a")
    fs.num_intersects = st.list_cap
    _frame._mark("rank bwd: bwd_a issued")
    exchange.rows(st.grad_rows, st.recv_caps, st.send_caps, backward=True, out=st.back)
    _frame._mark("rank bwd: rows exchanged")
    nn = max(n, 1)
    k_op = 1
    for d in opacity_shape[1:]:
        k_op *= int(d)
    k_rest = 1
    for d in rest_shape[1:]:
        k_rest *= int(d)
    # ONE allocation for the seven gradient tensors handed out and the three intermediates
    widths = (2, k_op, 3, 3, 4, 3, k_rest, 7)
    flat = torch.empty((nn * sum(widths),), **f32)
    parts = flat.split([nn * k for k in widths])
    v_xy, v_opac = parts[0][:2 * n].view(n, 2), parts[1][:k_op * n].view((n,) + tuple(opacity_shape[1:]))
    v_means, v_scales, v_quats, v_dc = (parts[2][:3 * n].view(n, 3), parts[3][:3 * n].view(n, 3), parts[4][:4 * n].view(n, 4),
                                        parts[5][:3 * n].view(n, 3))
    v_rest = parts[6][:k_rest * n].view((n,) + tuple(rest_shape[1:]))
    p_tmp = parts[7].data_ptr()
    fo.v_xy, fo.v_opacity = v_xy.data_ptr(), v_opac.data_ptr()
    fo.v_conic, fo.v_colors, fo.v_depth = p_tmp, p_tmp + 12 * nn, p_tmp + 24 * nn
    fo.v_means, fo.v_scales, fo.v_quats = v_means.data_ptr(), v_scales.data_ptr(), v_quats.data_ptr()
    fo.v_colors_dc, fo.v_colors_rest = v_dc.data_ptr(), v_rest.data_ptr()
    fo.sh_degree = int(sh_degree)
    _frame._mark("rank bwd: gradients allocated")
    _lib.check(lib.ts_shard_rank_bwd_b(ctypes.byref(fo), ctypes.byref(R), s), "ts_shard_rank_bwd_b")
    _frame._mark("rank bwd: bwd_b issued")
    return (v_means, v_scales, v_quats, v_opac, v_dc, v_rest), v_xy


class _ShardedFrame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin,
                background, fx, fy, sh_degree, with_depth, layout, exchange, expect_backward=True):
        dev = _need_hip(means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin, background)
        inputs = tuple(_f32c(t) for t in (means, scales, quats, opacities, colors_dc, colors_rest, view34,
                                          projview, origin))
        if means.shape[0] != layout.owned[1] - layout.owned[0]:
            raise ValueError("the model shard does not hold the rows this rank owns")
        ch = 4 if with_depth else 3
        lib = _lib.load()
        # (which path a frame takes must be the same on every rank - the two paths make different collectives -, so it
        # depends only on what all ranks share: switches, the call pattern, the gathered count matrix)
        st = None
        if RANK_EXECUTOR and exchange.can_gather and _comm.gather_usable and not kernel_timer.enabled \
                and not _frame.CAPACITY_ALLOC and not PADDED_EXCHANGE:
            cur = torch.cuda.current_stream(dev)
            st = _rank_state(lib, dev, layout, means.shape[0], colors_rest.shape[1] + 1, ch, fx, fy, cur.cuda_stream)
            if _state_in_use(st):
                st = None                                    # (a frame still waits for its backward pass: stage functions)
        if st is not None and st.caps is not None:
            with torch.cuda.device(dev):
                if st.mode != _frame._list_mode(dev.index, st.tiles16):
                    st.size(lib, st.caps, st.list_cap)       # the lists changed shape: this rank's own affair
                out, mat, total = _fast_forward(lib, cur, dev, st, exchange, inputs, background, int(sh_degree))
                rank_executor_frames[0] += 1
                if out is None:
                    st.renew(lib, mat, total)
            if out is not None:
                if not expect_backward:                      # no backward pass will come for this frame (no_grad)
                    st.renew(lib, mat, total)
                    ctx.state = None
                    ctx.mark_non_differentiable(st.xys, st.radii)
                    return out, st.xys, st.radii
                st.busy = True
                st.frame_id += 1
                ctx.token = _FrameToken()                    # the state is this frame's for as long as its graph lives
                st.token = weakref.ref(ctx.token)
                ctx.frame_id = st.frame_id
                ctx.state, ctx.layout, ctx.exchange = st, layout, exchange
                ctx.inputs = inputs
                ctx.caps_next = (mat, total)
                ctx.opacity_shape, ctx.rest_shape, ctx.sh_degree = opacities.shape, colors_rest.shape, int(sh_degree)
                xys, radii = st.xys, st.radii
                ctx.xys_out = xys
                ctx.mark_non_differentiable(xys, radii)
                ctx.set_materialize_grads(False)
                ctx.out_shape = tuple(out.shape)
                return out, xys, radii
            rank_executor_frames[1] += 1
        ctx.state = None
        rank_executor_frames[2] += 1
        with torch.cuda.device(dev):
            s = _stream(dev)
            O, send = _owner_stage(lib, s, dev, layout, exchange, *inputs, fx, fy, int(sh_degree), ch, keep=True)
            records = exchange.rows(send, O.send_counts, O.recv_counts)
            S, out = _stripe_stage(lib, s, dev, layout, records, background, fx, fy, ch, keep=True)
            if O.caps is not None:
                # the padded exchange's deferred look at the counts: every rank holds the same matrix
                O.count_event.synchronize()
                world = layout.world
                mat = O.count_host.numpy().reshape(world, world)
                fits = bool((mat <= O.caps).all())
                _route_caps[O.key] = _capacity(mat)
                padded_frames[0] += 1
                if not fits:                     # a group outgrew its capacity: again, with exact sizes (all ranks)
                    padded_frames[1] += 1
                    del S, out, records, send
                    O, send = _owner_stage(lib, s, dev, layout, exchange, *inputs, fx, fy, int(sh_degree), ch,
                                           keep=True, padded=False)
                    records = exchange.rows(send, O.send_counts, O.recv_counts)
                    S, out = _stripe_stage(lib, s, dev, layout, records, background, fx, fy, ch, keep=True)
        if st is not None and st.caps is None:
            # the first frame of a layout ran with exact sizes: its counts give the executor's first capacities.  The
            # count matrix has to be the same on every rank: one all_gather of the counts, outside the frame's path
            with torch.cuda.device(dev):
                mat = exchange.gather(O.counts).cpu().numpy().astype(np.int64)
                st.size(lib, _capacity(mat), int(S.total * _frame.CAPACITY_GROWTH) + 4096)
        O.inputs = inputs
        ctx.owner, ctx.stripe, ctx.layout, ctx.exchange = O, S, layout, exchange
        ctx.opacity_shape, ctx.rest_shape, ctx.sh_degree = opacities.shape, colors_rest.shape, int(sh_degree)
        xys, radii = O.xys, O.radii
        ctx.xys_out = xys
        ctx.mark_non_differentiable(xys, radii)
        ctx.set_materialize_grads(False)
        ctx.out_shape = tuple(out.shape)
        return out, xys, radii

    @staticmethod
    def backward(ctx, v_img, _v_xys, _v_radii):
        lib = _lib.load()
        st = ctx.state
        if st is not None:                                  # the rank executor's frame
            dev = st.dev
            if not st.busy or ctx.frame_id != st.frame_id:
                # a second backward pass over a retained graph: the state's buffers have been handed to later frames (or
                # resized by this frame's own first pass) - silently wrong gradients otherwise
                raise RuntimeError("this sharded frame's buffers were released by its first backward pass (the rank "
                                   "executor keeps one frame per layout); for retain_graph=True set TS_RANK_EXECUTOR=0")
            try:
                with torch.cuda.device(dev):
                    if v_img is None:
                        v_img = torch.zeros(ctx.out_shape, dtype=torch.float32, device=dev)
                    grads, v_xy = _fast_backward(lib, _stream(dev), dev, st, ctx.exchange, _f32c(v_img), ctx.sh_degree,
                                                 ctx.opacity_shape, ctx.rest_shape)
                    st.renew(lib, *ctx.caps_next)
                    _frame._mark("rank bwd: capacities renewed")
            finally:
                st.busy, st.token = False, None
            xo = ctx.xys_out
            v_xy = v_xy.clone()                             # (the gradients share one allocation; xys.grad outlives them)
            xo.grad = v_xy if xo.grad is None else xo.grad + v_xy
            return grads + (None,) * 11
        O, S, layout, exchange = ctx.owner, ctx.stripe, ctx.layout, ctx.exchange
        dev = O.xys.device
        with torch.cuda.device(dev):
            s = _stream(dev)
            if v_img is None:
                v_img = torch.zeros(ctx.out_shape, dtype=torch.float32, device=dev)
            grad_rows = _stripe_backward(lib, s, dev, S, O.ch, _f32c(v_img))
            back = exchange.rows(grad_rows, O.recv_counts, O.send_counts, backward=True)
            grads, v_xy = _owner_backward(lib, s, dev, layout, O, back, ctx.sh_degree, ctx.opacity_shape,
                                          ctx.rest_shape)
        xo = ctx.xys_out                       # what extras['xys'].grad holds in the reference, for the OWNED rows
        xo.grad = v_xy if xo.grad is None else xo.grad + v_xy
        return grads + (None,) * 11


def render_sharded(model_shard, camera, device, layout: ShardLayout, exchange: Exchange, with_depth: bool = False):
    """This rank's stripe of the reference frame (rasterize.py:26-62) from the Gaussians ALL ranks own.
    ``model_shard`` holds the rows ``layout.owned`` of the model.  -> (out[rows, W, 3 or 4] with RGB clamped to
    <= 1 and, with ``with_depth``, the depth map as channel 3; (row_begin_px, row_end_px); xys[n_owned, 2] whose
    ``.grad`` receives the owned Gaussians' 2-D position gradients in backward)."""
    view, projview, origin = camera_on_device(camera, device)
    out, xys, _ = _ShardedFrame.apply(model_shard.means, model_shard.scales, model_shard.quats,
                                      model_shard.opacities, model_shard.colors_dc, model_shard.colors_rest,
                                      view[:3, :], projview, origin, model_shard.background, camera.f_x, camera.f_y,
                                      model_shard.active_sh_degree, with_depth, layout, exchange,
                                      torch.is_grad_enabled() and any(p.requires_grad for p in model_shard.parameters()))
    y0 = 16 * layout.tile_rows[0]
    return out, (y0, y0 + out.shape[0]), xys


class _LocalCounts(Exchange):
    """counts() only: the owner stage of one rank run on its own (export_records, simulate_frame)."""

    def counts(self, counts_dev):
        c = counts_dev.cpu().tolist()
        return c, c


@torch.no_grad()
def export_records(model_shard, camera, device, layout: ShardLayout, with_depth: bool = False):
    """Owner stage only: the records this rank would send, grouped by destination -> (records, send_counts).
    (bench.py's --emulate-ranks records the other ranks' traffic with it.)"""
    dev = torch.device(device)
    view, projview, origin = camera_on_device(camera, dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        O, send = _owner_stage(lib, _stream(dev), dev, layout, _LocalCounts(), *_inputs(model_shard, view, projview, origin),
                               camera.f_x, camera.f_y, int(model_shard.active_sh_degree), 4 if with_depth else 3,
                               keep=False)
    return send, O.send_counts


@torch.no_grad()
def simulate_frame(model, camera, dims, device, world: int, v_img_of, with_depth: bool = False,
                   stripes: Optional[Sequence[int]] = None):
    """The sharded frame of ALL ``world`` ranks executed one rank after the other in THIS process on one GPU,
    with the two exchanges done by slicing each other's buffers: the same stage functions, kernels and buffer
    layouts as ``render_sharded`` under torch.distributed, no process group.  For full-size functional checks
    on 1-GPU boxes.  ``model``: the WHOLE model; ``v_img_of(rank, image, (y0, y1))`` -> upstream gradient of that
    rank's stripe image.  -> (list of stripe images, list of (y0, y1), gradients of the six whole parameter
    tensors in ``model.parameters()`` order, v_xy[N, 2], per-rank record counts)."""
    dev = torch.device(device)
    n_total = model.means.shape[0]
    view, projview, origin = camera_on_device(camera, dev)
    lib = _lib.load()
    ch = 4 if with_depth else 3
    layouts = [ShardLayout(n_total, world, k, dims, stripes) for k in range(world)]
    owners, sends, stripes_, images, rows_px = [], [], [], [], []
    with torch.cuda.device(dev):
        s = _stream(dev)
        for k, lay in enumerate(layouts):
            shard = shard_model(model, world, k)
            inp = _inputs(shard, view, projview, origin)
            O, send = _owner_stage(lib, s, dev, lay, _LocalCounts(), *inp, camera.f_x, camera.f_y,
                                   int(model.active_sh_degree), ch, keep=True)
            O.inputs = inp
            owners.append(O)
            sends.append(send)
        offs = [[0] for _ in range(world)]
        for k in range(world):
            for c in owners[k].send_counts:
                offs[k].append(offs[k][-1] + c)
        recv_counts = [[owners[src].send_counts[k] for src in range(world)] for k in range(world)]
        for k, lay in enumerate(layouts):
            records = torch.cat([sends[src][offs[src][k]:offs[src][k + 1]] for src in range(world)], dim=0)
            S, out = _stripe_stage(lib, s, dev, lay, records, model.background, camera.f_x, camera.f_y, ch, keep=True)
            stripes_.append(S)
            images.append(out)
            y0 = 16 * lay.tile_rows[0]
            rows_px.append((y0, y0 + out.shape[0]))
        grad_rows = []
        for k in range(world):
            v_img = _f32c(v_img_of(k, images[k], rows_px[k]))
            grad_rows.append(_stripe_backward(lib, s, dev, stripes_[k], ch, v_img))
        roff = [[0] for _ in range(world)]
        for k in range(world):
            for c in recv_counts[k]:
                roff[k].append(roff[k][-1] + c)
        grads, v_xys = [], []
        for k, lay in enumerate(layouts):
            back = torch.cat([grad_rows[d][roff[d][k]:roff[d][k + 1]] for d in range(world)], dim=0)
            g, v_xy = _owner_backward(lib, s, dev, lay, owners[k], back, int(model.active_sh_degree),
                                      model.opacities.shape, model.colors_rest.shape)
            grads.append(g)
            v_xys.append(v_xy)
    # (v_means, v_scales, v_quats, v_opac, v_dc, v_rest) per rank -> model.parameters() order
    cat = lambda j: torch.cat([g[j] for g in grads], dim=0)
    whole = [cat(0), cat(4), cat(5), cat(1), cat(2), cat(3)]
    return images, rows_px, whole, torch.cat(v_xys, dim=0), recv_counts
