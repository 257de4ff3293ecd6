"""Event timing of the collectives of a sharded frame (bench.py at N > 1: "how long did the ranks spend in
RCCL"), and the preflight that decides whether the Gaussian-sharded exchange can run on this node.

The timer is off unless a caller switches it on; switched on it brackets every collective of
sharded.DistExchange, frame.py and sharding.py with a pair of events on the current stream."""
from __future__ import annotations

import contextlib
from typing import List, Tuple

import torch
import torch.distributed as dist


class CollectiveTimer:
    def __init__(self):
        self.enabled = False
        self._pairs: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []
        self.calls = 0
        self.bytes = []            # (label, bytes this rank hands to the collective) per call since start()

    def start(self) -> None:
        self.enabled, self._pairs, self.calls, self.bytes = True, [], 0, []

    def stop(self) -> Tuple[float, int]:
        """-> (milliseconds inside collectives since start(), number of collective calls)."""
        self.enabled = False
        if self._pairs:
            self._pairs[-1][1].synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._pairs)
        calls, self._pairs, self.calls = self.calls, [], 0
        return ms, calls

    @contextlib.contextmanager
    def span(self, on_device: bool = True, label: str = "", nbytes: int = 0):
        """``label`` / ``nbytes``: what the call is and how many bytes THIS rank hands to it (its send buffer; for an
        all-reduce the buffer itself) - bench.py's N > 1 line reports them per step beside the time."""
        if self.enabled and label:
            self.bytes.append((label, int(nbytes)))
        if not self.enabled or not on_device or not torch.cuda.is_available():
            if self.enabled:
                self.calls += 1
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        try:
            yield
        finally:
            b.record()
            self._pairs.append((a, b))
            self.calls += 1


collective_timer = CollectiveTimer()


gather_usable = True     # cleared by the preflight when the backend cannot all_gather the record counts


def preflight_sharded_exchange(device, group=None) -> Tuple[bool, str]:
    """Every rank runs the SAME tiny ``all_to_all_single`` with uneven split sizes (the call shape of
    sharded.DistExchange.rows) and the ``all_gather`` of the record counts (DistExchange.gather) before any frame,
    then the ranks agree with an all-reduce: a backend that rejects the
    call does so on all ranks in this first collective, not on one rank in the middle of a frame with its peers
    already blocked in the next exchange.  -> (usable on ALL ranks, this rank's error text or "")."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ok, err = 1, ""
    try:
        via_host = dist.get_backend(group) == "gloo"
        dev = torch.device("cpu") if via_host else device
        send_counts = [1 + ((rank + d) % 3) for d in range(world)]               # rows to each destination
        recv_counts = [1 + ((s + rank) % 3) for s in range(world)]               # = what source s sends to this rank
        send = torch.full((sum(send_counts), 4), float(rank), device=dev)
        got = torch.empty((sum(recv_counts), 4), device=dev)
        dist.all_to_all_single(got, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
        want = torch.cat([torch.full((c, 4), float(s)) for s, c in enumerate(recv_counts)])
        if not torch.equal(got.cpu(), want):
            ok, err = 0, "all_to_all_single with split sizes returned wrong rows"
    except Exception as e:                                      # noqa: BLE001 - whatever the backend raises
        ok, err = 0, f"{type(e).__name__}: {e}"
    # the count matrix of the OPTIONAL padded exchange (sharded.DistExchange.gather): one all_gather of `world` ints.
    # A backend that cannot do it only loses that option (`gather_usable`), not the sharded design.
    g_ok = 1
    try:
        via_host = dist.get_backend(group) == "gloo"
        dev = torch.device("cpu") if via_host else device
        mine = torch.arange(world, dtype=torch.int32, device=dev) + 100 * rank
        if via_host:
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            mat = torch.stack(parts)
        else:
            mat = torch.empty((world, world), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(mat.view(-1), mine, group=group)
        want_m = torch.arange(world, dtype=torch.int32)[None, :] + 100 * torch.arange(world, dtype=torch.int32)[:, None]
        if not torch.equal(mat.cpu(), want_m):
            g_ok = 0
    except Exception:                                           # noqa: BLE001
        g_ok = 0
    flag = torch.tensor([ok, g_ok], dtype=torch.int32, device=device if dist.get_backend(group) != "gloo" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    global gather_usable
    gather_usable = bool(int(flag[1].item()))
    return bool(int(flag[0].item())), err
