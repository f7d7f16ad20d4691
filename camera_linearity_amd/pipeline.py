"""Streaming many host-resident exposure stacks through one GPU (BASELINE config 5: a batch of independent stacks).

A single merge is 0.14 ms of kernel time between 6 ms of host-to-device and 7 ms of device-to-host copy (config 2,
DESIGN.md section 6), so for a stream of stacks the copies are the pipeline. `MergePipeline` overlaps them: the H2D copy of
stack k+1, the fused merge of stack k and the D2H copy of stack k-1 run on three HIP streams, ordered by events,
over `depth` slots of pinned host staging and device buffers. PCIe is full duplex, so the steady state costs
max(H2D, D2H) per stack instead of their sum.

The producer fills the pinned input views of a slot in place (e.g. a TIFF decoder writing straight into them), so no
pageable-to-pinned copy sits in front of the DMA. Results are handed out as NumPy views of the slot's pinned output
buffer, valid until the next result is requested.

The reference has no counterpart: its loop loads, merges and saves one series at a time
(modules/exposure_series.py:399-419 called per series by its scripts).
"""
from __future__ import annotations

from typing import Callable, Iterator, Optional, Sequence

import numpy as np
import torch

from . import engine


class _Slot:
    pass


class MergePipeline:
    def __init__(self, n_frames: int, height: int, width: int, exposures: Sequence[float], icrf, icrf_diff=None,
                 channels: int = 3, with_std: bool = False, device=None, depth: int = 2):
        if depth < 2:
            raise ValueError("depth must be at least 2 (one slot in flight while the next is filled)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n, self.shape, self.with_std, self.depth = n_frames, (height, width, channels), with_std, depth
        if with_std and icrf_diff is None:
            raise ValueError("uncertainty propagation needs ICRF_diff")
        self.s_h2d = torch.cuda.Stream(self.device)
        self.s_run = torch.cuda.Stream(self.device)
        self.s_d2h = torch.cuda.Stream(self.device)
        self.slots = []
        for _ in range(depth):
            s = _Slot()
            s.h_frames = [torch.empty(self.shape, dtype=torch.uint8, pin_memory=True) for _ in range(n_frames)]
            s.h_stds = [torch.empty(self.shape, dtype=torch.float64, pin_memory=True) for _ in range(n_frames)] if with_std else None
            s.d_frames = [torch.empty(self.shape, dtype=torch.uint8, device=self.device) for _ in range(n_frames)]
            s.d_stds = [torch.empty(self.shape, dtype=torch.float64, device=self.device) for _ in range(n_frames)] if with_std else None
            s.plan = engine.plan_merge(s.d_frames, exposures, icrf, icrf_diff if with_std else None, s.d_stds)
            s.h_val = torch.empty(self.shape, dtype=torch.float64, pin_memory=True)
            s.h_std = torch.empty(self.shape, dtype=torch.float64, pin_memory=True) if with_std else None
            s.ev_h2d = torch.cuda.Event()
            s.ev_run = torch.cuda.Event()
            s.ev_d2h = torch.cuda.Event()
            s.busy = False
            self.slots.append(s)

    # ---- producer side
    def input_views(self, slot: int):
        """NumPy views of slot `slot`'s pinned staging: (frames [N x (H, W, C) uint8], stds [N x float64] or None)."""
        s = self.slots[slot]
        return [t.numpy() for t in s.h_frames], (None if s.h_stds is None else [t.numpy() for t in s.h_stds])

    # ---- one stack
    def _submit(self, s: _Slot) -> None:
        with torch.cuda.stream(self.s_h2d):
            for d, h in zip(s.d_frames, s.h_frames):
                d.copy_(h, non_blocking=True)
            if self.with_std:
                for d, h in zip(s.d_stds, s.h_stds):
                    d.copy_(h, non_blocking=True)
            s.ev_h2d.record(self.s_h2d)
        self.s_run.wait_event(s.ev_h2d)
        s.plan.launch(stream=self.s_run.cuda_stream)
        s.ev_run.record(self.s_run)
        self.s_d2h.wait_event(s.ev_run)
        with torch.cuda.stream(self.s_d2h):
            s.h_val.copy_(s.plan.outputs["val"], non_blocking=True)
            if self.with_std:
                s.h_std.copy_(s.plan.outputs["std"], non_blocking=True)
            s.ev_d2h.record(self.s_d2h)
        s.busy = True

    def _collect(self, s: _Slot):
        s.ev_d2h.synchronize()
        s.busy = False
        return s.h_val.numpy(), (None if s.h_std is None else s.h_std.numpy())

    def run(self, fill: Callable[[int, list, Optional[list]], bool]) -> Iterator[tuple]:
        """Drive the pipeline. `fill(k, frame_views, std_views)` writes stack k into the given pinned views and returns
        True, or returns False when there are no more stacks. Yields (k, val, std) in order; the arrays are views of the
        slot's pinned output staging and are valid until the next result is requested (the slot is refilled then)."""
        k = 0
        pending = []            # (k, slot) in submission order
        done = False
        while not done or pending:
            slot = self.slots[k % self.depth]
            if not done:
                if slot.busy:                        # the oldest stack still owns this slot: hand it out first
                    kk, ss = pending.pop(0)
                    val, std = self._collect(ss)
                    yield kk, val, std
                fv, sv = self.input_views(k % self.depth)
                if fill(k, fv, sv):
                    self._submit(slot)
                    pending.append((k, slot))
                    k += 1
                else:
                    done = True
            else:
                kk, ss = pending.pop(0)
                val, std = self._collect(ss)
                yield kk, val, std

    def merge_many(self, stacks: Sequence[Sequence[np.ndarray]], stds: Optional[Sequence[Sequence[np.ndarray]]] = None):
        """Convenience: merge a list of in-memory stacks (each N uint8 (H, W, C) arrays); returns copies of the results."""
        def fill(k, fv, sv):
            if k >= len(stacks):
                return False
            for dst, src in zip(fv, stacks[k]):
                np.copyto(dst, src)
            if sv is not None:
                for dst, src in zip(sv, stds[k]):
                    np.copyto(dst, src)
            return True
        out = []
        for _, val, std in self.run(fill):
            out.append((val.copy(), None if std is None else std.copy()))
        return out
