"""Host -> device image input of the hot path (second half; ref:icv_src/icv_datamodule.py:63-124: DataLoader
workers hand `processor.prepare_input` PIL images, which come back as normalised float tensors on the HOST and are moved with
`.to(device)` at the top of the step, ref:inference.py:277-278).

Here the bytes go over PCIe as they are — uint8, a quarter of the float32 the reference ships, half of bf16 — and become the model's
normalised bf16 pixel_values on the device (licv.frontend.preprocess_images, one HIP kernel):

    feeder = ImageFeeder(device, max_images, H, W)             # two pinned staging buffers + two device byte buffers
    t = feeder.submit(images)                                  # pack into pinned memory, async H2D + kernel on a SIDE stream
    ...                                                        # the step before this one is still computing on the main stream
    pixel_values, mask = feeder.get(t, B, N)                   # the main stream waits for THAT batch's event only

`submit` returns at once (the copy and the kernel are queued on the side stream); a staging buffer is reused only after the event
of the batch that last used it has completed.  32-shot at ~43 questions/s is ~1.4 k images/s per GPU (~210 MB/s of bytes): the
link is nowhere near busy, what matters is that nothing of it sits on the step's critical path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import frontend


class ImageFeeder:
    def __init__(self, device, max_images: int, height: int, width: int, mean=frontend.IDEFICS_MEAN, std=frontend.IDEFICS_STD,
                 rescale: float = 1 / 255, with_mask: bool = False, depth: int = 2):
        self.device = torch.device(device)
        self.max_images, self.H, self.W = int(max_images), int(height), int(width)
        self.mean, self.std, self.rescale, self.with_mask = tuple(mean), tuple(std), float(rescale), bool(with_mask)
        self.depth = int(depth)
        shape = (self.max_images, self.H, self.W, 3)
        self.host = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(self.depth)]
        self.host_hw = [torch.empty((self.max_images, 2), dtype=torch.int32).pin_memory() for _ in range(self.depth)]
        self.dev_u8 = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(self.depth)]
        self.dev_hw = [torch.empty((self.max_images, 2), dtype=torch.int32, device=self.device) for _ in range(self.depth)]
        self.out = [torch.empty((self.max_images, 3, self.H, self.W), dtype=torch.bfloat16, device=self.device) for _ in range(self.depth)]
        self.mask = [torch.empty((self.max_images, self.H, self.W), dtype=torch.uint8, device=self.device) if with_mask else None
                     for _ in range(self.depth)]
        self.stream = torch.cuda.Stream(device=self.device)
        self.done = [None] * self.depth                     # event of the batch that last used a slot
        self.consumed = [None] * self.depth                 # event after which the main stream no longer reads the slot's output
        self.handed_out = [False] * self.depth              # get() returned views of the slot and release() has not been called since
        self.turn = 0

    # -------------------------------------------------------------------------------------------- host side
    def _pack(self, slot: int, images) -> Tuple[int, bool]:
        """images: a (n, H, W, 3) uint8 array / tensor (uniform size), or a sequence of (h, w, 3) uint8 arrays (h <= H, w <= W:
        ragged, padded here with zeros).  Returns (n, ragged)."""
        host = self.host[slot]
        if isinstance(images, torch.Tensor) or (isinstance(images, np.ndarray) and images.ndim == 4):
            t = images if isinstance(images, torch.Tensor) else torch.from_numpy(images)
            n = t.shape[0]
            assert n <= self.max_images and tuple(t.shape[1:]) == (self.H, self.W, 3) and t.dtype == torch.uint8
            host[:n].copy_(t)
            return n, False
        n = len(images)
        assert n <= self.max_images
        hw = self.host_hw[slot]
        view = host.numpy()
        ragged = False
        for i, im in enumerate(images):
            if im is None:                                   # a missing image: all padding (Idefics2 drops all-zero images)
                view[i] = 0
                hw[i, 0] = 0; hw[i, 1] = 0
                ragged = True
                continue
            h, w = im.shape[:2]
            assert im.dtype == np.uint8 and im.shape[2] == 3 and h <= self.H and w <= self.W
            if h != self.H or w != self.W:
                view[i] = 0
                ragged = True
            view[i, :h, :w] = im
            hw[i, 0] = h; hw[i, 1] = w
        return n, ragged

    def submit(self, images) -> Tuple[int, int]:
        """Queue one batch; returns a ticket for get()."""
        slot = self.turn
        self.turn = (self.turn + 1) % self.depth
        if self.done[slot] is not None:
            self.done[slot].synchronize()                   # the H2D copy out of this pinned buffer has finished
        n, ragged = self._pack(slot, images)
        if self.handed_out[slot]:
            # The slot's output was handed out by get() and never release()d: the forward that reads it was issued on the caller's
            # stream some time between that get() and now, so an event recorded on that stream NOW is behind it.  (release() marks the
            # exact point and keeps more overlap; this is the safe default for callers that never call it.)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.consumed[slot] = ev
            self.handed_out[slot] = False
        with torch.cuda.stream(self.stream):
            if self.consumed[slot] is not None:
                self.stream.wait_event(self.consumed[slot])  # the step that read this slot's output has finished with it
            self.dev_u8[slot][:n].copy_(self.host[slot][:n], non_blocking=True)
            hw = None
            if ragged or self.with_mask:
                if not ragged:
                    self.host_hw[slot][:n, 0] = self.H; self.host_hw[slot][:n, 1] = self.W
                self.dev_hw[slot][:n].copy_(self.host_hw[slot][:n], non_blocking=True)
                hw = self.dev_hw[slot][:n]
            frontend.preprocess_images(self.dev_u8[slot][:n], self.mean, self.std, self.rescale, valid_hw=hw,
                                       out=self.out[slot][:n], mask_out=self.mask[slot][:n] if self.with_mask else None)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.done[slot] = ev
        return slot, n

    # -------------------------------------------------------------------------------------------- device side
    def get(self, ticket: Tuple[int, int], batch: Optional[int] = None, per_row: Optional[int] = None):
        """The current stream waits for the ticket's batch; returns (pixel_values, pixel_attention_mask or None), shaped
        (batch, per_row, 3, H, W) / (batch, per_row, H, W) when both counts are given, else (n, 3, H, W) / (n, H, W).  The tensors
        are views of the feeder's buffers: valid for every kernel issued on the current stream until `depth` further batches have
        been submitted (the submit that reuses the slot orders its copy behind everything the current stream holds at that point;
        release() marks an earlier point and so keeps more of the copy / compute overlap)."""
        slot, n = ticket
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.done[slot])
        pv = self.out[slot][:n]
        m = self.mask[slot][:n].view(torch.bool) if self.with_mask else None
        if batch is not None and per_row is not None:
            assert batch * per_row == n
            pv = pv.view(batch, per_row, 3, self.H, self.W)
            m = m.view(batch, per_row, self.H, self.W) if m is not None else None
        self.handed_out[slot] = True                        # until release(): submit() then orders the slot's reuse behind the reader
        return pv, m

    def release(self, ticket: Tuple[int, int]):
        """Mark the point on the current stream after which the ticket's output is no longer read (call after the forward that
        consumed it was issued).  Optional: without it the submit that reuses the slot takes the state of the current stream at ITS
        call as that point (always safe, less overlap)."""
        slot, _ = ticket
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.consumed[slot] = ev
        self.handed_out[slot] = False
