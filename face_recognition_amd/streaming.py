"""Host crops -> embeddings with the copy of one batch overlapping the forward of the batch before it.

The reference embeds what ``cv2`` / ``Face_alignment`` hand it one image at a time: ``test_transform(img).to(device)``
and a forward per image (facebank.py:50-62, cam_demo.py:91-96).  ``MobileFaceNet.forward_u8`` already takes the raw
uint8 HWC crops on the device; what is left between a host that produces crops and the GPU is the PCIe copy: 38.5 MB
per 1024 crops, about as long as half their forward.  ``EmbeddingStream`` hides it: ``depth`` pinned host buffers that
the producer fills in place, a copy stream that moves buffer i while the forward of buffer i - 1 runs, events both
ways (a device buffer is overwritten only after the forward that read it, a forward starts only after its copy).
Embeddings are those of ``forward_u8`` on the same crops, bit for bit.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional

import torch

from . import layers as L


class EmbeddingStream:
    """``stream = EmbeddingStream(net, max_batch=1024)``; then either

    * ``for emb in stream.embed(batches)``: ``batches`` yields uint8 [b,112,112,3] host tensors (pinned ones are copied
      straight from where they are, others through the stream's own pinned buffers), ``emb`` is f32 [b,E] on the device;
    * or ``buf = stream.host_buffer(i)`` / fill ``buf[:b]`` in place / ``emb = stream.submit(i, b)`` for a producer that
      decodes into pinned memory itself (``i`` cycles through ``range(depth)``).

    ``tta=True`` gives ``l2_norm(net(x) + net(hflip(x)))`` per crop (facebank.py:59-62)."""

    def __init__(self, net, max_batch: int = 1024, depth: int = 2, tta: bool = False, device: Optional[torch.device] = None):
        if depth < 2:
            raise ValueError("EmbeddingStream needs at least two buffers to overlap anything")
        if max_batch < 1:
            raise ValueError("max_batch must be positive")
        dev = device if device is not None else next(net.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("EmbeddingStream needs the network on a HIP device")
        self.net, self.tta, self.device, self.depth, self.max_batch = net, bool(tta), dev, int(depth), int(max_batch)
        shape = (self.max_batch, L.INPUT_HW, L.INPUT_HW, 3)
        self._host: List[torch.Tensor] = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(self.depth)]
        self._dev: List[torch.Tensor] = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(self.depth)]
        with torch.cuda.device(dev):
            self._copy_stream = torch.cuda.Stream(dev)
            self._copied = [torch.cuda.Event() for _ in range(self.depth)]     # copy into _dev[i] finished
            self._consumed = [torch.cuda.Event() for _ in range(self.depth)]   # the forward that read _dev[i] finished
            cur = torch.cuda.current_stream(dev)
            for e in self._copied + self._consumed:
                e.record(cur)

    def host_buffer(self, slot: int) -> torch.Tensor:
        """Pinned uint8 [max_batch,112,112,3] buffer ``slot``; safe to fill once the copy of its last submit is done
        (this call waits for that)."""
        self._copied[slot].synchronize()
        return self._host[slot]

    def submit(self, slot: int, n: int, src: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Copy the first ``n`` crops of host buffer ``slot`` (or of ``src``, a pinned uint8 [>=n,112,112,3] host
        tensor) to the device on the copy stream and queue their forward on the current stream behind it."""
        if not 0 < n <= self.max_batch:
            raise ValueError(f"batch {n} not in [1,{self.max_batch}]")
        host = self._host[slot] if src is None else src
        if host.dtype != torch.uint8 or host.dim() != 4 or tuple(host.shape[1:]) != (L.INPUT_HW, L.INPUT_HW, 3) or host.shape[0] < n:
            raise RuntimeError(f"expected uint8 [>={n},112,112,3] on the host, got {host.dtype} {tuple(host.shape)}")
        if host.is_cuda or not host.is_pinned():
            raise RuntimeError("submit() copies from pinned host memory (embed() stages other tensors)")
        dev = self.device
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(self._consumed[slot])
                self._dev[slot][:n].copy_(host[:n], non_blocking=True)
                self._copied[slot].record(self._copy_stream)
            cur.wait_event(self._copied[slot])
            out = self.net.forward_u8(self._dev[slot][:n], tta=self.tta)
            self._consumed[slot].record(cur)
        return out

    def embed(self, batches: Iterable[torch.Tensor]) -> Iterator[torch.Tensor]:
        """Embeddings of every batch, in order.  The result of batch i is yielded after batch i + 1 has been queued, so
        a consumer that synchronises on it does not stall the pipeline."""
        pending: Optional[torch.Tensor] = None
        for i, b in enumerate(batches):
            slot = i % self.depth
            n = int(b.shape[0])
            if isinstance(b, torch.Tensor) and not b.is_cuda and b.is_pinned():
                out = self.submit(slot, n, src=b)
            else:
                self.host_buffer(slot)[:n].copy_(torch.as_tensor(b))
                out = self.submit(slot, n)
            if pending is not None:
                yield pending
            pending = out
        if pending is not None:
            yield pending
