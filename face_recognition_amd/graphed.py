"""hipGraph replay of the small-batch path (the per-frame loop of cam_demo.py / Video_demo.py:
embed the faces of one frame, optionally with flip-TTA, and match them against the facebank).

One frame's work is ~20-45 short kernels; launched one by one from Python they are bound by launch
overhead.  ``GraphedRecognizer`` captures the whole sequence -- forward (+ mirrored forward + TTA
combine) (+ match, threshold, score) -- into ONE hipGraph per batch size and replays it, so a frame
costs one graph launch.  Inputs are copied into a static buffer (or written there by the caller:
``input_buffer``); outputs are static tensors that stay valid until the next call with the same batch size.
A graph keeps the small-batch mode (``face_model.set_small_batch_mode``) it was captured in.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch

from .face_model import MobileFaceNet, tta_combine
from .facebank import Gallery, match_facebank


class _Captured:
    def __init__(self, graph, x, emb, match):
        self.graph, self.x, self.emb, self.match = graph, x, emb, match


class GraphedRecognizer:
    """``rec = GraphedRecognizer(model, targets, tta=False, threshold=60)``;
    ``emb, (minimum, min_idx, score_100) = rec(x)`` with x f32 [B,3,112,112] on the HIP device.
    ``targets`` may be None (embeddings only: the second result is None)."""

    def __init__(self, model: MobileFaceNet, targets: Union[torch.Tensor, Gallery, None] = None, tta: bool = False,
                 threshold: float = 60, max_batch: int = 64):
        if model.training:
            raise RuntimeError("call model.eval() first")
        self.model, self.tta, self.threshold, self.max_batch = model, tta, threshold, max_batch
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("GraphedRecognizer needs the model on the HIP device")
        self.gallery: Optional[Gallery] = None
        if targets is not None:
            self.gallery = targets if isinstance(targets, Gallery) else Gallery(targets, store="f32", device=self.device)
            self.gallery.reserve(max_batch)      # no scratch growth (= no stale pointers) once graphs exist
        # a workspace of our own: the module's grows (and moves) with eager calls of other sizes
        self._ws = torch.empty(model.workspace_bytes(max_batch, self.device, tta=tta), dtype=torch.uint8,
                               device=self.device)
        self._graphs: Dict[int, _Captured] = {}

    def _run(self, x):
        emb = self.model.embed_tta(x, workspace=self._ws) if self.tta else self.model(x, workspace=self._ws)
        match = match_facebank(emb, self.gallery, self.threshold) if self.gallery is not None else None
        return emb, match

    def _capture(self, batch: int) -> _Captured:
        x = torch.zeros((batch, 3, 112, 112), dtype=torch.float32, device=self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):          # warm-up off the capture: handle creation, scratch growth, attributes
            for _ in range(2):
                self._run(x)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            emb, match = self._run(x)
        return _Captured(g, x, emb, match)

    def input_buffer(self, batch: int) -> torch.Tensor:
        """The static f32 [batch,3,112,112] input of the graph for ``batch`` crops.  A per-frame loop writes its crops
        there (the host -> device copy of a frame's faces, or the alignment kernel's output) and calls ``rec(buf)``
        with that same tensor: the replay then starts without the staging copy a foreign tensor needs."""
        b = int(batch)
        if b < 1 or b > self.max_batch:
            raise RuntimeError(f"batch {b} outside [1,{self.max_batch}]")
        c = self._graphs.get(b)
        if c is None:
            c = self._graphs[b] = self._capture(b)
        return c.x

    def __call__(self, x: torch.Tensor) -> Tuple[torch.Tensor, Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]]:
        b = int(x.shape[0])
        if b < 1 or b > self.max_batch:
            raise RuntimeError(f"batch {b} outside [1,{self.max_batch}]")
        c = self._graphs.get(b)
        if c is None:
            c = self._graphs[b] = self._capture(b)
        if x.data_ptr() != c.x.data_ptr():                 # crops that are not in the graph's input already
            c.x.copy_(x, non_blocking=True)
        c.graph.replay()
        return c.emb, c.match
