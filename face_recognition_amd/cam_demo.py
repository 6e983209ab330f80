"""The numeric part of one frame of the reference's camera / video demos, in one call on the device
(reference: Mobilefacenet/cam_demo.py:80-105, video_demo.py's same loop).

Per frame the reference aligns every detected face on the host (``Face_alignment``), pushes the crops through the
network ONE AT A TIME (two forwards per face with ``--tta``), and matches with a broadcast difference.  Here the
frame goes to the device once; the aligned crops, the embeddings (all faces of the frame as one batch, the mirrored
copy of ``--tta`` read inside the first kernel) and the match never leave it.  Detection (MTCNN), drawing and the
camera stay where they are: ``bboxes, landmarks = create_mtcnn_net(...)`` feeds this, the labels and scores feed the
reference's drawing code.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

from .align_trans import align_faces
from .facebank import Gallery, match_facebank


def recognize_frame(model, frame, landmarks, targets: Union[torch.Tensor, Gallery], names: Sequence[str],
                    threshold: float = 60, tta: bool = False, device=None
                    ) -> Tuple[List[str], torch.Tensor, torch.Tensor]:
    """``frame``: the BGR frame, uint8 [H,W,3] (numpy array as ``cap.read()`` returns it, or a tensor already on the
    device); ``landmarks``: MTCNN's rows ``[x1..x5, y1..y5]``, one per face, in frame coordinates (cam_demo.py:75-77);
    ``targets`` / ``names``: what ``load_facebank`` / ``prepare_facebank`` return (``names[0]`` is the unknown slot).

    Returns ``(labels, score_100, results)``: ``names[results[i] + 1]`` per face (:120), the 0..100 score the demo
    prints (:105) and the facebank index or -1 (:101), the last two as tensors on the device.  No faces: three empties
    (the reference's loop fails at ``torch.cat([])`` and prints 'detect error')."""
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    n = 0 if landmarks is None else len(landmarks)
    if n == 0:
        return [], torch.empty(0, device=dev), torch.empty(0, dtype=torch.int64, device=dev)
    if not isinstance(frame, torch.Tensor):
        frame = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))
    crops = align_faces(frame.to(dev, non_blocking=True), landmarks)       # uint8 [F,112,112,3], :79
    embs = model.forward_u8(crops, tta=tta)                                # :87-94, all faces in one batch
    _, results, score_100 = match_facebank(embs, targets, threshold)       # :96-105
    idx = results.cpu().tolist()
    return [names[i + 1] for i in idx], score_100, results
