"""CPU oracle for the MobileFaceNet forward.  TEST INFRASTRUCTURE ONLY.

This file restates, op for op and un-fused, what the reference's
``Mobilefacenet/face_model.py`` computes in eval mode, as plain functional
PyTorch fp32 on the CPU.  It is the checker for the HIP path; nothing under
``face_recognition_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do.

Pinning: ``tests/golden/make_golden.py`` ran the reference module itself
(imported from /root/reference) and this restatement on the same seeded
weights and inputs; they agree to fp32 round-off, and the reference's outputs
are committed as ``tests/golden/embeddings_b4.npz``.  The reference has no
tests or golden vectors of its own for this path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _t(sd, key) -> torch.Tensor:
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def l2_norm(x: torch.Tensor, axis: int = 1) -> torch.Tensor:
    """face_model.py:19-22 -- x / ||x||_2 along ``axis``; no epsilon (zero rows give NaN)."""
    return torch.div(x, torch.norm(x, 2, axis, True))


def _conv_bn(sd, prefix: str, x, stride: int, pad: int, groups: int) -> torch.Tensor:
    """Conv2d(bias=False) then BatchNorm2d with running statistics (face_model.py:33-34, 44-45)."""
    y = F.conv2d(x, _t(sd, prefix + ".conv.weight"), None, stride, pad, 1, groups)
    return F.batch_norm(y, _t(sd, prefix + ".bn.running_mean"), _t(sd, prefix + ".bn.running_var"),
                        _t(sd, prefix + ".bn.weight"), _t(sd, prefix + ".bn.bias"), False, 0.0, BN_EPS)


def conv_block(sd, prefix, x, stride=1, pad=0, groups=1) -> torch.Tensor:
    """face_model.py:26-36 -- conv, bn, per-channel PReLU."""
    return F.prelu(_conv_bn(sd, prefix, x, stride, pad, groups), _t(sd, prefix + ".prelu.weight"))


def linear_block(sd, prefix, x, stride=1, pad=0, groups=1) -> torch.Tensor:
    """face_model.py:38-46 -- conv, bn, no activation."""
    return _conv_bn(sd, prefix, x, stride, pad, groups)


def depth_wise(sd, prefix, x, cexp: int, stride: int, residual: bool) -> torch.Tensor:
    """face_model.py:48-65 -- 1x1 expand, 3x3 depthwise, 1x1 project, optional shortcut."""
    y = conv_block(sd, prefix + ".conv", x)
    y = conv_block(sd, prefix + ".conv_dw", y, stride, 1, cexp)
    y = linear_block(sd, prefix + ".project", y)
    return x + y if residual else y


# (prefix, expanded channels, stride, residual) in forward order: face_model.py:82-87, 108-118
_BLOCKS: List[Tuple[str, int, int, bool]] = (
    [("conv_23", 128, 2, False)]
    + [(f"conv_3.model.{i}", 128, 1, True) for i in range(4)]
    + [("conv_34", 256, 2, False)]
    + [(f"conv_4.model.{i}", 256, 1, True) for i in range(6)]
    + [("conv_45", 512, 2, False)]
    + [(f"conv_5.model.{i}", 256, 1, True) for i in range(2)]
)


def forward(sd: Dict[str, object], x: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """face_model.py:103-129.  x: f32 [B,3,112,112] -> f32 [B,E], rows unit-norm.

    ``taps`` (optional dict) receives every intermediate activation by name, NCHW f32.
    """
    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    with torch.no_grad():
        out = tap("conv1", conv_block(sd, "conv1", x, 2, 1, 1))
        out = tap("conv2_dw", conv_block(sd, "conv2_dw", out, 1, 1, 64))
        for prefix, cexp, stride, res in _BLOCKS:
            out = tap(prefix, depth_wise(sd, prefix, out, cexp, stride, res))
        out = tap("conv_6_sep", conv_block(sd, "conv_6_sep", out))
        out = tap("conv_6_dw", linear_block(sd, "conv_6_dw", out, 1, 0, 512))
        out = out.view(out.size(0), -1)                       # Flatten, face_model.py:15-17
        out = F.linear(out, _t(sd, "linear.weight"))          # face_model.py:91,126
        out = F.batch_norm(out, _t(sd, "bn.running_mean"), _t(sd, "bn.running_var"),
                           _t(sd, "bn.weight"), _t(sd, "bn.bias"), False, 0.0, BN_EPS)
        tap("pre_norm", out)
        return l2_norm(out)


def forward_numpy(sd, x: np.ndarray, threads: Optional[int] = None) -> np.ndarray:
    if threads is not None:
        torch.set_num_threads(threads)
    return forward(sd, torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).numpy()


def tta_combine(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """facebank.py:62, cam_demo.py:92, Evaluation.py:51-52 -- l2_norm(emb + emb_mirror)."""
    return l2_norm(a + b)


def embed_tta(sd, x: torch.Tensor) -> torch.Tensor:
    """Embedding with flip test-time augmentation; cv2.flip(img, 1) == flip of the W axis."""
    return tta_combine(forward(sd, x), forward(sd, torch.flip(x, dims=[3])))
