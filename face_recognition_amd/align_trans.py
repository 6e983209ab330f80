"""Face alignment on the HIP device: the reference's ``utils/align_trans.py``.

``transformation_from_points`` (align_trans.py:16-29) is five points of float64 arithmetic and stays in
numpy, written with the reference's expressions.  The image warp (``cv2.warpAffine`` + crop, :50-52)
runs on the device for every face of a frame in one launch (``fa_align_warp``) and produces the
uint8 HWC batch ``MobileFaceNet.forward_u8`` reads, so a frame goes detector -> crops -> embeddings
without leaving HBM.
"""
from __future__ import annotations

from typing import List, Sequence, Union

import numpy as np
import torch

from . import _lib

# align_trans.py:41-45 (square 112x112 template) and :56-60 (96-wide template)
COORD5POINT_SQUARE = [[38.29459953, 51.69630051], [73.53179932, 51.50139999], [56.02519989, 71.73660278],
                      [41.54930115, 92.3655014], [70.72990036, 92.20410156]]
COORD5POINT_96 = [[30.29459953, 51.69630051], [65.53179932, 51.50139999], [48.02519989, 71.73660278],
                  [33.54930115, 92.3655014], [62.72990036, 92.20410156]]


def transformation_from_points(points1, points2) -> np.ndarray:
    """Least-squares similarity transform taking points1 [n,2] onto points2 [n,2] (align_trans.py:16-29).

    Returns the 3x3 matrix [[sR, t], [0, 0, 1]] as a float64 ndarray (the reference builds the same values as
    an ``np.matrix``; callers slice ``M[:2]`` either way)."""
    p1 = np.array(points1, dtype=np.float64)
    p2 = np.array(points2, dtype=np.float64)
    c1, c2 = p1.mean(axis=0), p2.mean(axis=0)
    p1 -= c1
    p2 -= c2
    s1, s2 = np.std(p1), np.std(p2)           # one spread per point set, over both coordinates
    p1 /= s1
    p2 /= s2
    U, _, Vt = np.linalg.svd(p1.T @ p2)
    sR = (s2 / s1) * (U @ Vt).T
    M = np.eye(3)
    M[:2, :2] = sR
    M[:2, 2] = c2 - sR @ c1
    return M


def alignment_matrices(landmarks, template=COORD5POINT_SQUARE) -> np.ndarray:
    """MTCNN landmark rows [x1..x5, y1..y5] -> f64 [F,6], the M[:2] the reference hands to warpAffine (:36-50)."""
    landmarks = np.asarray(landmarks, dtype=np.float64).reshape(-1, 10)
    out = np.empty((landmarks.shape[0], 6), dtype=np.float64)
    pts2 = np.asarray(template, dtype=np.float64)
    for i in range(landmarks.shape[0]):
        out[i] = transformation_from_points(landmarks[i].reshape(2, 5).T, pts2)[:2].reshape(6)
    return out


def warp_faces(frame: torch.Tensor, matrices: Union[np.ndarray, torch.Tensor], out_h: int = 112, out_w: int = 112) -> torch.Tensor:
    """frame uint8 [H,W,3] on the device, matrices f64 [F,6] -> uint8 [F,out_h,out_w,3] on the device."""
    if not (isinstance(frame, torch.Tensor) and frame.is_cuda):
        raise RuntimeError("warp_faces: the frame must be a tensor on the HIP device")
    if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
        raise RuntimeError(f"warp_faces: expected uint8 [H,W,3], got {frame.dtype} {tuple(frame.shape)}")
    dev = frame.device
    frame = frame.contiguous()
    m = torch.as_tensor(matrices, dtype=torch.float64).reshape(-1, 6).contiguous().to(dev)
    n = m.shape[0]
    crops = torch.empty((n, out_h, out_w, 3), dtype=torch.uint8, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.fa_align_warp(frame.data_ptr(), frame.shape[0], frame.shape[1], m.data_ptr() if n else None, n,
                                     crops.data_ptr(), out_h, out_w, _lib.current_stream_ptr(dev)), "fa_align_warp")
    return crops


def align_faces(frame: torch.Tensor, landmarks) -> torch.Tensor:
    """Device pipeline form of ``Face_alignment(img, True, landmarks)``: uint8 [F,112,112,3] on the device."""
    return warp_faces(frame, alignment_matrices(landmarks))


def Face_alignment(img, default_square: bool = True, landmarks: Sequence = (), device=None) -> List[np.ndarray]:
    """Drop-in for align_trans.py:31-69: list of aligned crops (uint8 arrays, 112x112x3), one per landmark row.

    ``img`` is the BGR frame as a numpy array (as the reference passes it) or a uint8 tensor already on
    the device.  The reference's ``default_square=False`` branch reads an undefined name
    (``landmarks_one``, :62) and cannot run; it raises here as well."""
    if landmarks is None or len(landmarks) == 0:
        return []
    if not default_square:
        raise NameError("name 'landmarks_one' is not defined (the reference's 96-wide branch, align_trans.py:62, is broken; "
                        "use warp_faces(frame, alignment_matrices(landmarks, COORD5POINT_96), 112, 96))")
    if isinstance(img, torch.Tensor):
        frame = img
    else:
        frame = torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint8)).to(device if device is not None else "cuda")
    crops = align_faces(frame, landmarks).cpu().numpy()
    return [crops[i] for i in range(crops.shape[0])]
