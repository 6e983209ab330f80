"""CPU oracle for the face alignment step.  TEST INFRASTRUCTURE ONLY.

Restates ``Mobilefacenet/utils/align_trans.py:16-69`` of the reference: the 5-landmark similarity
transform (``transformation_from_points``, :16-29, numpy float64, same expressions) and the crop
``cv2.warpAffine(img, M[:2], (W, H))[0:112, 0:112]`` (:51-53).

PARITY UNPINNED for the warp: cv2 is not installed in this environment and the reference holds no
aligned-crop fixtures, so ``warp_affine`` below restates OpenCV's published algorithm for
INTER_LINEAR / BORDER_CONSTANT(0) on 8-bit images (imgproc/src/imgwarp.cpp, ``warpAffine`` +
``remapBilinear``: the matrix is inverted in double precision, source coordinates are tracked in
fixed point with 10 fractional bits and rounded to 1/32 pixel, the four bilinear weights are
(32-fx)(32-fy)*32 ... as 16-bit integers summing to 2^15 (the one weight that would be 2^15 itself
saturates to 32767), and the result is (sum + 2^14) >> 15) without having been checked against cv2.
``transformation_from_points`` is plain numpy and is pinned by its own known answers in the tests.
Nothing under ``face_recognition_amd/`` may import this file.
"""
from __future__ import annotations

import numpy as np

TEMPLATE_SQUARE = [[38.29459953, 51.69630051], [73.53179932, 51.50139999], [56.02519989, 71.73660278],
                   [41.54930115, 92.3655014], [70.72990036, 92.20410156]]          # align_trans.py:42-46


def transformation_from_points(points1, points2) -> np.ndarray:
    """align_trans.py:16-29 with ndarray products written out (the reference uses np.matrix, where * is matmul)."""
    points1 = np.asarray(points1, dtype=np.float64).copy()
    points2 = np.asarray(points2, dtype=np.float64).copy()
    c1 = np.mean(points1, axis=0)
    c2 = np.mean(points2, axis=0)
    points1 -= c1
    points2 -= c2
    s1 = np.std(points1)
    s2 = np.std(points2)
    points1 /= s1
    points2 /= s2
    U, S, Vt = np.linalg.svd(points1.T @ points2)
    R = (U @ Vt).T
    top = np.hstack(((s2 / s1) * R, (c2.reshape(2, 1) - (s2 / s1) * R @ c1.reshape(2, 1))))
    return np.vstack([top, [0.0, 0.0, 1.0]])


def landmarks_to_matrix(landmark10) -> np.ndarray:
    """One row of MTCNN landmarks [x1..x5, y1..y5] -> the 2x3 matrix handed to warpAffine (:38-50)."""
    pts1 = np.asarray(landmark10, dtype=np.float64).reshape(2, 5).T
    return transformation_from_points(pts1, np.asarray(TEMPLATE_SQUARE, dtype=np.float64))[:2]


def invert_affine(M) -> np.ndarray:
    """The double-precision inversion warpAffine applies when WARP_INVERSE_MAP is not set."""
    m = np.asarray(M, dtype=np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2] = b1; m[5] = b2
    return m


def _round_int(v):
    return np.rint(v).astype(np.int64)          # saturate_cast<int>(double): round half to even


def warp_affine(img: np.ndarray, M, out_h: int = 112, out_w: int = 112) -> np.ndarray:
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C]: the top-left out_h x out_w window of warpAffine(img, M, (W,H))."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, C = img.shape
    m = invert_affine(M)
    AB_SCALE = 1 << 10
    xs = np.arange(out_w, dtype=np.float64)
    ys = np.arange(out_h, dtype=np.float64)
    adelta = _round_int(m[0] * xs * AB_SCALE)
    bdelta = _round_int(m[3] * xs * AB_SCALE)
    X0 = _round_int((m[1] * ys + m[2]) * AB_SCALE) + 16
    Y0 = _round_int((m[4] * ys + m[5]) * AB_SCALE) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5          # 1/32 pixel units
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00 = np.minimum((32 - fx) * (32 - fy) * 32, 32767)
    w01 = fx * (32 - fy) * 32
    w10 = (32 - fx) * fy * 32
    w11 = fx * fy * 32

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return v * ok[..., None]

    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] +
           tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None])
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def face_alignment(img: np.ndarray, landmarks: np.ndarray):
    """Face_alignment(img, True, landmarks) (:31-54): list of 112x112 crops, one per landmark row."""
    return [warp_affine(img, landmarks_to_matrix(lm)) for lm in np.asarray(landmarks).reshape(-1, 10)]
