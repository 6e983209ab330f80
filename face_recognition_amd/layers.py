"""Static description of the MobileFaceNet graph.

One table drives everything that has to agree on the network's shape: the
``nn.Module`` parameter tree (state_dict key names), the synthetic weight
generator and the tests.  The C++ side of the library carries the same table
(``csrc/mfn_model.cpp``) and the loader checks every tensor against it.

Topology follows ``Mobilefacenet/face_model.py:77-92`` of the reference:
stem 3x3/s2, one depthwise 3x3, five inverted-residual stages, a 1x1
expansion to 512, the 7x7 global depthwise conv, and a bias-free linear
layer with BatchNorm1d.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

INPUT_HW = 112
INPUT_C = 3
BN_EPS = 1e-5  # torch.nn.BatchNorm{1,2}d default, used by face_model.py:30,42,92


@dataclass(frozen=True)
class ConvSpec:
    """One conv + BatchNorm2d (+ per-channel PReLU) unit.

    ``prefix`` is the state_dict prefix, e.g. ``conv_3.model.0.conv_dw``.
    ``kind``: 'dense' (stem), 'dw' (depthwise k x k), 'pw' (1x1).
    """
    prefix: str
    kind: str
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    prelu: bool
    hin: int  # input spatial size (square)

    @property
    def groups(self) -> int:
        return self.cin if self.kind == "dw" else 1

    @property
    def hout(self) -> int:
        return (self.hin + 2 * self.pad - self.k) // self.stride + 1

    @property
    def weight_shape(self) -> Tuple[int, int, int, int]:
        return (self.cout, self.cin // self.groups, self.k, self.k)

    @property
    def macs(self) -> int:
        return self.hout * self.hout * self.cout * (self.cin // self.groups) * self.k * self.k


@dataclass(frozen=True)
class BlockSpec:
    """expand 1x1 (PReLU) -> depthwise 3x3 (PReLU) -> project 1x1 (linear) [+ input]."""
    prefix: str
    cin: int
    cexp: int
    cout: int
    stride: int
    residual: bool
    hin: int

    def convs(self) -> List[ConvSpec]:
        hmid = self.hin
        hout = (self.hin + 2 - 3) // self.stride + 1
        return [
            ConvSpec(self.prefix + ".conv", "pw", self.cin, self.cexp, 1, 1, 0, True, hmid),
            ConvSpec(self.prefix + ".conv_dw", "dw", self.cexp, self.cexp, 3, self.stride, 1, True, hmid),
            ConvSpec(self.prefix + ".project", "pw", self.cexp, self.cout, 1, 1, 0, False, hout),
        ]


def _stage(name: str, n: int, c: int, cexp: int, h: int) -> List[BlockSpec]:
    return [BlockSpec(f"{name}.model.{i}", c, cexp, c, 1, True, h) for i in range(n)]


STEM = ConvSpec("conv1", "dense", 3, 64, 3, 2, 1, True, 112)
CONV2_DW = ConvSpec("conv2_dw", "dw", 64, 64, 3, 1, 1, True, 56)
BLOCKS: List[BlockSpec] = (
    [BlockSpec("conv_23", 64, 128, 64, 2, False, 56)]
    + _stage("conv_3", 4, 64, 128, 28)
    + [BlockSpec("conv_34", 64, 256, 128, 2, False, 28)]
    + _stage("conv_4", 6, 128, 256, 14)
    + [BlockSpec("conv_45", 128, 512, 128, 2, False, 14)]
    + _stage("conv_5", 2, 128, 256, 7)
)
CONV6_SEP = ConvSpec("conv_6_sep", "pw", 128, 512, 1, 1, 0, True, 7)
CONV6_DW = ConvSpec("conv_6_dw", "dw", 512, 512, 7, 1, 0, False, 7)
FEAT_C = 512  # input width of the final linear layer


def all_convs() -> List[ConvSpec]:
    """Every conv unit in execution order (49 of them)."""
    out = [STEM, CONV2_DW]
    for b in BLOCKS:
        out.extend(b.convs())
    out.extend([CONV6_SEP, CONV6_DW])
    return out


def state_dict_keys(embedding_size: int = 512) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, dtype) for all 333 entries, in the reference module's registration order."""
    keys: List[Tuple[str, Tuple[int, ...], str]] = []
    for c in all_convs():
        keys.append((c.prefix + ".conv.weight", c.weight_shape, "float32"))
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            keys.append((f"{c.prefix}.bn.{leaf}", (c.cout,), "float32"))
        keys.append((c.prefix + ".bn.num_batches_tracked", (), "int64"))
        if c.prelu:
            keys.append((c.prefix + ".prelu.weight", (c.cout,), "float32"))
    keys.append(("linear.weight", (embedding_size, FEAT_C), "float32"))
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        keys.append((f"bn.{leaf}", (embedding_size,), "float32"))
    keys.append(("bn.num_batches_tracked", (), "int64"))
    return keys


def macs_per_face(embedding_size: int = 512) -> int:
    return sum(c.macs for c in all_convs()) + FEAT_C * embedding_size
