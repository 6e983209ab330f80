"""Drop-in for the reference's ``face_model.MobileFaceNet`` / ``l2_norm`` on MI355X.

``MobileFaceNet(embedding_size)`` is a ``torch.nn.Module`` whose parameter tree carries exactly
the reference's 333 state_dict keys (``Mobilefacenet/face_model.py:77-92``), so checkpoints load
unchanged; ``forward`` (``face_model.py:103-129``) does not run those modules: it hands the input
to the HIP library, which runs the whole network with BatchNorm folded and PReLU / residual /
L2-norm fused into hand-written gfx950 kernels.  Inference (eval-mode) semantics only.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from . import layers as L


def l2_norm(input: torch.Tensor, axis: int = 1) -> torch.Tensor:
    """``input / ||input||_2`` along ``axis``, no epsilon (reference: face_model.py:19-22).

    Contiguous f32 [B,E] tensors on the HIP device with ``axis=1`` use the row-normalisation
    kernel; anything else (CPU tensors, other axes) is computed with the reference's two torch ops.
    """
    if input.is_cuda and input.dtype == torch.float32 and input.dim() == 2 and axis in (1, -1) \
            and input.is_contiguous() and input.numel() > 0:
        lib = _lib.load()
        out = torch.empty_like(input)
        with torch.cuda.device(input.device):
            _lib.check(lib.mfn_tta_combine(input.data_ptr(), None, input.shape[0], input.shape[1], out.data_ptr(),
                                           _lib.current_stream_ptr(input.device)), "mfn_tta_combine")
        return out
    norm = torch.norm(input, 2, axis, True)
    return torch.div(input, norm)


def tta_combine(emb: torch.Tensor, emb_mirror: torch.Tensor) -> torch.Tensor:
    """``l2_norm(emb + emb_mirror)`` in one kernel (facebank.py:62, cam_demo.py:92, Evaluation.py:51-52)."""
    if not (emb.is_cuda and emb_mirror.is_cuda):
        return l2_norm(emb + emb_mirror)
    a = emb.contiguous().float()
    b = emb_mirror.contiguous().float()
    if a.shape != b.shape or a.dim() != 2:
        raise RuntimeError(f"tta_combine: shapes {tuple(a.shape)} and {tuple(b.shape)} differ or are not [B,E]")
    lib = _lib.load()
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.check(lib.mfn_tta_combine(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], out.data_ptr(),
                                       _lib.current_stream_ptr(a.device)), "mfn_tta_combine")
    return out


def set_small_batch_mode(mode: int) -> None:
    """How batches of up to about 36 crops run conv_4 / conv_5 (``fra_set_small_batch_mode``; process wide).
    1 (default): the shortest launches -- the per-frame loop of cam_demo.py:87-105; such a batch's embeddings equal
    those of a larger batch to rounding (cosine 0.99999) and are the same bits for every batch size up to the limit.
    0: an embedding is the same bits whatever batch it is computed in."""
    _lib.check(_lib.load().fra_set_small_batch_mode(int(mode)), "fra_set_small_batch_mode")


def small_batch_mode() -> int:
    return int(_lib.load().fra_get_small_batch_mode())


class _ConvUnit(nn.Module):
    """Parameter holder for conv (+bn, +prelu): keys ``conv.weight``, ``bn.*``, ``prelu.weight``."""

    def __init__(self, spec: L.ConvSpec):
        super().__init__()
        self.conv = nn.Conv2d(spec.cin, spec.cout, spec.k, spec.stride, spec.pad, groups=spec.groups, bias=False)
        self.bn = nn.BatchNorm2d(spec.cout)
        if spec.prelu:
            self.prelu = nn.PReLU(spec.cout)


class _Block(nn.Module):
    def __init__(self, spec: L.BlockSpec):
        super().__init__()
        e, d, p = spec.convs()
        self.conv, self.conv_dw, self.project = _ConvUnit(e), _ConvUnit(d), _ConvUnit(p)


class _Stage(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.model = nn.Sequential(*[_Block(b) for b in blocks])


class MobileFaceNet(nn.Module):
    """MobileFaceNet(embedding_size); ``forward(x)``: f32 [B,3,112,112] (BGR, [-1,1]) on the HIP
    device -> new f32 [B,embedding_size] tensor with unit-norm rows."""

    @property
    def batch_invariant(self) -> bool:
        """True when a face's embedding is the same bits whatever batch it is computed in (every kernel configuration a
        batch size selects then runs the same arithmetic per output value): ``set_small_batch_mode(0)``.  In the default
        mode that holds among batches of at least ``batch_invariant_from()`` crops (and among the smaller ones), and
        callers that regroup batches (Evaluation.getFeature) keep to those."""
        return small_batch_mode() == 0

    def batch_invariant_from(self, device=None) -> int:
        """Smallest batch from which on an embedding does not depend on the batch size, bit for bit."""
        if small_batch_mode() == 0:
            return 1
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        cus = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 256
        return cus // 7 + 1        # the band launches of 7 B <= compute units are the ones the mode changes

    def __init__(self, embedding_size: int = 512, act_dtype: str = "f16"):
        super().__init__()
        self.embedding_size = int(embedding_size)
        self._act_dtype = _lib.DTYPE_BY_NAME[act_dtype]
        blocks: Dict[str, list] = {}
        for b in L.BLOCKS:
            blocks.setdefault(b.prefix.split(".")[0], []).append(b)
        self.conv1 = _ConvUnit(L.STEM)
        self.conv2_dw = _ConvUnit(L.CONV2_DW)
        self.conv_23 = _Block(blocks["conv_23"][0])
        self.conv_3 = _Stage(blocks["conv_3"])
        self.conv_34 = _Block(blocks["conv_34"][0])
        self.conv_4 = _Stage(blocks["conv_4"])
        self.conv_45 = _Block(blocks["conv_45"][0])
        self.conv_5 = _Stage(blocks["conv_5"])
        self.conv_6_sep = _ConvUnit(L.CONV6_SEP)
        self.conv_6_dw = _ConvUnit(L.CONV6_DW)
        self.linear = nn.Linear(L.FEAT_C, self.embedding_size, bias=False)
        self.bn = nn.BatchNorm1d(self.embedding_size)
        # same initial distribution as the reference (face_model.py:95-101)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        # native state (not part of the state_dict)
        self._handle: Optional[int] = None
        self._handle_device: Optional[torch.device] = None
        self._dirty = True
        self._workspace: Optional[torch.Tensor] = None
        self._batch_chunk = 0
        self._fused = True
        self._chain = True

    # ---- weight life cycle --------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):
        self._dirty = True
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        """Accepts the reference's checkpoints as they are written: a plain state_dict, keys with
        the ``module.`` prefix nn.DataParallel adds (test_lfw_agedb_cfpfp.py:63-67), or the training
        wrapper ``{'net_state_dict': ...}`` (Train_mobile.py:167-176)."""
        if "net_state_dict" in state_dict and not any(k.endswith(".weight") for k in state_dict):
            state_dict = state_dict["net_state_dict"]
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        self._dirty = True
        return super().load_state_dict(state_dict, strict=strict, **kwargs)

    def refresh_weights(self) -> None:
        """Call after modifying parameters in place; load_state_dict / .to() do this themselves."""
        self._dirty = True

    def set_batch_chunk(self, images: int) -> None:
        """Process large batches in slices of ``images`` (0 = whole batch)."""
        self._batch_chunk = int(images)
        if self._handle is not None:
            _lib.check(_lib.load().mfn_set_batch_chunk(self._handle, self._batch_chunk), "mfn_set_batch_chunk")

    def set_fused(self, fused: bool) -> None:
        """True (default): one kernel per inverted-residual block; False: one kernel per conv."""
        self._fused = bool(fused)
        if self._handle is not None:
            _lib.check(_lib.load().mfn_set_fused(self._handle, int(self._fused)), "mfn_set_fused")

    def set_chain(self, chain: bool) -> None:
        """True (default): runs of identical residual blocks (conv_4) go out as one launch for large batches;
        False: one launch per block.  Results are bit-identical either way."""
        self._chain = bool(chain)
        if self._handle is not None:
            _lib.check(_lib.load().mfn_set_chain(self._handle, int(self._chain)), "mfn_set_chain")

    def launch_plan(self, batch: int):
        """For every op of op_table(): index of the op whose launch covers it in a batch of ``batch`` crops."""
        lib = _lib.load()
        if self._handle is None:
            raise RuntimeError("launch_plan: run a forward first")
        n = lib.mfn_op_count(self._handle)
        first = (C.c_int * n)()
        _lib.check(lib.mfn_launch_plan(self._handle, int(batch), first, n), "mfn_launch_plan")
        return list(first)

    def _ensure_handle(self, device: torch.device):
        lib = _lib.load()
        if self._handle is not None and self._handle_device != device:
            lib.mfn_destroy(self._handle)
            self._handle = None
        if self._handle is None:
            h = C.c_void_p()
            _lib.check(lib.mfn_create(C.byref(h), device.index if device.index is not None else
                                      torch.cuda.current_device(), self.embedding_size, self._act_dtype),
                       "mfn_create")
            self._handle, self._handle_device, self._dirty = h.value, device, True
            _lib.check(lib.mfn_set_batch_chunk(self._handle, self._batch_chunk), "mfn_set_batch_chunk")
            _lib.check(lib.mfn_set_fused(self._handle, int(self._fused)), "mfn_set_fused")
            _lib.check(lib.mfn_set_chain(self._handle, int(getattr(self, "_chain", True))), "mfn_set_chain")
        if self._dirty:
            sd = self.state_dict()
            keep, descs = [], (_lib.TensorDesc * len(sd))()
            n = 0
            for k, v in sd.items():
                if k.endswith("num_batches_tracked"):
                    continue
                a = v.detach().to("cpu", torch.float32).contiguous()
                keep.append(a)
                descs[n].name = k.encode()
                descs[n].data = a.data_ptr()
                descs[n].ndim = a.dim()
                for d in range(a.dim()):
                    descs[n].shape[d] = a.shape[d]
                n += 1
            _lib.check(lib.mfn_load_weights(self._handle, descs, n), "mfn_load_weights")
            self._dirty = False
        return lib

    def _ensure_workspace(self, lib, batch: int, device: torch.device, tta: bool = False) -> torch.Tensor:
        need = C.c_size_t()
        _lib.check(lib.mfn_workspace_bytes_ex(self._handle, batch, int(tta), C.byref(need)), "mfn_workspace_bytes_ex")
        ws = self._workspace
        if ws is None or ws.device != device or ws.numel() < need.value:
            self._workspace = None
            ws = torch.empty(need.value, dtype=torch.uint8, device=device)
            self._workspace = ws
        return ws

    # ---- forward --------------------------------------------------------------------------------
    def _check_input(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("MobileFaceNet (face_recognition_amd) is inference-only: call .eval() first")
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or tuple(x.shape[1:]) != (3, L.INPUT_HW, L.INPUT_HW):
            raise RuntimeError(f"expected input [B,3,112,112], got {tuple(getattr(x, 'shape', ()))}")
        if not x.is_cuda:
            raise RuntimeError("MobileFaceNet (face_recognition_amd) runs on the HIP device only; "
                               "move the model and the input to 'cuda' (there is no CPU path)")
        if x.shape[0] == 0:
            raise RuntimeError("empty batch")
        return x.detach().to(torch.float32).contiguous()

    def workspace_bytes(self, batch: int, device=None, tta: bool = False) -> int:
        """Bytes of scratch a forward of ``batch`` crops needs (for callers that bring their own workspace)."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        with torch.cuda.device(dev):
            lib = self._ensure_handle(dev)
            need = C.c_size_t()
            _lib.check(lib.mfn_workspace_bytes_ex(self._handle, batch, int(tta), C.byref(need)), "mfn_workspace_bytes_ex")
        return int(need.value)

    def forward(self, x: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``workspace``: optional caller-owned uint8 device tensor of at least ``workspace_bytes(B)``
        bytes (256-byte aligned); by default the module keeps and grows its own."""
        x = self._check_input(x)
        dev = x.device
        with torch.cuda.device(dev):
            lib = self._ensure_handle(dev)
            ws = workspace if workspace is not None else self._ensure_workspace(lib, x.shape[0], dev)
            out = torch.empty((x.shape[0], self.embedding_size), dtype=torch.float32, device=dev)
            _lib.check(lib.mfn_forward(self._handle, x.data_ptr(), x.shape[0], out.data_ptr(), ws.data_ptr(),
                                       ws.numel(), _lib.current_stream_ptr(dev)), "mfn_forward")
        return out

    def _forward_ex(self, x: torch.Tensor, fmt: int, tta: bool, workspace: Optional[torch.Tensor]) -> torch.Tensor:
        dev = x.device
        with torch.cuda.device(dev):
            lib = self._ensure_handle(dev)
            ws = workspace if workspace is not None else self._ensure_workspace(lib, x.shape[0], dev, tta)
            out = torch.empty((x.shape[0], self.embedding_size), dtype=torch.float32, device=dev)
            _lib.check(lib.mfn_forward_ex(self._handle, x.data_ptr(), fmt, x.shape[0], int(tta), out.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(dev)), "mfn_forward_ex")
        return out

    def embed_tta(self, x: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``l2_norm(net(x) + net(hflip(x)))`` -- the flip test-time augmentation of facebank.py:59-62.
        With the fused schedule the mirrored forward reads ``x`` backwards inside the first kernel; no
        flipped copy of the input is made."""
        if not self._fused:
            return tta_combine(self.forward(x), self.forward(torch.flip(x, dims=[3])))
        return self._forward_ex(self._check_input(x), 0, True, workspace)

    def forward_u8(self, x_u8: torch.Tensor, tta: bool = False, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Embed raw crops: ``x_u8`` uint8 [B,112,112,3] HWC BGR on the HIP device (what cv2.imread /
        Face_alignment return, stacked).  Equivalent to ``net(test_transform(img))`` per crop with
        ToTensor + Normalize(0.5, 0.5) (facebank.py:22-24) computed inside the first kernel."""
        if self.training:
            raise RuntimeError("MobileFaceNet (face_recognition_amd) is inference-only: call .eval() first")
        if not (isinstance(x_u8, torch.Tensor) and x_u8.dtype == torch.uint8 and x_u8.dim() == 4
                and tuple(x_u8.shape[1:]) == (L.INPUT_HW, L.INPUT_HW, 3)):
            raise RuntimeError(f"expected uint8 [B,112,112,3], got {getattr(x_u8, 'dtype', None)} "
                               f"{tuple(getattr(x_u8, 'shape', ()))}")
        if not x_u8.is_cuda or x_u8.shape[0] == 0:
            raise RuntimeError("forward_u8 needs a non-empty tensor on the HIP device")
        if not self._fused:
            raise RuntimeError("forward_u8 needs the fused schedule (set_fused(True))")
        return self._forward_ex(x_u8.detach().contiguous(), 1, tta, workspace)

    def forward_timed(self, x: torch.Tensor):
        """(embeddings, per-op milliseconds, op table) -- benchmark helper, synchronises."""
        x = self._check_input(x)
        dev = x.device
        with torch.cuda.device(dev):
            lib = self._ensure_handle(dev)
            ws = self._ensure_workspace(lib, x.shape[0], dev)
            out = torch.empty((x.shape[0], self.embedding_size), dtype=torch.float32, device=dev)
            n = lib.mfn_op_count(self._handle)
            ms = (C.c_float * n)()
            _lib.check(lib.mfn_forward_timed(self._handle, x.data_ptr(), x.shape[0], out.data_ptr(), ws.data_ptr(),
                                             ws.numel(), _lib.current_stream_ptr(dev), ms, n), "mfn_forward_timed")
        return out, np.array(list(ms), dtype=np.float64), self.op_table()

    def op_table(self):
        """[(name, kind, flops_per_image, activation_bytes_per_image, weight_bytes)] for every kernel of the forward."""
        lib = _lib.load()
        if self._handle is None:
            raise RuntimeError("op_table: run a forward first")
        kinds = ["stem", "dw3x3", "pw", "gdconv7", "linear", "block", "stemdw", "tail"]
        rows = []
        for i in range(lib.mfn_op_count(self._handle)):
            name = C.create_string_buffer(96)
            kind, fl, by, wb = C.c_int(), C.c_double(), C.c_double(), C.c_double()
            _lib.check(lib.mfn_op_info(self._handle, i, name, 96, C.byref(kind), C.byref(fl), C.byref(by), C.byref(wb)),
                       "mfn_op_info")
            rows.append((name.value.decode(), kinds[kind.value], fl.value, by.value, wb.value))
        return rows

    def op_variants(self, batch: int):
        """For every op of op_table(): 0 = the kernel of its kind, 1 = rblock_kernel, 2 = pblock_kernel (the row
        pipeline) -- what a batch of ``batch`` crops runs."""
        lib = _lib.load()
        if self._handle is None:
            raise RuntimeError("op_variants: run a forward first")
        out = []
        for i in range(lib.mfn_op_count(self._handle)):
            v = lib.mfn_op_variant(self._handle, i, int(batch))
            if v < 0:
                _lib.check(v, "mfn_op_variant")
            out.append(v)
        return out

    def debug_prefix(self, x: torch.Tensor, n_ops: int) -> torch.Tensor:
        """Activation after the first ``n_ops`` kernels as f32 NHWC [B,H,W,C] (test helper)."""
        x = self._check_input(x)
        dev = x.device
        with torch.cuda.device(dev):
            lib = self._ensure_handle(dev)
            ws = self._ensure_workspace(lib, x.shape[0], dev)
            out = torch.empty(x.shape[0] * 56 * 56 * 128, dtype=torch.float32, device=dev)
            per = C.c_int64()
            _lib.check(lib.mfn_debug_prefix(self._handle, x.data_ptr(), x.shape[0], n_ops, out.data_ptr(),
                                            C.byref(per), ws.data_ptr(), ws.numel(),
                                            _lib.current_stream_ptr(dev)), "mfn_debug_prefix")
        return out[: x.shape[0] * per.value].view(x.shape[0], -1)

    def __getstate__(self):
        # native handles are per process and per object: copies and pickles rebuild their own
        d = self.__dict__.copy()
        d.update(_handle=None, _handle_device=None, _workspace=None, _dirty=True)
        return d

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.load().mfn_destroy(self._handle)     # frees under a device guard of its own
                self._handle = None
        except Exception:
            pass
