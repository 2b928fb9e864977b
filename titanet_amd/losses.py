"""Loss heads with the reference's constructor surface (reference src/losses.py:7-183, :264-270).

The classes own the final ``fc`` parameters (same ``state_dict`` keys as the reference:
``loss_function.fc.weight`` / ``.bias``) and carry the loss hyper-parameters; the arithmetic
(logits / cosines, soft-max, margins, argmax and the backward pass) runs inside the HIP head
kernels of ``libtitanet_amd.so`` when the loss is attached to a :class:`titanet_amd.models.TitaNet`
(``TitaNet.forward(spectrograms, speakers)``, reference src/models.py:339).
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _HeadFunction(torch.autograd.Function):
    """``loss(inputs, targets)`` outside TitaNet.forward: tn_head_forward / tn_head_backward (include/titanet_amd.h)."""

    @staticmethod
    def forward(ctx, inputs, weight, bias, head, targets):
        lib = _lib.load()
        if not inputs.is_cuda:
            raise RuntimeError("titanet_amd loss heads need ROCm device tensors; there is no CPU execution path")
        x = inputs.detach().contiguous().float()
        B, E = x.shape
        NC = int(weight.shape[0])
        y = targets.detach().to(device=x.device, dtype=torch.int64).contiguous()
        lt, hs, sc, m1, m2, m3, eps = head.native_config()
        norm = torch.empty_like(x)
        preds = torch.empty(B, dtype=torch.int64, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        save = torch.empty(int(lib.tn_head_save_floats(B, E, NC)), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(lib.tn_head_forward(lt, B, E, NC, _p(x), _p(y), _p(weight.data), _p(bias.data if bias is not None else None),
                                       hs, sc, m1, m2, m3, eps, _p(norm), _p(preds), _p(loss), _p(save), C.c_void_p(stream)),
                   "tn_head_forward")
        ctx.cfg, ctx.save, ctx.weight, ctx.has_bias = (lt, B, E, NC), save, weight, bias is not None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(preds)
        return norm, preds, loss

    @staticmethod
    def backward(ctx, g_norm, g_preds, g_loss):
        lib = _lib.load()
        lt, B, E, NC = ctx.cfg
        dev = ctx.save.device
        g_in = torch.empty(B, E, dtype=torch.float32, device=dev)
        g_w = torch.empty(NC, E, dtype=torch.float32, device=dev)
        g_b = torch.empty(NC, dtype=torch.float32, device=dev) if ctx.has_bias else None
        scale = 1.0
        if g_loss is None:
            scale = 0.0
        else:
            g_loss = g_loss.contiguous().float()
        if g_norm is not None:
            g_norm = g_norm.contiguous().float()
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.tn_head_backward(lt, B, E, NC, _p(ctx.weight.data), _p(ctx.save), C.c_float(scale), _p(g_loss), _p(g_norm),
                                        _p(g_in), _p(g_w), _p(g_b), C.c_void_p(stream)), "tn_head_backward")
        return g_in, g_w, g_b, None, None


class _FC(nn.Module):
    """Parameter holder laid out like ``nn.Linear`` (weight [C, E], optional bias [C])."""

    def __init__(self, in_features, out_features, bias=True, device="cpu"):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        w = torch.empty(out_features, in_features, device=device)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # nn.Linear default init
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1.0 / math.sqrt(in_features)
            self.bias = nn.Parameter(torch.empty(out_features, device=device).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)


class MetricLearningLoss(nn.Module):
    """reference src/losses.py:7-19"""

    tn_loss_type = _lib.TN_LOSS_NONE

    def __init__(self, embedding_size, n_classes, device="cpu", *args, **kwargs):
        super().__init__()
        self.embedding_size = embedding_size
        self.n_classes = n_classes
        self.device = device

    def forward(self, inputs, targets):
        """reference src/losses.py:32-44 / :77-132: (normalised inputs, predictions, loss).  Inside
        ``TitaNet.forward(spectrograms, speakers)`` the same head kernels run fused with the decoder tail; this entry
        is the stand-alone call of the reference's loss objects."""
        if not hasattr(self, "fc"):
            raise NotImplementedError
        return _HeadFunction.apply(inputs, self.fc.weight, self.fc.bias, self, targets)

    def native_config(self):
        """(loss_type, has_scale, scale, m1, m2, m3, eps) for tn_config."""
        return (self.tn_loss_type, 1, 1.0, 1.0, 0.0, 0.0, 1e-6)


class CELoss(MetricLearningLoss):
    """reference src/losses.py:22-44"""

    tn_loss_type = _lib.TN_LOSS_CE

    def __init__(self, embedding_size, n_classes, device="cpu"):
        super().__init__(embedding_size, n_classes, device=device)
        self.fc = _FC(embedding_size, n_classes, bias=True, device=device)


class AngularMarginLoss(MetricLearningLoss):
    """reference src/losses.py:47-132"""

    tn_loss_type = _lib.TN_LOSS_MARGIN

    def __init__(self, embedding_size, n_classes, device="cpu", scale=None, m1=1, m2=0, m3=0, eps=1e-6):
        super().__init__(embedding_size, n_classes, device=device)
        self.fc = _FC(embedding_size, n_classes, bias=False, device=device)
        self.scale, self.m1, self.m2, self.m3, self.eps = scale, m1, m2, m3, eps

    def native_config(self):
        return (self.tn_loss_type, 0 if self.scale is None else 1, 0.0 if self.scale is None else float(self.scale),
                float(self.m1), float(self.m2), float(self.m3), float(self.eps))


class SphereFaceLoss(AngularMarginLoss):
    """reference src/losses.py:135-148"""

    def __init__(self, embedding_size, n_classes, device="cpu", scale=None, margin=3, eps=1e-6):
        assert margin > 1, "Margin out of bounds"
        super().__init__(embedding_size, n_classes, device=device, scale=scale, m1=margin, eps=eps)


class CosFaceLoss(AngularMarginLoss):
    """reference src/losses.py:151-166"""

    def __init__(self, embedding_size, n_classes, device="cpu", scale=64, margin=0.2, eps=1e-6):
        assert margin > 0 and margin < 1 - np.cos(np.pi / 4), "Margin out of bounds"
        super().__init__(embedding_size, n_classes, device=device, scale=scale, m3=margin, eps=eps)


class ArcFaceLoss(AngularMarginLoss):
    """reference src/losses.py:169-183"""

    def __init__(self, embedding_size, n_classes, device="cpu", scale=64, margin=0.5, eps=1e-6):
        assert margin > 0 and margin < 1, "Margin out of bounds"
        super().__init__(embedding_size, n_classes, device=device, scale=scale, m2=margin, eps=eps)


class GE2ELoss(MetricLearningLoss):
    """reference src/losses.py:186-261 — out of the hot-path scope: not implemented."""

    def __init__(self, embedding_size, n_classes, device="cpu", *a, **k):
        raise NotImplementedError("GE2E is outside the MI355X hot-path scope (SURVEY.md §2)")


# reference src/losses.py:264-270
LOSSES = {
    "ce": CELoss,
    "sphere": SphereFaceLoss,
    "cos": CosFaceLoss,
    "arc": ArcFaceLoss,
    "ge2e": GE2ELoss,
}
