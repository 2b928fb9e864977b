"""TitaNet with the reference's ``nn.Module`` surface, executed by libtitanet_amd.so on MI355X.

Mirrors (names, arguments, return values, ``state_dict`` keys, assertion messages):
  * ``TitaNet.__init__``           reference src/models.py:175-219
  * ``TitaNet.get_n_params``       reference src/models.py:221-228
  * ``TitaNet.find_n_mega_blocks`` reference src/models.py:230-260
  * ``TitaNet.get_titanet``        reference src/models.py:262-316
  * ``TitaNet.forward``            reference src/models.py:318-339

Host-side design (MI355X-first): every learnable tensor is a view into ONE flat float32 device
buffer laid out by the native library (``tn_model_tensor_info``), gradients live in one flat buffer
of the same layout (one RCCL all-reduce / one fused Adam launch over it), BatchNorm running
statistics in a third.  ``forward`` is a single C call that enqueues every kernel of the step on
the current HIP stream; ``loss.backward()`` is a single C call writing the flat gradient buffer.
There is no CPU fallback: tensors must live on a ROCm device to run ``forward``.
"""
import ctypes as C
import math
from collections import OrderedDict
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import _lib, losses
from ._lib import TnConfig, check


class _Node(nn.Module):
    """Plain container node of the mirrored module tree (encoder.prolog.conv_block.0 ...)."""

    def forward(self, *a, **k):
        raise RuntimeError("sub-modules of titanet_amd.TitaNet are parameter containers; call the TitaNet module")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _Plan:
    def __init__(self, handle, workspace, key):
        self.handle, self.workspace, self.key = handle, workspace, key
        self.generation = 0          # forwards run on this plan (a backward belongs to exactly one of them)


class PackedSpectrograms:
    """A batch the mel front end wrote straight into a plan's prolog operand (``MelSpectrogram.batch(..., into=model)``:
    bf16 rows x n_mels, no float32 ``[B, n_mels, T]`` tensor).  ``model(packed, speakers)`` consumes it; ``lengths`` (valid
    frames per utterance) ride along and become the padding mask of a ragged batch."""

    requires_grad = False

    def __init__(self, plan, generation, batch, n_mels, frames, lengths, device):
        self.plan, self.generation = plan, generation
        self.shape = (batch, n_mels, frames)
        self.lengths, self.device = lengths, device
        self.is_cuda = True

    def dim(self):
        return 3


class _TitaNetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward = tn_forward, backward = tn_backward."""

    @staticmethod
    def forward(ctx, spectrograms, anchor, module, speakers, lengths=None):
        emb, preds, loss, plan = module._native_forward(spectrograms, speakers, lengths=lengths)
        ctx.module, ctx.plan, ctx.generation = module, plan, plan.generation
        ctx.in_shape = spectrograms.shape if (isinstance(spectrograms, torch.Tensor) and spectrograms.requires_grad) else None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(preds)
        return emb, preds, loss

    @staticmethod
    def backward(ctx, g_emb, g_preds, g_loss):
        module, plan = ctx.module, ctx.plan
        # forwards of OTHER plans (another shape, or the other mode: a validation forward under model.eval() between
        # `loss = model(...)` and `loss.backward()`, as the reference allows) leave this plan's saved state alone
        if plan.handle is None or plan.generation != ctx.generation:
            raise RuntimeError("titanet_amd: the saved activations of this forward() are gone (another forward of the same "
                               "shape and mode ran on this module, or its plan was evicted, before backward())")
        g_in = module._native_backward(plan, g_emb, g_loss, ctx.in_shape)
        return g_in, None, None, None, None


class TitaNet(nn.Module):
    """TitaNet speaker-embedding network (reference src/models.py:162-339) on hand-written HIP kernels."""

    TARGET_PARAMS = {"s": 6.4, "m": 13.4, "l": 25.3}
    MAX_PLANS = 6

    def __init__(
        self,
        n_mels,
        n_mega_blocks,
        n_sub_blocks,
        encoder_hidden_size,
        encoder_output_size,
        embedding_size,
        mega_block_kernel_size,
        prolog_kernel_size=3,
        epilog_kernel_size=1,
        attention_hidden_size=128,
        se_reduction=16,
        simple_pool=False,
        loss_function=None,
        dropout=0.5,
        device="cpu",
        precision="fp32",
    ):
        super().__init__()
        assert isinstance(loss_function, losses.MetricLearningLoss) or loss_function is None, "Unsupported loss function"
        self.precision = precision
        self._lib = _lib.load()
        cfg = TnConfig()
        cfg.n_mels, cfg.n_mega_blocks, cfg.n_sub_blocks = n_mels, n_mega_blocks, n_sub_blocks
        cfg.hidden, cfg.enc_out, cfg.emb = encoder_hidden_size, encoder_output_size, embedding_size
        cfg.kernel, cfg.prolog_kernel, cfg.epilog_kernel = mega_block_kernel_size, prolog_kernel_size, epilog_kernel_size
        cfg.attn_hidden, cfg.se_reduction = attention_hidden_size, se_reduction
        cfg.dropout = float(dropout)
        cfg.simple_pool = 1 if simple_pool else 0
        cfg.loss_type, cfg.n_classes = _lib.TN_LOSS_NONE, 0
        cfg.has_scale, cfg.scale, cfg.m1, cfg.m2, cfg.m3, cfg.loss_eps = 1, 1.0, 1.0, 0.0, 0.0, 1e-6
        if loss_function is not None:
            lt, hs, sc, m1, m2, m3, eps = loss_function.native_config()
            cfg.loss_type, cfg.n_classes = lt, int(loss_function.n_classes)
            cfg.has_scale, cfg.scale, cfg.m1, cfg.m2, cfg.m3, cfg.loss_eps = hs, sc, m1, m2, m3, eps
            assert loss_function.embedding_size == embedding_size, "loss embedding size mismatch"
        self._cfg = cfg
        self.dropout = float(dropout)
        handle = C.c_void_p()
        check(self._lib.tn_model_create(C.byref(cfg), C.byref(handle)), "tn_model_create")
        self._model = handle
        self._plans = OrderedDict()
        self._active_plan = None
        self._step = 0
        self._seed_base = int(torch.initial_seed()) & 0xFFFFFFFFFFFF
        self._opt_state = None
        import os
        self.grad_groups = 1   # > 1 (set by the data-parallel Trainer): tn_backward finalises the gradient in 1 + grad_groups buckets (data-parallel overlap)

        # ---- flat storage + mirrored module tree
        dev = torch.device(device)
        n_par = int(self._lib.tn_model_param_floats(handle))
        n_buf = int(self._lib.tn_model_buffer_floats(handle))
        n_bn = int(self._lib.tn_model_num_bn(handle))
        self._flat = {
            "params": torch.zeros(n_par, dtype=torch.float32, device=dev),
            "bnbuf": torch.zeros(n_buf, dtype=torch.float32, device=dev),
            "nbt": torch.zeros(max(n_bn, 1), dtype=torch.int64, device=dev),
        }
        self._flat_grad = None       # accumulated .grad storage (views handed to the parameters)
        self._flat_gtmp = None       # what tn_backward writes (overwrite semantics)
        self._layout = []            # (name, kind, offset, numel, shape)
        name_buf = C.create_string_buffer(160)
        kind, ndim = C.c_int32(), C.c_int32()
        off, numel = C.c_int64(), C.c_int64()
        shape = (C.c_int64 * 4)()
        for i in range(int(self._lib.tn_model_num_tensors(handle))):
            check(self._lib.tn_model_tensor_info(handle, i, name_buf, C.byref(kind), C.byref(off), C.byref(numel),
                                                 C.byref(ndim), shape), "tn_model_tensor_info")
            self._layout.append((name_buf.value.decode(), kind.value, off.value, numel.value,
                                 tuple(int(shape[d]) for d in range(ndim.value))))
        self.encoder, self.decoder = _Node(), _Node()
        self.loss_function = loss_function
        self._build_tree(loss_function)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)

    # ------------------------------------------------------------------ tree / storage
    def _view(self, kind, offset, numel, shape):
        flat = self._flat[("params", "bnbuf", "nbt")[kind]]
        return flat[offset:offset + numel].view(shape)

    def _build_tree(self, loss_function):
        weights = {}
        for name, kind, off, numel, shape in self._layout:
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if not hasattr(node, p) or getattr(node, p) is None:
                    node.add_module(p, _Node())
                node = getattr(node, p)
            leaf = parts[-1]
            view = self._view(kind, off, numel, shape)
            if parts[0] == "loss_function":
                # adopt the loss head's existing parameters (values kept, storage moved into the flat buffer)
                param = getattr(node, leaf)
                with torch.no_grad():
                    view.copy_(param.detach().to(view.device))
                param.data = view
                continue
            if kind == _lib.TN_KIND_PARAM:
                node.register_parameter(leaf, nn.Parameter(view))
                weights[name] = view
            else:
                node.register_buffer(leaf, view)
        # ---- default torch initialisation (nn.Conv1d / nn.Linear / nn.BatchNorm1d defaults)
        with torch.no_grad():
            for name, kind, off, numel, shape in self._layout:
                if name.startswith("loss_function."):
                    continue
                v = self._view(kind, off, numel, shape)
                leaf = name.rsplit(".", 1)[-1]
                if kind == _lib.TN_KIND_NBT:
                    v.zero_()
                elif leaf == "running_mean":
                    v.zero_()
                elif leaf == "running_var":
                    v.fill_(1.0)
                elif len(shape) == 1 and (name[:-len(leaf)] + "running_mean") in self._names():
                    v.fill_(1.0) if leaf == "weight" else v.zero_()     # BatchNorm affine
                elif leaf == "weight":
                    nn.init.kaiming_uniform_(v, a=math.sqrt(5))
                elif leaf == "bias":
                    w = weights[name[:-4] + "weight"]
                    fan_in = int(np.prod(w.shape[1:]))
                    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
                    v.uniform_(-bound, bound)

    def _names(self):
        if not hasattr(self, "_name_set"):
            self._name_set = {n for n, *_ in self._layout}
        return self._name_set

    def _apply(self, fn, recurse=True):
        """Keep the flat-buffer layout across .to() / .cuda() / .cpu()."""
        new = {k: fn(v) for k, v in self._flat.items()}
        if new["params"].dtype != torch.float32:
            raise TypeError("titanet_amd.TitaNet keeps float32 master weights; select precision='bf16' for bf16 compute")
        self._flat = new
        self._flat_grad = self._flat_gtmp = None
        self._opt_state = None
        self._drop_plans()
        mods = dict(self.named_modules())
        for name, kind, off, numel, shape in self._layout:
            mod_name, _, leaf = name.rpartition(".")
            mod = mods[mod_name]
            view = self._view(kind, off, numel, shape)
            if kind == _lib.TN_KIND_PARAM:
                p = mod._parameters[leaf]
                p.data = view
                p.grad = None
            else:
                mod._buffers[leaf] = view
        self._anchor = torch.zeros(1, device=new["params"].device, requires_grad=True)
        return self

    def __del__(self):
        try:
            self._drop_plans()
            if getattr(self, "_model", None):
                self._lib.tn_model_destroy(self._model)
                self._model = None
        except Exception:
            pass

    def _drop_plans(self):
        for plan in getattr(self, "_plans", {}).values():
            self._lib.tn_plan_destroy(plan.handle)
            plan.handle = None
        self._plans = OrderedDict()
        self._active_plan = None

    # ------------------------------------------------------------------ reference API
    def get_n_params(self, div=1):
        """reference src/models.py:221-228"""
        return sum([np.prod(p.size()) for p in self.parameters() if p.requires_grad]) / div

    @classmethod
    def find_n_mega_blocks(cls, embedding_size, n_mels, model_size, loss_function=None, n_mega_blocks_trials=None):
        """reference src/models.py:230-260"""
        if n_mega_blocks_trials is None:
            n_mega_blocks_trials = list(range(1, 20))
        target_params = cls.TARGET_PARAMS[model_size]
        best_value, min_distance = None, np.inf
        for n_mega_blocks in n_mega_blocks_trials:
            titanet = cls.get_titanet(embedding_size=embedding_size, n_mels=n_mels, n_mega_blocks=n_mega_blocks,
                                      model_size=model_size, loss_function=loss_function)
            params = titanet.get_n_params(div=1e6)
            distance = abs(target_params - params)
            if distance < min_distance:
                best_value = n_mega_blocks
                min_distance = distance
        return best_value

    @classmethod
    def get_titanet(cls, embedding_size=192, n_mels=80, n_mega_blocks=None, model_size="s", attention_hidden_size=128,
                    simple_pool=False, loss_function=None, dropout=0.5, device="cpu", precision="fp32"):
        """reference src/models.py:262-316"""
        assert isinstance(model_size, str) and model_size.lower() in ("s", "m", "l"), "Unsupported model size"
        assert isinstance(loss_function, losses.MetricLearningLoss) or loss_function is None, "Unsupported loss function"
        if n_mega_blocks is None:
            n_mega_blocks = cls.find_n_mega_blocks(embedding_size, n_mels, model_size, loss_function=loss_function)
        titanet = partial(cls, n_mels=n_mels, n_mega_blocks=n_mega_blocks, n_sub_blocks=3, encoder_output_size=1536,
                          embedding_size=embedding_size, attention_hidden_size=attention_hidden_size,
                          simple_pool=simple_pool, loss_function=loss_function, dropout=dropout, device=device,
                          precision=precision)
        if model_size.lower() == "s":
            return titanet(encoder_hidden_size=256, mega_block_kernel_size=3)
        elif model_size.lower() == "m":
            return titanet(encoder_hidden_size=512, mega_block_kernel_size=7)
        elif model_size.lower() == "l":
            return titanet(encoder_hidden_size=1024, mega_block_kernel_size=11)

    def forward(self, spectrograms, speakers=None, lengths=None):
        """reference src/models.py:318-339: [B, M, T] -> normalised embeddings [B, E] (inference) or
        (normalised embeddings, predictions, loss) when ``speakers`` is given.

        ``lengths`` (extension; the reference's callers drop collate_fn's lengths, src/learn.py:88): int64 ``[B]`` valid
        frames per utterance of a zero-padded batch — padded frames are then masked out of every layer, statistic and
        gradient (``tn_forward_masked``, include/titanet_amd.h); ``None`` = the reference's semantics."""
        if speakers is not None:
            assert self.loss_function is not None, "Loss function should not be None in training mode"
        needs_grad = torch.is_grad_enabled() and (bool(getattr(spectrograms, "requires_grad", False)) or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            if self._anchor.device != spectrograms.device:
                self._anchor = torch.zeros(1, device=spectrograms.device, requires_grad=True)
            emb, preds, loss = _TitaNetFunction.apply(spectrograms, self._anchor, self, speakers, lengths)
        else:
            emb, preds, loss, _ = self._native_forward(spectrograms, speakers, lengths=lengths)
        if speakers is None:
            return emb
        return emb, preds, loss

    # ------------------------------------------------------------------ native calls
    def _prec(self):
        return {"fp32": _lib.TN_PREC_FP32, "bf16": _lib.TN_PREC_BF16, "fp8": _lib.TN_PREC_FP8, "fp8_fwd": _lib.TN_PREC_FP8_FWD}[self.precision]

    def _get_plan(self, batch, frames):
        key = (batch, frames, self._prec(), bool(self.training))      # train and eval forwards never share saved state
        plan = self._plans.get(key)
        if plan is not None:
            self._plans.move_to_end(key)
            return plan
        flat = self._flat["params"]
        if not flat.is_cuda:
            raise RuntimeError("titanet_amd.TitaNet.forward needs the module on a ROCm device (model.to('cuda')); "
                               "there is no CPU execution path")
        while len(self._plans) >= self.MAX_PLANS:
            _, old = self._plans.popitem(last=False)
            if old is self._active_plan:
                self._active_plan = None
            self._lib.tn_plan_destroy(old.handle)
            old.handle = None
        handle = C.c_void_p()
        check(self._lib.tn_plan_create(self._model, batch, frames, self._prec(), C.byref(handle)), "tn_plan_create")
        if self.grad_groups > 1:
            check(self._lib.tn_plan_set_grad_groups(handle, int(self.grad_groups)), "tn_plan_set_grad_groups")
        nbytes = int(self._lib.tn_plan_workspace_bytes(handle))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
        if self._flat_gtmp is None:
            self._flat_gtmp = torch.zeros_like(flat)
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        check(self._lib.tn_plan_bind(handle, _ptr(flat), _ptr(self._flat_gtmp), _ptr(self._flat["bnbuf"]),
                                     _ptr(self._flat["nbt"]), _ptr(ws), nbytes, C.c_void_p(stream)), "tn_plan_bind")
        plan = _Plan(handle, ws, key)
        self._read_buckets(plan)
        self._plans[key] = plan
        return plan

    def _read_buckets(self, plan):
        """gradient buckets of the plan in completion order (include/titanet_amd.h): [(begin, end)] float ranges"""
        plan.buckets = []
        lo, hi = C.c_int64(), C.c_int64()
        for i in range(int(self._lib.tn_plan_num_grad_buckets(plan.handle))):
            check(self._lib.tn_plan_grad_bucket(plan.handle, i, C.byref(lo), C.byref(hi)), "tn_plan_grad_bucket")
            plan.buckets.append((lo.value, hi.value))

    def _native_forward(self, spectrograms, speakers, fixed_seed=False, lengths=None):
        if spectrograms.dim() != 3 or spectrograms.shape[1] != self._cfg.n_mels:
            raise ValueError(f"expected spectrograms of shape [B, {self._cfg.n_mels}, T], got {tuple(spectrograms.shape)}")
        if not spectrograms.is_cuda:
            raise RuntimeError("titanet_amd.TitaNet.forward needs ROCm device tensors; there is no CPU execution path")
        packed = spectrograms if isinstance(spectrograms, PackedSpectrograms) else None
        if packed is not None:
            x = None
            if lengths is None:
                lengths = packed.lengths
        else:
            x = spectrograms.detach()
            if x.dtype != torch.float32 or not x.is_contiguous():
                x = x.contiguous().float()
        B, _, T = spectrograms.shape
        if self.training and B < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(spectrograms.shape)}")
        plan = self._get_plan(B, T)
        if packed is not None and (packed.plan is not plan or plan.handle is None or packed.generation != plan.generation):
            raise RuntimeError("titanet_amd: this PackedSpectrograms batch was written for another plan (the model changed mode, "
                               "precision or ran another batch of this shape since MelSpectrogram.batch(..., into=model))")
        dev = spectrograms.device
        emb = torch.empty(B, self._cfg.emb, dtype=torch.float32, device=dev)
        preds = loss = y = None
        if speakers is not None:
            y = speakers.detach().to(device=dev, dtype=torch.int64).contiguous()
            preds = torch.empty(B, dtype=torch.int64, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
        if fixed_seed:      # the per-step variation comes from the plan's device-resident step word (trainer graph mode)
            seed = self._seed_base & 0xFFFFFFFFFFFFFFFF
        else:
            seed = (self._seed_base + self._step * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            self._step += 1
        stream = torch.cuda.current_stream(dev).cuda_stream
        if packed is not None:
            ln = None
            if lengths is not None:
                ln = torch.as_tensor(lengths).detach().to(device="cpu", dtype=torch.int64).contiguous()
                if ln.numel() != B or int(ln.min()) < 1 or int(ln.max()) > T:
                    raise ValueError(f"lengths must be {B} values in [1, {T}]")
                if int(ln.min()) == T:
                    ln = None
            check(self._lib.tn_forward_prepacked(plan.handle, C.c_void_p(ln.data_ptr() if ln is not None else 0), _ptr(y),
                                                 1 if self.training else 0, C.c_uint64(seed), _ptr(emb), _ptr(preds), _ptr(loss),
                                                 C.c_void_p(stream)), "tn_forward_prepacked")
        elif lengths is None:
            check(self._lib.tn_forward(plan.handle, _ptr(x), _ptr(y), 1 if self.training else 0, C.c_uint64(seed), _ptr(emb),
                                       _ptr(preds), _ptr(loss), C.c_void_p(stream)), "tn_forward")
        else:
            ln = torch.as_tensor(lengths).detach().to(device="cpu", dtype=torch.int64).contiguous()    # host lengths (collate_fn)
            if ln.numel() != B or int(ln.min()) < 1 or int(ln.max()) > T:
                raise ValueError(f"lengths must be {B} values in [1, {T}], got {tuple(ln.tolist())[:8]}...")
            check(self._lib.tn_forward_masked(plan.handle, _ptr(x), C.c_void_p(ln.data_ptr()), _ptr(y), 1 if self.training else 0,
                                              C.c_uint64(seed), _ptr(emb), _ptr(preds), _ptr(loss), C.c_void_p(stream)),
                  "tn_forward_masked")
        plan.input_ref = x      # keep the input alive for backward (prolog weight gradient re-reads it)
        plan.generation += 1
        self._active_plan = plan
        if speakers is None:
            preds = torch.empty(0, dtype=torch.int64, device=dev)
            loss = torch.zeros((), dtype=torch.float32, device=dev)
        return emb, preds, loss, plan

    def _native_backward(self, plan, g_emb, g_loss, in_shape):
        dev = self._flat["params"].device
        stream = torch.cuda.current_stream(dev).cuda_stream
        g_in = torch.empty(in_shape, dtype=torch.float32, device=dev) if in_shape is not None else None
        if g_emb is not None:
            g_emb = g_emb.contiguous().float()
        scale = 1.0
        if g_loss is None:
            scale = 0.0
        else:
            g_loss = g_loss.contiguous().float()
        check(self._lib.tn_backward(plan.handle, C.c_float(scale), _ptr(g_loss), _ptr(g_emb), _ptr(g_in),
                                    C.c_void_p(stream)), "tn_backward")
        # ---- hand the flat gradient to the parameters with torch's accumulate semantics
        first = next(self.parameters())
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self._flat["params"])
        if first.grad is None or first.grad.data_ptr() != self._flat_grad.data_ptr() + 0:
            self._flat_grad.copy_(self._flat_gtmp)
            self._attach_grads()
        else:
            self._flat_grad.add_(self._flat_gtmp)
        return g_in

    def _attach_grads(self):
        mods = dict(self.named_modules())
        g = self._flat_grad
        for name, kind, off, numel, shape in self._layout:
            if kind != _lib.TN_KIND_PARAM:
                continue
            mod_name, _, leaf = name.rpartition(".")
            mods[mod_name]._parameters[leaf].grad = g[off:off + numel].view(shape)

    # ------------------------------------------------------------------ introspection (tests)
    def debug_fetch(self, what, shape):
        plan = self._active_plan
        assert plan is not None, "run forward first"
        out = torch.empty(shape, dtype=torch.float32, device=self._flat["params"].device)
        stream = torch.cuda.current_stream(out.device).cuda_stream
        check(self._lib.tn_debug_fetch(plan.handle, what.encode(), _ptr(out), out.numel(), C.c_void_p(stream)),
              f"tn_debug_fetch({what})")
        return out

    def flat_parameters(self):
        """The single float32 buffer holding every learnable tensor (state_dict order)."""
        return self._flat["params"]

    def flat_gradients(self):
        """The flat buffer the last backward wrote (overwrite semantics; what DP all-reduces)."""
        return self._flat_gtmp
