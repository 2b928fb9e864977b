"""Training-step engine: forward + backward + (data-parallel gradient all-reduce) + Adam on the flat
buffers, without going through per-parameter autograd bookkeeping.

Reproduces the reference's step protocol (reference src/learn.py:88-135: ``model(spectrograms,
speakers=...)`` -> ``optimizer.zero_grad(); loss.backward(); optimizer.step()``) and its optimizer
choice (always Adam, lr 1e-3, weight decay 0: reference src/train.py:130-135, parameters.yml:5-10),
as four native calls per step on the current HIP stream.

Data parallelism (new — the reference is single-device): one process per GPU, full replica, the global
batch sharded contiguously over ranks, BatchNorm statistics local to each rank (DDP semantics), the
flat float32 gradient buffer summed with RCCL (``torch.distributed`` backend "nccl" == RCCL on ROCm)
in a few large buckets sized for the xGMI links (fewer, larger collectives), the 1/world factor folded
into the fused Adam kernel.  The collectives run on a side stream and OVERLAP backward: tn_backward finalises the
gradient bucket by bucket from the decoder side down and records an event per bucket (include/titanet_amd.h,
"gradient buckets"); each bucket's all-reduce is enqueued behind its event.

``use_graph=True`` (single GPU): the whole step — ~470 kernel launches for TitaNet-S — is captured once per input shape
into ONE hipGraph and replayed; the per-step state that used to be kernel arguments (dropout stream, Adam step count)
lives in device memory (``tn_plan_step_tick`` / ``tn_adam_step_plan``, include/titanet_amd.h).  Small batches are
launch-bound (batch 8: the reference's own parameters.yml batch), which is where this pays.
"""
import ctypes as C
import time

import torch
import torch.distributed as dist

from ._lib import check


def bucket_ranges(n, n_buckets):
    """Split [0, n) into <= n_buckets contiguous ranges aligned to 1024 elements."""
    if n_buckets <= 1 or n <= 1024:
        return [(0, n)]
    step = ((n + n_buckets - 1) // n_buckets + 1023) // 1024 * 1024
    out, lo = [], 0
    while lo < n:
        hi = min(n, lo + step)
        out.append((lo, hi))
        lo = hi
    return out


class FlatAllReducer:
    """Sum-all-reduce of one flat gradient buffer in buckets (works on any backend; RCCL on GPU)."""

    def __init__(self, n_buckets=4, group=None):
        self.n_buckets, self.group = n_buckets, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._stream = None
        # measure = True: every reduction records (end of backward on the compute stream, end of the last collective on the
        # communication stream) — the part of the all-reduce backward did NOT hide is the time between the two (exposed_ms)
        self.measure = False
        self._marks = []

    def _mark(self, cur, done_stream):
        if not self.measure:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        return e0

    def exposed_ms(self):
        """Per measured step: max(0, last collective's end - backward's end) in ms (synchronises the recorded events)."""
        out = []
        for e0, e1 in self._marks:
            e1.synchronize()
            out.append(max(0.0, e0.elapsed_time(e1)))
        self._marks = []
        return out

    def all_reduce_overlapped_(self, flat, plan, lib):
        """The overlapped form: backward has been ENQUEUED on the current stream and records one event per gradient bucket
        (completion order: decoder side first, include/titanet_amd.h "gradient buckets").  Each bucket's all-reduce is
        enqueued on the communication stream behind its event, so RCCL moves bucket i over xGMI while the kernels of the
        mega blocks below it still run; the current stream only waits for the last collective."""
        if self.world == 1:
            return flat
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=flat.device)
        cur = torch.cuda.current_stream(flat.device)
        with torch.cuda.stream(self._stream):
            for i, (lo, hi) in enumerate(plan.buckets):
                check(lib.tn_plan_wait_grad_bucket(plan.handle, i, C.c_void_p(self._stream.cuda_stream)), "tn_plan_wait_grad_bucket")
                dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            done = torch.cuda.Event(enable_timing=self.measure)
            done.record(self._stream)
        e0 = self._mark(cur, self._stream)      # backward's last kernel is the last thing enqueued on the compute stream
        if e0 is not None:
            self._marks.append((e0, done))
        cur.wait_event(done)
        return flat

    def all_reduce_(self, flat):
        if self.world == 1:
            return flat
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            cur = torch.cuda.current_stream(flat.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ready)
                # buckets from the END of the buffer first: backward finishes the head/decoder/last
                # blocks (high offsets) first
                for lo, hi in reversed(bucket_ranges(flat.numel(), self.n_buckets)):
                    dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
                done = torch.cuda.Event(enable_timing=self.measure)
                done.record(self._stream)
            if self.measure:
                self._marks.append((self._mark(cur, self._stream), done))
            cur.wait_event(done)
        else:
            for lo, hi in reversed(bucket_ranges(flat.numel(), self.n_buckets)):
                dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        return flat


class Trainer:
    """``step(spectrograms, speakers)`` == one iteration of reference src/learn.py:88-135."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, n_buckets=2, group=None,
                 use_graph=False, graph_warmup=2, max_steps_in_flight=0):
        self.model = model
        # A step is 400-600 kernel launches and the host enqueues one in ~1.2 ms: left alone it runs a few steps ahead of the
        # GPU until the HIP runtime's own back-pressure blocks it (a sleeping wait: harmless, the GPU stays fed).  Optional
        # bound on the steps in flight (0 = the runtime's): polls a word of pinned memory the stream stores to at the end of
        # every step (tn_mark_host), sleeping between polls — never spinning, see _lib.HostMarks.
        self.max_steps_in_flight = max(0, int(max_steps_in_flight))
        self._marks, self._mark_i = None, 0
        self.host_enqueue_s, self.host_enqueue_steps = 0.0, 0
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.reducer = FlatAllReducer(n_buckets, group)
        self.step_count = 0
        self.use_graph = bool(use_graph) and self.reducer.world == 1
        self.graph_warmup = graph_warmup
        self._graphs = {}          # (B, T) -> dict(graph, x, y, out, eager_steps)
        flat = model.flat_parameters()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        if self.reducer.world > 1:
            # replicas start identical (rank 0's weights), as DDP does
            dist.broadcast(flat, src=0, group=group)
            dist.broadcast(model._flat["bnbuf"], src=0, group=group)
            # backward finalises the gradient in 1 + n_buckets buckets, each all-reduced as soon as it is final.  Default 2
            # groups of mega blocks: every extra group costs the single-GPU step ~0.1 ms (split-K slabs, smaller launches:
            # 10.72 / 10.80 / 11.04 ms at 1 / 2 / 4 groups) while the whole 24.8 MB gradient is only ~0.3 ms of xGMI ring
            # all-reduce, so two groups already hide all but the last ~10 MB
            if model.grad_groups != n_buckets and n_buckets > 1:
                model.grad_groups = n_buckets
                model._drop_plans()

    def forward_backward(self, spectrograms, speakers, lengths=None):
        """forward + backward into the flat gradient buffer; returns (embeddings, preds, loss).  ``lengths``: valid frames
        per utterance of a zero-padded batch (collate_fn's second output): the padding mask of ``TitaNet.forward``."""
        m = self.model
        emb, preds, loss, plan = m._native_forward(spectrograms, speakers, lengths=lengths)
        dev = emb.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        vp = C.c_void_p
        check(m._lib.tn_backward(plan.handle, C.c_float(1.0), vp(0), vp(0), vp(0), vp(stream)), "tn_backward")
        self._last_plan = plan
        return emb, preds, loss

    def optimizer_step(self):
        m = self.model
        flat, grads = m.flat_parameters(), m.flat_gradients()
        self.step_count += 1
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        vp = C.c_void_p
        check(m._lib.tn_adam_step(vp(flat.data_ptr()), vp(grads.data_ptr()), vp(self.exp_avg.data_ptr()),
                                  vp(self.exp_avg_sq.data_ptr()), flat.numel(), self.lr, self.betas[0], self.betas[1],
                                  self.eps, self.weight_decay, self.step_count, 1.0 / self.reducer.world, vp(stream)),
              "tn_adam_step")

    def _throttle(self):
        if self.max_steps_in_flight and self._marks is not None:
            self._marks.wait(self._mark_i % self.max_steps_in_flight)       # the step `max_steps_in_flight` steps ago has finished

    def _mark_step(self, device):
        if not self.max_steps_in_flight:
            return
        if self._marks is None:
            from ._lib import HostMarks
            self._marks = HostMarks(self.max_steps_in_flight)
        self._marks.mark(self._mark_i % self.max_steps_in_flight, torch.cuda.current_stream(device).cuda_stream)
        self._mark_i += 1

    def step(self, spectrograms, speakers, lengths=None):
        self._throttle()
        t0 = time.perf_counter()
        out = self._step(spectrograms, speakers, lengths)
        # host time to ENQUEUE the step (launches are asynchronous): what must stay below the step time on every rank for the
        # GPU not to starve — 8 ranks share the host's cores (bench.py reports it per step)
        self.host_enqueue_s += time.perf_counter() - t0
        self.host_enqueue_steps += 1
        self._mark_step(out[0].device)
        return out

    def _step(self, spectrograms, speakers, lengths=None):
        # (the captured step copies a float32 tensor into its static input: a batch the mel front end packed into the plan
        #  — PackedSpectrograms — and ragged batches take the eager path)
        if self.use_graph and lengths is None and isinstance(spectrograms, torch.Tensor):
            return self._graph_step(spectrograms, speakers)
        out = self.forward_backward(spectrograms, speakers, lengths=lengths)
        grads = self.model.flat_gradients()
        if grads.is_cuda and len(self._last_plan.buckets) > 1:
            self.reducer.all_reduce_overlapped_(grads, self._last_plan, self.model._lib)
        else:
            self.reducer.all_reduce_(grads)
        self.optimizer_step()
        return out

    # ------------------------------------------------------------------ optimizer state (torch.optim.Adam layout)
    def _param_slices(self):
        from . import _lib as L
        return [(off, numel, shape) for _, kind, off, numel, shape in self.model._layout if kind == L.TN_KIND_PARAM]

    def optimizer_state_dict(self):
        """The fused optimizer's state in ``torch.optim.Adam.state_dict()`` layout (what the reference checkpoints as
        ``"optimizer"``, src/learn.py:188-195): per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` keyed by the index
        of the parameter in ``model.parameters()`` order, one param group.  ``torch.optim.Adam(model.parameters())
        .load_state_dict(...)`` accepts it, and :meth:`load_optimizer_state_dict` accepts an Adam state_dict."""
        sl = self._param_slices()
        state = {}
        for i, (off, numel, shape) in enumerate(sl):
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + numel].view(shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + numel].view(shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(sl)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        sl = self._param_slices()
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        step = 0
        for i, (off, numel, shape) in enumerate(sl):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.exp_avg[off:off + numel].view(shape).copy_(st["exp_avg"])
            self.exp_avg_sq[off:off + numel].view(shape).copy_(st["exp_avg_sq"])
            step = max(step, int(float(st["step"])))
        self.step_count = step
        for st in self._graphs.values():      # captured graphs hold no optimizer state, but restart their warm-up
            st["graph"], st["eager"] = None, 0

    # ------------------------------------------------------------------ one hipGraph per step
    def _device_step(self, x, y):
        """tick -> forward -> backward -> Adam, with the step count and the dropout word in device memory"""
        m = self.model
        vp = C.c_void_p
        plan = m._get_plan(x.shape[0], x.shape[2])
        stream = torch.cuda.current_stream(x.device).cuda_stream
        check(m._lib.tn_plan_step_tick(plan.handle, vp(stream)), "tn_plan_step_tick")
        emb, preds, loss, plan = m._native_forward(x, y, fixed_seed=True)
        check(m._lib.tn_backward(plan.handle, C.c_float(1.0), vp(0), vp(0), vp(0), vp(stream)), "tn_backward")
        flat, grads = m.flat_parameters(), m.flat_gradients()
        # lr < 0: read the plan's device lr word (written by _sync_device_state before every step, so a scheduler that
        # assigns trainer.lr keeps working under graph replay); betas / eps / weight decay are captured by value — changing
        # them re-captures (see _graph_step)
        check(m._lib.tn_adam_step_plan(plan.handle, vp(flat.data_ptr()), vp(grads.data_ptr()), vp(self.exp_avg.data_ptr()),
                                       vp(self.exp_avg_sq.data_ptr()), flat.numel(), -1.0, self.betas[0], self.betas[1],
                                       self.eps, self.weight_decay, 1.0, vp(stream)), "tn_adam_step_plan")
        return emb, preds, loss, plan

    def _sync_device_state(self, plan, device):
        """ONE step counter per Trainer: every plan's device counter / dropout word / lr word is set from the trainer's
        state right before the step that uses it (two one-thread kernels outside the graph), so alternating input shapes
        (RandomChunk's 1.5 / 2 / 3 s lengths give three plans) cannot drift apart or replay a stale learning rate."""
        m = self.model
        stream = torch.cuda.current_stream(device).cuda_stream
        check(m._lib.tn_plan_step_set(plan.handle, self.step_count - 1, C.c_void_p(stream)), "tn_plan_step_set")
        check(m._lib.tn_plan_set_lr(plan.handle, C.c_float(float(self.lr)), C.c_void_p(stream)), "tn_plan_set_lr")

    def _graph_step(self, spectrograms, speakers):
        key = (tuple(spectrograms.shape), spectrograms.device.index)
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"graph": None, "eager": 0, "hp": None,
                                      "x": torch.empty_like(spectrograms), "y": torch.empty_like(speakers)}
        st["x"].copy_(spectrograms)
        st["y"].copy_(speakers)
        self.step_count += 1
        hp = (tuple(self.betas), self.eps, self.weight_decay)
        if st["graph"] is not None and (self.model._plans.get(st["plan"].key) is not st["plan"] or st["hp"] != hp):
            st["graph"], st["eager"] = None, 0          # plan evicted from the model's cache / captured hyper-parameters changed
        plan = self.model._get_plan(st["x"].shape[0], st["x"].shape[2])
        self._sync_device_state(plan, st["x"].device)
        if st["graph"] is None:
            if st["eager"] < self.graph_warmup:
                st["eager"] += 1
                return self._device_step(st["x"], st["y"])[:3]
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                out = self._device_step(st["x"], st["y"])
            st["out"], st["plan"], st["graph"], st["hp"] = out[:3], out[3], g, hp
            # the capture itself does not execute: run the captured step for this call
        st["graph"].replay()
        return st["out"]
