"""Speaker-verification metrics with the reference's function surface (reference src/utils.py:294-403)
and a batched verification harness (reference src/learn.py:409-459, src/datasets.py:165-183).

The reference computes EER with sklearn.roc_curve + scipy interp1d/brentq and minDCF with the
voxceleb_trainer recipe; both are restated here in numpy (exact for the piecewise-linear ROC) and pinned
by golden values produced by the reference's own functions (tests/golden/metrics.npz).
"""
import itertools

import numpy as np
import torch


def _roc(scores, labels):
    """(fpr, tpr) at every distinct threshold, starting at (0, 0) — sklearn.metrics.roc_curve semantics."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels).astype(bool)
    order = np.argsort(-scores, kind="mergesort")
    s, y = scores[order], labels[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, len(s) - 1]
    tps = np.cumsum(y)[idx].astype(np.float64)
    fps = (1 + idx - np.cumsum(y)[idx]).astype(np.float64)
    tps, fps = np.r_[0.0, tps], np.r_[0.0, fps]
    return fps / max(fps[-1], 1e-300), tps / max(tps[-1], 1e-300)


def compute_eer(scores, labels):
    """reference src/utils.py:294-300: root of 1 - x - interp1d(fpr, tpr)(x) on [0, 1]."""
    fpr, tpr = _roc(scores, labels)
    f = 1.0 - fpr - tpr          # decreasing along the curve; linear on every ROC segment
    for i in range(len(fpr) - 1):
        if f[i] == 0.0:
            return float(fpr[i])
        if f[i] > 0.0 >= f[i + 1]:
            if fpr[i + 1] == fpr[i]:       # vertical segment: f(x) jumps through zero at x = fpr[i]
                return float(fpr[i])
            # x in [fpr_i, fpr_{i+1}]: tpr(x) = tpr_i + slope (x - fpr_i); solve 1 - x - tpr(x) = 0
            slope = (tpr[i + 1] - tpr[i]) / (fpr[i + 1] - fpr[i])
            return float((1.0 - tpr[i] + slope * fpr[i]) / (1.0 + slope))
    return float(fpr[-1])


def compute_error_rates(scores, labels, eps=1e-6):
    """reference src/utils.py:303-344 (voxceleb_trainer): fnrs, fprs at thresholds = sorted scores."""
    order = np.argsort(np.asarray(scores, dtype=np.float64), kind="mergesort")
    y = np.asarray(labels)[order].astype(np.float64)
    fnrs = np.cumsum(y)
    fprs = np.cumsum(1.0 - y)
    fnrs_norm = y.sum()
    fprs_norm = len(y) - fnrs_norm
    fnrs = fnrs / (fnrs_norm + eps)
    fprs = 1.0 - fprs / (fprs_norm + eps)
    return list(fnrs), list(fprs)


def compute_mindcf(scores, labels, p_target=1e-2, c_fa=1, c_miss=1, eps=1e-6):
    """reference src/utils.py:347-367"""
    fnrs, fprs = compute_error_rates(scores, labels)
    c_det = c_miss * np.asarray(fnrs) * p_target + c_fa * np.asarray(fprs) * (1 - p_target)
    c_def = min(c_miss * p_target, c_fa * (1 - p_target))
    return float(c_det.min() / (c_def + eps))


def get_test_metrics(scores, labels, mindcf_p_target=1e-2, mindcf_c_fa=1, mindcf_c_miss=1, prefix=None):
    """reference src/utils.py:385-403"""
    metrics = {"eer": compute_eer(scores, labels),
               "mindcf": compute_mindcf(scores, labels, p_target=mindcf_p_target, c_fa=mindcf_c_fa, c_miss=mindcf_c_miss)}
    if prefix is not None:
        metrics = {f"{prefix}/{k}": v for k, v in metrics.items()}
    return metrics


@torch.no_grad()
def verification_test(model, spectrograms, speakers, mindcf_p_target=1e-2, mindcf_c_fa=1, mindcf_c_miss=1, batch_size=64):
    """``learn.test`` (reference src/learn.py:409-459) without its 200x redundant forward passes:
    every utterance is embedded ONCE, ``batch_size`` utterances per forward: the batch is zero-padded to its longest
    utterance and run with the padding mask (``lengths=``), which makes every row equal to that utterance embedded on
    its own (eval mode: running statistics; tests/test_mask_gpu.py) — what the reference computes with B = 1 per pair.
    Then all ordered pairs including self-pairs (itertools.product(indices, repeat=2), src/datasets.py:171-183) are scored
    with the cosine similarity of the L2-normalised embeddings.
    spectrograms: list of [n_mels, T_i] or [1, n_mels, T_i] tensors; speakers: list of ids."""
    was_training = model.training
    model.eval()
    dev = model.flat_parameters().device
    specs = [(s[0] if s.dim() == 3 else s) for s in spectrograms]
    n_utt = len(specs)
    emb_of = [None] * n_utt
    if getattr(getattr(model, "_cfg", None), "simple_pool", False):
        # the mean-pool decoder has no padding-mask form (tn_forward_masked refuses it): one utterance per forward, as the
        # reference does
        for i, s in enumerate(specs):
            emb_of[i] = model(s.to(device=dev, dtype=torch.float32).unsqueeze(0))
    else:
        # utterances sorted by length, `batch_size` per forward, the frame axis padded to a multiple of 128: few distinct
        # (batch, frames) shapes, so the evaluation does not churn through the module's plan cache (MAX_PLANS) and evict the
        # training plans; the padding mask makes the extra frames invisible
        order = sorted(range(n_utt), key=lambda i: int(specs[i].shape[-1]))
        for lo in range(0, n_utt, batch_size):
            idx = order[lo:lo + batch_size]
            chunk = [specs[i] for i in idx]
            lens = torch.tensor([int(s.shape[-1]) for s in chunk], dtype=torch.int64)
            T_pad = (int(lens.max()) + 127) // 128 * 128
            x = torch.zeros(len(chunk), chunk[0].shape[0], T_pad, dtype=torch.float32, device=dev)
            for i, s in enumerate(chunk):
                x[i, :, :s.shape[-1]] = s.to(device=dev, dtype=torch.float32)
            e = model(x, lengths=lens) if int(lens.min()) < T_pad else model(x)
            for k, i in enumerate(idx):
                emb_of[i] = e[k:k + 1]
    embs = emb_of
    model.train(was_training)
    e = torch.cat(embs, dim=0)
    e = e / e.norm(dim=1, keepdim=True).clamp(min=1e-8)       # F.cosine_similarity eps
    sim = (e @ e.t()).cpu().numpy()
    n = len(spectrograms)
    pairs = list(itertools.product(range(n), repeat=2))
    scores = np.array([sim[i, j] for i, j in pairs])
    labels = np.array([int(speakers[i] == speakers[j]) for i, j in pairs])
    return get_test_metrics(scores, labels, mindcf_p_target, mindcf_c_fa, mindcf_c_miss, prefix="test"), scores, labels
