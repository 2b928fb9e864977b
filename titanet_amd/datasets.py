"""Batch assembly contract of the reference data path (reference src/datasets.py:48-73).

Only ``collate_fn`` is on the boundary of the hot path: it defines what ``TitaNet.forward`` receives —
``spectrograms`` float32 ``[B, n_mels, max T]`` zero-padded on the right, ``lengths`` int64 ``[B]``, ``speakers``
int64 ``[B]``.  The dataset classes / downloaders themselves are out of scope.

MI355X notes: examples may already live on the GPU (the mel front end here runs on the device), in which case the
batch is assembled on the device; host examples are assembled in pageable memory (pin with ``DataLoader(pin_memory=True)``,
which pins in the parent process, so the H2D copy of the caller's ``.to(device)``, reference src/learn.py:95, can be asynchronous).  Padded frames are ordinary zeros for the model — the
reference applies no length mask, and neither does this path (the lengths are returned for the caller, as there).
"""
import torch


def collate_fn(batch, n_mels=80):
    """list of ``{"spectrogram": [1 or n_mels-leading, n_mels, T_i], "speaker_id": int}`` -> (spectrograms, lengths, speakers)."""
    lengths = torch.tensor([int(e["spectrogram"].size(-1)) for e in batch], dtype=torch.int64)
    speakers = torch.tensor([int(e["speaker_id"]) for e in batch], dtype=torch.int64)
    first = batch[0]["spectrogram"]
    max_t = int(lengths.max()) if len(batch) else 0
    if first.is_cuda:
        out = torch.zeros(len(batch), n_mels, max_t, dtype=torch.float32, device=first.device)
    else:
        # no pinning here: as a DataLoader collate_fn this runs in forked worker processes, where a pinned allocation would
        # initialise the GPU runtime ("Cannot re-initialize CUDA in forked subprocess"); DataLoader(pin_memory=True) pins in
        # the parent process instead
        out = torch.zeros(len(batch), n_mels, max_t, dtype=torch.float32)
    for i, e in enumerate(batch):
        out[i, :, :lengths[i]] = e["spectrogram"].to(torch.float32).reshape(n_mels, -1)
    return out, lengths, speakers
