"""Golden-vector case table shared by ``make_golden.py`` (reference side) and the tests.

All channel counts are multiples of 8 (the HIP kernels move 8-channel vectors).
"""

import numpy as np

from oracle import detgen

_TINY = dict(n_mels=16, n_mega_blocks=2, hidden=32, enc_out=96, emb=16, kernel=3, attn_hidden=16)

CASES = {
    # tiny S-like (K=3) config with per-layer intermediates, all grads, CE + ArcFace + Cos + Sphere
    "tiny_k3": dict(cfg=dict(_TINY), batch=4, frames=37, n_classes=10, seed=1, inter=True,
                    losses=("ce", "arc", "cos", "sphere"), grads="all",
                    buffers=("encoder.prolog.conv_block.1.running_mean",
                             "encoder.prolog.conv_block.1.running_var",
                             "encoder.prolog.conv_block.1.num_batches_tracked",
                             "encoder.mega_blocks.1.sub_blocks.2.conv_block.1.running_var",
                             "encoder.mega_blocks.0.skip_connection.1.running_mean",
                             "decoder.pool.1.running_var", "decoder.linear.1.running_mean")),
    # M-like kernel (K=7), odd T
    "tiny_k7": dict(cfg=dict(_TINY, kernel=7, n_mega_blocks=1), batch=3, frames=21, n_classes=7, seed=2,
                    inter=True, losses=("ce",), grads="all", buffers=()),
    # L-like kernel (K=11) with T < K (edge case: SURVEY.md §7.1d)
    "tiny_k11_short": dict(cfg=dict(_TINY, kernel=11, n_mega_blocks=1), batch=2, frames=5, n_classes=5, seed=3,
                           inter=True, losses=("ce",), grads="all", buffers=()),
    # tile-straddling shape: B*T not a multiple of any tile, T = 301 (3 s chunk), wider hidden
    "mid_k3": dict(cfg=dict(n_mels=80, n_mega_blocks=1, hidden=64, enc_out=128, emb=32, kernel=3, attn_hidden=32),
                   batch=3, frames=301, n_classes=11, seed=4, inter=False, losses=("ce", "arc"),
                   grads=("encoder.prolog.conv_block.0.bias", "encoder.mega_blocks.0.sub_blocks.1.conv_block.0.conv.1.bias",
                          "loss_function.fc.weight", "decoder.linear.0.bias",
                          "encoder.mega_blocks.0.sub_blocks.3.excitation.0.weight"), buffers=()),
    # BASELINE.json configs[0]: TitaNet-S (17 mega blocks, parameters.yml:54), B=8, 80 x 300
    "s17_b8": dict(cfg=dict(n_mels=80, n_mega_blocks=17, hidden=256, enc_out=1536, emb=192, kernel=3, attn_hidden=128),
                   batch=8, frames=300, n_classes=251, seed=42, inter=False, losses=("ce", "arc"),
                   grads=("encoder.prolog.conv_block.0.bias", "loss_function.fc.weight",
                          "encoder.mega_blocks.8.sub_blocks.1.conv_block.0.conv.1.bias",
                          "encoder.mega_blocks.16.sub_blocks.3.excitation.2.weight",
                          "decoder.linear.0.bias", "encoder.epilog.conv_block.1.weight"), buffers=()),
    # Decoder(simple_pool=True): mean over time -> Linear(D, 2D) (reference src/models.py:497-502)
    "tiny_simple_pool": dict(cfg=dict(_TINY, n_mega_blocks=1), simple_pool=True, batch=4, frames=29, n_classes=9, seed=5,
                             inter=False, losses=("ce", "arc"), grads="all", buffers=()),
}


def head_clamp_inputs():
    """Deterministic inputs of the ``head_clamp`` fixture (rebuilt by the tests): embeddings whose cosine with a class
    row is exactly +1 / -1 before the clamp (SURVEY.md 8c "include a row with cos -> +-1 clamp"): row 0 is a positive
    multiple of class row 2, row 1 a negative multiple of class row 5 (neither is that row's target: the reference's
    arccos'(+-1) is infinite at the TARGET column, src/losses.py:100-110), rows 2.. are generic."""
    B, E, NC = 6, 16, 9
    w = detgen.tensor_for("head_clamp.fc.weight", (NC, E), seed=7).astype(np.float32)
    x = (detgen.tensor_for("head_clamp.inputs", (B, E), seed=7) * 4.0).astype(np.float32)
    x[0] = 3.0 * w[2]
    x[1] = -2.0 * w[5]
    y = np.array([4, 1, 0, 8, 3, 3], dtype=np.int64)
    return x, w, y
