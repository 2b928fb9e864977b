"""TEST / BENCH INFRASTRUCTURE ONLY — the reference's hot path as an eager ``nn.Module`` graph on the CPU.

This is what ``bench.py``'s ``cpu_baseline`` leg times: the SAME module graph the reference
dispatches per step — ``nn.Conv1d`` / ``nn.BatchNorm1d`` / ``nn.Linear`` / ``nn.Dropout`` leaf modules, ``F.pad`` before
every conv, the margin loss's per-row Python loop — executed by PyTorch's CPU kernels, as opposed to
``oracle/titanet_oracle.py`` (the explicit-arithmetic restatement the parity tests use).  The reference's files cannot
travel to the GPU box, so the graph is rebuilt here from the reference's structure (cited per class); it is pinned to the
restatement (and through it to the reference's golden vectors) by ``tests/test_oracle_golden.py``.  Nothing under
``titanet_amd/`` imports this file.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class SamePadConv1d(nn.Conv1d):
    """reference src/modules.py:5-40: explicit zero padding on both sides, then the convolution without padding."""

    def forward(self, x):
        k, s, d = self.kernel_size[0], self.stride[0], self.dilation[0]
        w = x.shape[-1]
        pad = (s * (w - 1) - w + k + (d - 1) * (k - 1)) // 2
        return self._conv_forward(F.pad(x, (pad, pad)), self.weight, self.bias)


class DepthwiseSeparable(nn.Module):
    """reference src/modules.py:43-93: depthwise(k, groups=C, bias) then pointwise(1, bias); keys conv.0 / conv.1."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Sequential(SamePadConv1d(cin, cin, k, groups=cin), SamePadConv1d(cin, cout, 1))

    def forward(self, x):
        return self.conv(x)


class ConvBlock(nn.Module):
    """reference src/modules.py:96-148: conv -> BatchNorm1d -> ReLU [-> Dropout]; keys conv_block.0 / .1."""

    def __init__(self, cin, cout, k, activation=True, dropout=0.0, depthwise=False):
        super().__init__()
        mods = [DepthwiseSeparable(cin, cout, k) if depthwise else SamePadConv1d(cin, cout, k), nn.BatchNorm1d(cout)]
        if activation:
            mods.append(nn.ReLU())
        if dropout > 0:
            mods.append(nn.Dropout(p=dropout))
        self.conv_block = nn.Sequential(*mods)

    def forward(self, x):
        return self.conv_block(x)


class SqueezeExcite(nn.Module):
    """reference src/modules.py:151-189: mean over time -> Linear(C, C/r, no bias) -> ReLU -> Linear(C/r, C, no bias) -> sigmoid."""

    def __init__(self, c, reduction=16):
        super().__init__()
        self.excitation = nn.Sequential(nn.Linear(c, c // reduction, bias=False), nn.ReLU(), nn.Linear(c // reduction, c, bias=False),
                                        nn.Sigmoid())

    def forward(self, x):
        g = self.excitation(F.adaptive_avg_pool1d(x, 1).squeeze(-1))
        return x * g.unsqueeze(-1)


class MegaBlock(nn.Module):
    """reference src/models.py:407-472."""

    def __init__(self, c, k, n_sub, dropout, reduction=16):
        super().__init__()
        self.dropout = dropout
        subs = [ConvBlock(c, c, k, activation=True, dropout=dropout, depthwise=True) for _ in range(n_sub)]
        subs.append(SqueezeExcite(c, reduction))
        self.sub_blocks = nn.Sequential(*subs)
        self.skip_connection = nn.Sequential(nn.Conv1d(c, c, 1), nn.BatchNorm1d(c))

    def forward(self, x):
        y = self.skip_connection(x) + self.sub_blocks(x)
        return F.dropout(F.relu(y), p=self.dropout, training=self.training)


class Encoder(nn.Module):
    """reference src/models.py:342-404."""

    def __init__(self, n_mels, n_mega, n_sub, hidden, out, k, dropout):
        super().__init__()
        self.prolog = ConvBlock(n_mels, hidden, 3)
        self.mega_blocks = nn.Sequential(*[MegaBlock(hidden, k, n_sub, dropout) for _ in range(n_mega)])
        self.epilog = ConvBlock(hidden, out, 1)

    def forward(self, x):
        return self.epilog(self.mega_blocks(self.prolog(x)))


class AttentiveStatsPool(nn.Module):
    """reference src/models.py:532-584."""

    def __init__(self, d, a, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.in_linear = nn.Linear(d, a)
        self.out_linear = nn.Linear(a, d)

    def forward(self, enc):
        e = self.out_linear(torch.tanh(self.in_linear(enc.transpose(1, 2)))).transpose(1, 2)
        alphas = torch.softmax(e, dim=2)
        means = torch.sum(alphas * enc, dim=2)
        resid = torch.sum(alphas * enc ** 2, dim=2) - means ** 2
        return torch.cat([means, torch.sqrt(resid.clamp(min=self.eps))], dim=1)


class Decoder(nn.Module):
    """reference src/models.py:475-529 (attentive pooling variant)."""

    def __init__(self, d, a, emb):
        super().__init__()
        self.pool = nn.Sequential(AttentiveStatsPool(d, a), nn.BatchNorm1d(2 * d))
        self.linear = nn.Sequential(nn.Linear(2 * d, emb), nn.BatchNorm1d(emb))

    def forward(self, enc):
        return self.linear(self.pool(enc))


class CEHead(nn.Module):
    """reference src/losses.py:22-44."""

    def __init__(self, emb, n_classes):
        super().__init__()
        self.fc = nn.Linear(emb, n_classes)

    def forward(self, x, y):
        logits = self.fc(x)
        return F.normalize(x, p=2, dim=1), logits.argmax(dim=1), F.cross_entropy(logits, y)


class MarginHead(nn.Module):
    """reference src/losses.py:47-132, including its per-row exclusion loop (:119-126) — the baseline times what the
    reference executes, not a vectorised rewrite."""

    def __init__(self, emb, n_classes, scale=30.0, m1=1.0, m2=0.2, m3=0.0, eps=1e-6):
        super().__init__()
        self.fc = nn.Linear(emb, n_classes, bias=False)
        self.scale, self.m1, self.m2, self.m3, self.eps = scale, m1, m2, m3, eps

    def forward(self, x, y):
        self.fc.weight.data = F.normalize(self.fc.weight.data, p=2, dim=1)
        norms = torch.norm(x, p=2, dim=-1, keepdim=True)
        xn = x / norms
        scales = norms.squeeze(-1) if self.scale is None else torch.full((len(x),), float(self.scale), dtype=x.dtype)
        cos = self.fc(xn).clamp(-1, 1)
        preds = cos.argmax(dim=1)
        theta = torch.arccos(cos)
        rows = torch.arange(len(y))
        num = scales * (torch.cos(self.m1 * theta[rows, y] + self.m2) - self.m3)
        others = []
        for i in range(len(y)):                      # one small tensor per utterance, as the reference builds them
            row = cos[i]
            others.append(torch.cat([row[:y[i]], row[y[i] + 1:]]).unsqueeze(0))
        others = torch.cat(others, dim=0)
        den = torch.exp(num) + torch.sum(torch.exp(scales.unsqueeze(-1) * others), dim=1)
        return xn, preds, -torch.mean(num - torch.log(den + self.eps))


class EagerTitaNet(nn.Module):
    """reference src/models.py:162-339 (TitaNet) with the reference's state_dict key names."""

    def __init__(self, n_mels=80, n_mega_blocks=17, n_sub_blocks=3, hidden=256, enc_out=1536, emb=192, kernel=3, attn_hidden=128,
                 dropout=0.1, loss=None, n_classes=251):
        super().__init__()
        self.encoder = Encoder(n_mels, n_mega_blocks, n_sub_blocks, hidden, enc_out, kernel, dropout)
        self.decoder = Decoder(enc_out, attn_hidden, emb)
        self.loss_function = None
        if loss == "ce":
            self.loss_function = CEHead(emb, n_classes)
        elif loss == "arc":
            self.loss_function = MarginHead(emb, n_classes, scale=30.0, m1=1.0, m2=0.2, m3=0.0)      # parameters.yml:42-44

    def forward(self, spectrograms, speakers=None):
        emb = self.decoder(self.encoder(spectrograms))
        if speakers is None:
            return F.normalize(emb, p=2, dim=1)
        return self.loss_function(emb, speakers)
