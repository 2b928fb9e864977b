"""titanet_amd — MI355X-native (gfx950 / CDNA4) TitaNet hot path behind the reference's module surface.

    from titanet_amd import TitaNet, LOSSES
    loss = LOSSES["ce"](192, 251, device="cuda")
    model = TitaNet.get_titanet(n_mega_blocks=17, model_size="s", loss_function=loss, dropout=0.1, device="cuda")
    embeddings, preds, loss_value = model(spectrograms, speakers=speakers)   # reference src/learn.py:95-97
    loss_value.backward()
"""
from .losses import LOSSES, ArcFaceLoss, CELoss, CosFaceLoss, MetricLearningLoss, SphereFaceLoss  # noqa: F401
from .models import TitaNet  # noqa: F401

__all__ = ["TitaNet", "LOSSES", "CELoss", "ArcFaceLoss", "CosFaceLoss", "SphereFaceLoss", "MetricLearningLoss"]
