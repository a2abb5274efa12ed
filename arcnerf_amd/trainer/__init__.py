"""The step either side of the path: what the reference's trainer does around `model(feed_in)` every iteration -
`step_optimize` (arcnerf/trainer/arcnerf_trainer.py:319-333), the dynamic batch size of its ray pipeline
(arcnerf/trainer/pipeline.py:9-317: centre precrop, shuffle, batch fetch with the random background blend, dynamic batch size) and the EMA it applies after the optimiser (arcnerf/trainer/ema.py).  Datasets, logging,
checkpoint rotation and evaluation are the caller's (out of scope)."""
from .dynamic_bs import DynamicBsMeter
from .ema import EMA
from .fused_neus_step import FusedNeusNgpStep
from .fused_step import FusedNgpStep
from .loss import AllLoss, EikonalLoss, HuberLoss, ImgLoss, build_loss
from .pipeline import Pipeline, TrainBatches, get_model_feed_in
from .step import step_optimize, train_epoch

__all__ = ['AllLoss', 'DynamicBsMeter', 'EMA', 'EikonalLoss', 'FusedNeusNgpStep', 'FusedNgpStep', 'HuberLoss', 'ImgLoss', 'build_loss', 'Pipeline', 'TrainBatches', 'get_model_feed_in', 'step_optimize', 'train_epoch']
