"""MI355X-native implementation of the streamflow model's training / inference hot path.

The public names mirror the reference's modules (src/model/*.py, src/loss.py); everything
computes in hand-written gfx950 kernels reached through libsfhip.so (include/sfhip.h).
There is no CPU or eager-PyTorch fallback: without the built library or without a GPU the
modules raise.
"""
from . import ops
from .baselines import AttentionGRUModel, LightweightCNNReducer, PureBPModel, PureLSTMModel, build_baseline_model
from .dataset import ResidentStreamflowDataset, window_shards
from .evaluation import Evaluator, calculate_metrics, calculate_psnr, calculate_ssim, merge_partials
from .functional import compute_dtype, set_compute_dtype
from .modules import (CNNEncoder, CombinedLoss, GANLoss, ImageLoss, LSTMModule, PatchDiscriminator,
                      PositionalEncoding, SimpleUpsamplingDecoder, StationLoss, StreamflowPredictionModel,
                      TransformerModule)

__all__ = ["ops", "AttentionGRUModel", "LightweightCNNReducer", "PureBPModel", "PureLSTMModel", "build_baseline_model", "ResidentStreamflowDataset", "window_shards", "compute_dtype", "set_compute_dtype", "CNNEncoder", "CombinedLoss", "GANLoss", "ImageLoss",
           "LSTMModule", "PatchDiscriminator", "PositionalEncoding", "SimpleUpsamplingDecoder", "StationLoss",
           "StreamflowPredictionModel", "TransformerModule", "Evaluator", "calculate_metrics", "calculate_psnr",
           "calculate_ssim", "merge_partials"]
