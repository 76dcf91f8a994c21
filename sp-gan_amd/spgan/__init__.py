"""spgan: MI355X-native SP-GAN train-step hot path (HIP kernels behind the reference's nn.Module surface).

    from spgan import Generator, Discriminator, EdgeBlock, AdaptivePointNorm, get_edge_features
    from spgan import dis_loss, gen_loss, GradientPenalty, TrainStep

Importing the package does not touch the GPU; the shared library is loaded on first use and its
absence is a hard error (there is no CPU fallback).
"""
from . import dataset, fixture_rng, metrics, ops, pointnet_util, sampling  # noqa: F401
from .capture import CapturedBody                                          # noqa: F401
from .losses import GradientPenalty, dis_loss, gen_loss                    # noqa: F401
from .modules import AdaptivePointNorm, Discriminator, EdgeBlock, Generator, get_edge_features   # noqa: F401
from .optim import Adam, flatten_module                                    # noqa: F401
from .parallel import DataParallel, init_process_group_from_env, shard_batch   # noqa: F401
from .train import TrainStep, requires_grad                                # noqa: F401

__all__ = ["Generator", "Discriminator", "EdgeBlock", "AdaptivePointNorm", "get_edge_features", "dis_loss", "gen_loss",
           "GradientPenalty", "TrainStep", "CapturedBody", "Adam", "DataParallel", "requires_grad", "ops", "fixture_rng", "sampling"]
