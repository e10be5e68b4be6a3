"""Minimal pieces of the training loop the reference delegates to nest/mmengine."""
from .optim import FlatAdamW, LossScaler, build_optimizer, lr_mult_for  # noqa: F401
from .data import DefaultSampler, SyntheticRGBD, batches, device_preprocess, device_sample
from . import metrics  # noqa: F401  # noqa: F401
from .checkpoint import load_checkpoint, load_checkpoint_file, load_pretrained, save_checkpoint  # noqa: F401
from .config import CosineByEpoch, Runner, build_model, build_optim, load_config  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401
from . import registry  # noqa: F401
from .registry import build as build_exported, export, install_nest_shim, register_model  # noqa: F401
