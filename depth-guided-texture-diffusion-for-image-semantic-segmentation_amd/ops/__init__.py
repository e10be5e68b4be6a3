"""torch.autograd.Function wrappers over the C ABI — the same pattern as the reference's own native op
(twig/ops/functions/ms_deform_attn_func.py:19-46): forward saves its inputs, backward is
once_differentiable and returns one gradient per input."""
from . import _native  # noqa: F401
from ._native import NO_RUN, block_run, mark_flush_point  # noqa: F401
from .layernorm import layer_norm, layer_norm_fork  # noqa: F401
from .attention import sra_attention  # noqa: F401
from .diffuser import diffuser_state, diffuse_tail  # noqa: F401
from .dwconv import dwconv_fork, dwconv_nhwc  # noqa: F401
from .elementwise import colsum, linear, linear_gelu, linear_residual, mlp_residual, scale_residual  # noqa: F401
from . import conv3x3 as conv3x3_ops  # noqa: F401
from .conv3x3 import conv3x3, conv3x3_stack  # noqa: F401
from .hitnet import (batch_norm, batch_norm_supported, bilinear_resize, ca_gate, cab, cat_channels, prelu, sam, sam_supported,  # noqa: F401
                     stack, unstack)
from .loss import seg_loss, ssim_value  # noqa: F401
from .ms_deform_attn import MSDeformAttnFunction, ms_deform_attn  # noqa: F401
from .conv_gemm import conv2d as conv2d_gemm, conv2d_tokens  # noqa: F401
