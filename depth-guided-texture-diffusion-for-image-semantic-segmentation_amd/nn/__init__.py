"""Drop-in nn.Module replacements for the reference's hot path (twig/model/cod.py): same class names,
constructor/forward signatures and state_dict keys, backed by the gfx950 kernels in libdgtd.so."""
from .modules import (Attention, Linear, Conv2d, wb, BasicConv2d, Block, CAB, CALayer, DropPath, DWConv, Hitnet, LayerNorm,  # noqa: F401
                      MessagePassing, Mlp, OverlapPatchEmbed, PyramidVisionTransformerImpr, SAM, ShapePropDecoder,
                      ShapePropEncoder, ShapePropWeightRegressor, cod, convnext_Block, prompt_decoder, prompt_encoder,
                      pvt_v2_b1, pvt_v2_b2, pvt_v2_b3, pvt_v2_b4, pvt_v2_b5, PVT_VARIANTS, cal_loss, ssim_value, fft_highpass)
