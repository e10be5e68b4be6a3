"""HIP-backed modules.  Layout policy: everything token-major / NHWC ([B, H*W, C]) inside the two
trunks, so LayerNorm, attention and the depthwise convs see contiguous channel rows and no
NCHW<->NHWC permute copies exist on the path; strided k==s convolutions (ConvNeXt stem/downsample,
the attention spatial-reduction conv) are patchify + GEMM.

Reference citations are to /root/reference/twig/model/cod.py.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

LATENT = 24
# A/B switches for tools/ and bench runs (default: every native path on)
_USE = {k: os.environ.get("DGTD_" + k.upper(), "1") != "0" for k in ("bilinear", "conv3x3", "cab_glue", "fused_linear", "ln_fork", "conv_gemm", "mlp_fused", "async_flush", "cab_node", "batchnorm", "sam_node")}


# ------------------------------------------------------------------------------------------------ helpers
class DropPath(nn.Module):
    """Stochastic depth, timm 0.6.13 semantics (cod.py:816): identity in eval or when p == 0."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

        self.index, self.plan = -1, None   # set by cod: all masks of a step are drawn in one launch (see cod._run)

    def scale(self, batch: int, device):
        """Per-sample scale [B] fp32 (0 or 1/keep) for this step, or None when the layer is the identity."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        plan = self.plan
        if plan is not None and plan.get("masks") is not None and plan["masks"].shape[1] == batch:
            return plan["masks"][self.index]
        keep = 1.0 - self.drop_prob
        return torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def forward(self, x):
        s = self.scale(x.shape[0], x.device)
        if s is None:
            return x
        return x * s.to(x.dtype).view((x.shape[0],) + (1,) * (x.ndim - 1))


class Linear(nn.Linear):
    """nn.Linear whose compute can run on a low-precision WORKING COPY of the fp32 master parameters.
    ``_w``/``_b`` are plain attributes installed by dist.GradReducer (not parameters, not in state_dict)."""
    _w = None
    _b = None

    def forward(self, x):
        return ops.linear(x, *wb(self)) if x.is_cuda else F.linear(x, *wb(self))


class Conv2d(nn.Conv2d):
    """nn.Conv2d with the same optional working copy."""
    _w = None
    _b = None

    def forward(self, x):
        w, b = wb(self)
        if (x.is_cuda and _USE["conv_gemm"] and self.groups == 1 and self.dilation == (1, 1) and self.kernel_size[0] == self.kernel_size[1]
                and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] and isinstance(self.padding[0], int)
                and self.padding_mode == "zeros"):
            return ops.conv2d_gemm(x, w, b, self.stride[0], self.padding[0])      # patch gather + library GEMM, no MIOpen
        return self._conv_forward(x, w, b)


def wb(m):
    """(weight, bias) to compute with: the working copy when one is installed, else the master parameters."""
    w = m._w
    if w is None:
        return m.weight, m.bias
    return w, m._b


def _init_weights(m: nn.Module) -> None:
    """cod.py:1401-1414."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, (nn.LayerNorm, LayerNorm)):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


def _patchify(x: torch.Tensor, H: int, W: int, k: int) -> torch.Tensor:
    """tokens [B, H*W, C] -> [B, (H/k)*(W/k), C*k*k] in (c, ky, kx) order = Conv2d weight.flatten(1) order."""
    B, _, Cc = x.shape
    x = x.view(B, H // k, k, W // k, k, Cc).permute(0, 1, 3, 5, 2, 4)
    return x.reshape(B, (H // k) * (W // k), Cc * k * k)


def _tokens_to_nchw(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """[B, H*W, C] tokens -> logical [B, C, H, W] WITHOUT moving data: the result is a channels_last tensor, which is what
    MIOpen's NHWC convolution kernels, BatchNorm and the bilinear resamplers consume natively."""
    B, _, Cc = x.shape
    return x.view(B, H, W, Cc).permute(0, 3, 1, 2)


def _nchw_to_tokens(x: torch.Tensor) -> torch.Tensor:
    """logical [B, C, H, W] -> [B, H*W, C]; a free view when x is channels_last, one transpose copy otherwise."""
    B, Cc, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, H * W, Cc)


class LayerNorm(nn.Module):
    """cod.py:1025-1049 and nn.LayerNorm.  Both data formats normalise the channel dim; in the token-major
    layout of this package that is always the last dim, so one kernel serves both."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps, self.data_format = eps, data_format
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_first" and x.ndim == 4:  # stand-alone NCHW use keeps the reference contract
            y = ops.layer_norm(x.permute(0, 2, 3, 1).contiguous(), self.weight, self.bias, self.eps)
            return y.permute(0, 3, 1, 2)
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def fork(self, x):
        """(norm(x), x): the second value is for the skip connection of a pre-norm block (gradient add fused into the backward)."""
        return ops.layer_norm_fork(x, self.weight, self.bias, self.eps)


# ------------------------------------------------------------------------------------------------ PVTv2
class OverlapPatchEmbed(nn.Module):
    """cod.py:964-1004."""

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size, self.stride = patch_size, stride
        self.proj = Conv2d(in_chans, embed_dim, patch_size, stride, patch_size // 2)
        self.norm = LayerNorm(embed_dim, eps=1e-5)
        self.apply(_init_weights)

    def forward(self, x):
        if x.is_cuda and _USE["conv_gemm"]:
            # patch gather straight from the producer's layout (NCHW fp32 image or channels_last map; the 16-bit cast of the image
            # happens inside the gather) + one GEMM: the result already is the token matrix
            t, H, W = ops.conv2d_tokens(x, *wb(self.proj), self.stride, self.patch_size // 2)
            return self.norm(t), H, W
        x = self.proj(x.contiguous(memory_format=torch.channels_last))   # no-op when the producer already is channels_last
        H, W = x.shape[-2:]
        return self.norm(_nchw_to_tokens(x)), H, W


class Attention(nn.Module):
    """cod.py:862-921."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        assert dim // num_heads == 64, "the gfx950 attention kernel is specialised for head_dim 64 (cod.py:1785)"
        assert attn_drop == 0. and proj_drop == 0., "reference configs use zero dropout (cod.py:1787)"
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.q = Linear(dim, dim, bias=qkv_bias)
        self.kv = Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = LayerNorm(dim, eps=1e-5)
        self.apply(_init_weights)

    def forward(self, x, H, W):
        return self.proj(self.core(x, H, W))

    def core(self, x, H, W, run=ops.NO_RUN):
        """Everything up to (not including) the output projection; Block fuses the projection with its residual epilogue.
        Deferred Linears of a Block (ops.block_run): 0 = q, 1 = kv, 2 = proj, 3 = fc1, 4 = fc2."""
        q = self.q(x)
        if self.sr_ratio > 1:  # k == s conv == patchify + GEMM, stays token-major
            w, b = wb(self.sr)
            r = ops.linear(_patchify(x, H, W, self.sr_ratio), w.flatten(1), b)
            run.roles(out=1)                      # the reduced tokens after their LayerNorm = input of kv
            r = self.norm(r)
        else:
            r = x                                 # kv reads the same arena slot as q
        kv = self.kv(r)
        run.roles(out=2, grad_a=0, grad_b=1)      # attention output = input of proj; its backward produces d(q out) and d(kv out)
        return ops.sra_attention(q, kv, self.num_heads, self.scale)


class DWConv(nn.Module):
    """cod.py:1520-1531."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W, gelu: bool = False):
        B, N, Cc = x.shape
        return ops.dwconv_nhwc(x.view(B, H, W, Cc), *wb(self.dwconv), gelu).view(B, N, Cc)


class Mlp(nn.Module):
    """cod.py:824-859."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.fc2 = Linear(hidden_features, out_features)
        self.apply(_init_weights)

    def forward(self, x, H, W):
        return self.fc2(self.hidden(x, H, W))

    def hidden(self, x, H, W, run=ops.NO_RUN):
        h = self.fc1(x)
        run.roles(out=4, grad_a=3)                # dw3x3 output = input of fc2; its backward produces d(fc1 out)
        return self.dwconv(h, H, W, gelu=True)    # dw3x3 + bias + exact GELU in one pass


class Block(nn.Module):
    """cod.py:924-961."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=None, sr_ratio=1):
        super().__init__()
        eps = 1e-6  # pvt_v2_b2 passes partial(nn.LayerNorm, eps=1e-6) (cod.py:1786)
        self.norm1 = LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, sr_ratio=sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = LayerNorm(dim, eps=eps)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.apply(_init_weights)

    def _scale(self, x):
        return self.drop_path.scale(x.shape[0], x.device) if isinstance(self.drop_path, DropPath) else None

    def forward(self, x, H, W, run=ops.NO_RUN):
        if _USE["fused_linear"] and x.is_cuda:   # projection / fc2 + DropPath + residual as one node (bias gradient fused in its backward)
            run.roles(out=0)                      # norm1 output = input of q (and of kv without spatial reduction)
            v, xs = self.norm1.fork(x) if _USE["ln_fork"] else (self.norm1(x), x)
            x = ops.linear_residual(self.attn.core(v, H, W, run), *wb(self.attn.proj), xs, self._scale(x))
            run.roles(out=3)                      # norm2 output = input of fc1
            v, xs = self.norm2.fork(x) if _USE["ln_fork"] else (self.norm2(x), x)
            return ops.linear_residual(self.mlp.hidden(v, H, W, run), *wb(self.mlp.fc2), xs, self._scale(x))
        x = ops.scale_residual(x, self.attn(self.norm1(x), H, W), self._scale(x))   # x + DropPath(attn), one pass
        return ops.scale_residual(x, self.mlp(self.norm2(x), H, W), self._scale(x))


# ------------------------------------------------------------------------------------------------ diffuser
class convnext_Block(nn.Module):
    """cod.py:1082-1117, operating on NHWC tokens [B, H, W, C]."""

    def __init__(self, dim, drop_path=0., layer_scale_init_value=1e-6):
        super().__init__()
        self.dwconv = Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = Linear(dim, 4 * dim)
        self.pwconv2 = Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) if layer_scale_init_value > 0 else None
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward_nhwc(self, x, run=ops.NO_RUN):
        s = self.drop_path.scale(x.shape[0], x.device) if isinstance(self.drop_path, DropPath) else None
        if _USE["fused_linear"] and x.is_cuda:   # pwconv1+GELU and pwconv2+gamma+DropPath+residual as two nodes with fused backward passes
            y, xs = ops.dwconv_fork(x, *wb(self.dwconv)) if _USE["ln_fork"] else (ops.dwconv_nhwc(x, *wb(self.dwconv)), x)
            run.roles(out=0)                      # LayerNorm output = input of pwconv1 (deferred Linear 0 of the block)
            y = self.norm(y)
            run.roles(out=1)                      # GELU output = input of pwconv2 (deferred Linear 1)
            if _USE["mlp_fused"]:                 # both Linears + GELU + layer scale + DropPath + residual: one node on the package's own GEMM
                return ops.mlp_residual(y, *wb(self.pwconv1), *wb(self.pwconv2), xs, s, self.gamma)
            h = ops.linear_gelu(y, *wb(self.pwconv1))
            return ops.linear_residual(h, *wb(self.pwconv2), xs, s, self.gamma)
        y = ops.dwconv_nhwc(x, *wb(self.dwconv))
        y = self.pwconv2(F.gelu(self.pwconv1(self.norm(y))))
        return ops.scale_residual(x, y, s, self.gamma)   # x + DropPath(gamma * y) in one pass

    def forward(self, x):  # reference contract: NCHW in, NCHW out
        return self.forward_nhwc(x.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)


class ShapePropEncoder(nn.Module):
    """cod.py:1119-1177: ConvNeXt-B trunk (NHWC inside) + 4 taps -> out_dim channels at stride 4 (NCHW out)."""

    def __init__(self, in_channels, out_dim):
        super().__init__()
        dims, depths = [128, 256, 512, 1024], [3, 3, 27, 3]
        self.dims = dims
        self.downsample_layers = nn.ModuleList(
            [nn.Sequential(Conv2d(3, dims[0], kernel_size=4, stride=4), LayerNorm(dims[0], eps=1e-6, data_format="channels_first"))]
            + [nn.Sequential(LayerNorm(dims[i], eps=1e-6, data_format="channels_first"),
                             Conv2d(dims[i], dims[i + 1], kernel_size=2, stride=2)) for i in range(3)])
        rates = [x.item() for x in torch.linspace(0, 0.4, sum(depths))]  # cod.py:1140-1145
        self.stages, cur = nn.ModuleList(), 0
        for i in range(4):
            self.stages.append(nn.Sequential(*[convnext_Block(dims[i], rates[cur + j], 1.0) for j in range(depths[i])]))
            cur += depths[i]
        self.convs = nn.ModuleList([Conv2d(dims[i], out_dim, 1) for i in range(4)])
        self.fusion_conv = Conv2d(out_dim * 4, out_dim, 1)

    def forward(self, x):
        B, _, H, W = x.shape
        outs = []
        # stem: conv k4 s4 == patchify + GEMM
        conv, norm = self.downsample_layers[0]
        t = x.view(B, 3, H // 4, 4, W // 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 4) * (W // 4), 48)
        w, b = wb(conv)
        t = norm(ops.linear(t, w.flatten(1), b))
        h, w = H // 4, W // 4
        for i in range(4):
            if i > 0:
                norm, conv = self.downsample_layers[i]
                cw, cb = wb(conv)
                t = ops.linear(_patchify(norm(t), h, w, 2), cw.flatten(1), cb)
                h, w = h // 2, w // 2
            ops.mark_flush_point(t)            # backward: this stage (and everything after it) is done when this gradient arrives
            t4 = t.view(B, h, w, -1)
            with ops.block_run(self, i, len(self.stages[i]), t4) as run:     # identical blocks: their weight-gradient GEMMs batch
                for j, blk in enumerate(self.stages[i]):
                    run.at(j)
                    t4 = blk.forward_nhwc(t4, run)
            t = t4.reshape(B, h * w, -1)
            outs.append((t, h, w))
        size = (outs[0][1], outs[0][2])
        taps = []
        for (t, h, w), conv in zip(outs, self.convs):  # 1x1 conv == GEMM on tokens, then bilinear to stride 4
            cw, cb = wb(conv)
            y = _tokens_to_nchw(ops.linear(t, cw.flatten(1), cb), h, w)
            if (h, w) == size:
                taps.append(y)                                   # bilinear resize to the same size is the identity
            elif _USE["bilinear"] and y.is_cuda and y.shape[1] % (16 // y.element_size()) == 0:
                taps.append(ops.bilinear_resize(y, size[0], size[1], False))
            else:
                taps.append(F.interpolate(y, size=size, mode="bilinear", align_corners=False))
        cat = ops.cat_channels(taps)
        fw, fb = wb(self.fusion_conv)                            # 1x1 conv == GEMM on tokens
        Bc, Cc, Hc, Wc = cat.shape
        return _tokens_to_nchw(ops.linear(_nchw_to_tokens(cat), fw.flatten(1), fb), Hc, Wc) if cat.is_cuda else self.fusion_conv(cat)


class ShapePropWeightRegressor(nn.Module):
    """cod.py:1051-1060."""

    def __init__(self, in_channels, latent_dim):
        super().__init__()
        self.latent_dim = latent_dim
        self.reg = Conv2d(in_channels, latent_dim * 49, kernel_size=1)

    def forward(self, x):
        return torch.sigmoid(self.reg(x))


class MessagePassing(nn.Module):
    """cod.py:1180-1208 (random-walk normalisation branch)."""

    def __init__(self, latent_dim, img_size=384, k=7, max_step=4, sym_norm=False):
        super().__init__()
        assert not sym_norm, "the reference's sym_norm branch references undefined attributes (cod.py:1194-1198)"
        self.k, self.size, self.max_step, self.img_size = k, k * k, max_step, img_size
        self.conv = Conv2d(latent_dim, 3, 1)

    def forward(self, input, weight):
        n, c, h, w = input.shape
        wgt = weight.view(n, weight.shape[1] // self.size, self.size, h * w)
        wgt = wgt / (wgt.sum(2, keepdim=True) + 1e-5)
        x = input
        for _ in range(max(h, w) if self.max_step < 0 else self.max_step):
            x = (F.unfold(x, self.k, padding=self.k // 2).view(n, c, self.size, h * w) * wgt).sum(2).view(n, c, h, w)
        x = self.conv(x)
        return F.interpolate(x, size=(self.img_size, self.img_size), mode="bilinear", align_corners=False)


def fft_highpass(x: torch.Tensor, rate: float) -> torch.Tensor:
    """cod.py:1256-1271 on the tensor's own device (rocFFT); no gradient path (input has none)."""
    w, h = x.shape[-2:]
    line = int((w * h * rate) ** .5 // 2)
    x = x.float()
    f = torch.fft.fftshift(torch.fft.fft2(x, norm="forward"), dim=(-2, -1))
    f[..., w // 2 - line:w // 2 + line, h // 2 - line:h // 2 + line] = 0
    return torch.fft.ifft2(torch.fft.ifftshift(f, dim=(-2, -1)), norm="forward").real.abs()


class prompt_encoder(nn.Module):
    """cod.py:1228-1306."""

    def __init__(self, latent_dim, embed_dim, depth, fusion=False):
        super().__init__()
        self.embed_dim, self.depth = embed_dim, depth
        self.propagation_weight_regressor = ShapePropWeightRegressor(3, latent_dim)
        self.encoder1 = Conv2d(1, latent_dim, 1)
        self.encoder2 = ShapePropEncoder(3, 24)
        self.adaptor = Conv2d(6, 3, 1)  # constructed, never called (cod.py:1251)
        self.message_passing = MessagePassing(latent_dim, img_size=384, sym_norm=False)
        self.freq_nums = 0.3

    def fft(self, x, rate):
        return fft_highpass(x, rate)

    def forward(self, image, cues, cross=False, x_hp=None):
        with torch.autocast("cuda", enabled=False):
            image32 = image.float()
            x = self.fft(image32, self.freq_nums) if x_hp is None else x_hp
            # one fused launch per (image, latent channel): regressor + depth embedding + 4 propagation steps
            reg = self.propagation_weight_regressor.reg
            x4 = ops.diffuser_state(x, cues, *wb(reg), *wb(self.encoder1))
            # 1x1 conv 24->3, bilinear 12 -> S (the reference pins 384, cod.py:1252), + image: one HBM-bound pass
            mp = self.message_passing.conv
            fused = ops.diffuse_tail(x4, *wb(mp), image32)
        return x, self.encoder2(fused)


class ShapePropDecoder(nn.Module):
    """cod.py:1210-1226."""

    def __init__(self, out_dim, latent_dim):
        super().__init__()
        self.decoder = nn.Sequential(Conv2d(latent_dim, latent_dim, 3, 1, 1), nn.ReLU(True),
                                     Conv2d(latent_dim, latent_dim, 3, 1, 1), nn.ReLU(True),
                                     Conv2d(latent_dim, out_dim, 3, 1, 1))

    def forward(self, embedding):
        return self.decoder(embedding)

    def forward_tokens(self, embedding, H, W, trunk=None):
        """== F.interpolate(self.decoder(embedding), (H, W), 'bilinear') as tokens [B, H*W, C] (cod.py:1471), without ever
        materialising the full-resolution prompt: for an integer power-of-two down-scale s the align_corners=False bilinear
        sample is the mean of the 2x2 centre pixels, and mean-of-conv3x3 == one 4x4 convolution with stride s whose
        kernel is the mean of the four shifted 3x3 kernels (zero padding carries over unchanged)."""
        # ``trunk``: the output of the two conv3x3+ReLU layers, NHWC, when the caller ran them for all decoders in one launch
        h = trunk.permute(0, 3, 1, 2) if trunk is not None else self.decoder[3](self.decoder[2](self.decoder[1](self.decoder[0](embedding))))
        conv = self.decoder[4]
        w, b = wb(conv)
        Hin = h.shape[-2]
        s_ = Hin // H
        if s_ == 1:
            if _USE["conv3x3"] and w.dtype == h.dtype and ops.conv3x3_ops.supported(h, w.shape[1], w.shape[0], Hin, h.shape[-1]):
                y = ops.conv3x3(h, w, b)                # stage-1 prompts (24 -> 64 at S/4): same kernel as the trunk layers
            elif h.is_cuda and _USE["conv_gemm"]:
                return ops.conv2d_tokens(h, w, b, 1, 1)[0]
            else:
                y = F.conv2d(h, w, b, padding=1)
        elif s_ in (2, 4, 8) and Hin == H * s_ and h.shape[-1] == W * s_:
            w4 = _fold_2x2_mean(w)                 # mean of the four shifted 3x3 kernels (zero padded), one launch
            off = s_ // 2 - 2                      # first input row/col of the 4x4 window of output 0
            if h.is_cuda and _USE["conv_gemm"]:
                # the offset window grid is a strided view the patch gather reads directly (s_ = 4: the windows tile the map)
                src, pad = (h, -off) if off < 0 else (h[:, :, off:, off:], 0)
                t, Ht, Wt = ops.conv2d_tokens(src, w4, b, s_, pad)
                assert (Ht, Wt) == (H, W)
                return t
            if off < 0:
                y = F.conv2d(h, w4, b, stride=s_, padding=-off)
            else:
                y = F.conv2d(h[:, :, off:, off:], w4, b, stride=s_)
        else:                                      # any other geometry: the literal reference sequence
            y = F.interpolate(F.conv2d(h, w, b, padding=1), size=(H, W), mode="bilinear", align_corners=False)
        return _nchw_to_tokens(y)


_FOLD = {}


def _fold_2x2_mean(w: torch.Tensor) -> torch.Tensor:
    """[O,I,3,3] -> [O,I,4,4]: w4[a,b] = mean_{h in {a-1,a}, v in {b-1,b}} w[h,v] (zero outside) = the kernel of `mean of the 2x2
    centre pixels of conv3x3(.)` as ONE 4x4 convolution.  Written as a constant [16,9] matrix applied to the taps (a batched GEMM
    over the output channels; both views are free for the O,H,W,I storage the gradient reducer gives these weights).
    NOT F.avg_pool2d(w, 2, 1, 1): on ROCm 7.2 / torch 2.10 its BACKWARD is wrong for channels_last inputs (max error 2.3 on N(0,1)
    data, tools/debug_prompt_tail.py) - with the reducer's channels_last weights that corrupted these 13 weight gradients."""
    key = (w.device, w.dtype)
    T = _FOLD.get(key)
    if T is None:
        T = torch.zeros(16, 9)
        for a in range(4):
            for b in range(4):
                for hh in (a - 1, a):
                    for v in (b - 1, b):
                        if 0 <= hh < 3 and 0 <= v < 3:
                            T[a * 4 + b, hh * 3 + v] = 0.25
        T = _FOLD[key] = T.to(device=w.device, dtype=w.dtype)
    O, I = w.shape[:2]
    w4 = torch.matmul(T, w.permute(0, 2, 3, 1).reshape(O, 9, I))      # [O,16,I]
    return w4.view(O, 4, 4, I).permute(0, 3, 1, 2)                    # logical [O,I,4,4] in channels_last memory


class prompt_decoder(nn.Module):
    """cod.py:1308-1323."""

    def __init__(self, latent_dim, embed_dim, depth, fusion=False):
        super().__init__()
        self.depth = depth
        self.decoder = nn.Sequential(*[ShapePropDecoder(embed_dim, 24) for _ in range(depth)])

    def forward(self, embedding, cross=False):
        return [self.decoder[i](embedding) for i in range(self.depth)]

    def forward_tokens(self, embedding, H, W, trunks=None):
        return [self.decoder[i].forward_tokens(embedding, H, W, None if trunks is None else trunks[i]) for i in range(self.depth)]


class PyramidVisionTransformerImpr(nn.Module):
    """cod.py:1340-1517."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=[64, 128, 256, 512],
                 num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=None, depths=[3, 4, 6, 3], sr_ratios=[8, 4, 2, 1]):
        super().__init__()
        self.num_classes, self.depths = num_classes, depths
        chans = [in_chans] + list(embed_dims[:3])
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(4):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(img_size=img_size // (1 if i == 0 else 2 ** (i + 1)),
                                                                   patch_size=7 if i == 0 else 3, stride=4 if i == 0 else 2,
                                                                   in_chans=chans[i], embed_dim=embed_dims[i]))
            setattr(self, f"block{i + 1}", nn.ModuleList([
                Block(dim=embed_dims[i], num_heads=num_heads[i], mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[cur + j],
                      sr_ratio=sr_ratios[i]) for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", LayerNorm(embed_dims[i], eps=1e-6))
            cur += depths[i]
        self.latent_dim = LATENT
        self.prompt_encoder = prompt_encoder(self.latent_dim, embed_dims, depths, True)
        self.prompt_decoder = nn.Sequential(*[prompt_decoder(self.latent_dim, embed_dims[i], depths[i], True) for i in range(4)])
        self.apply(_init_weights)
        self.batch = 0

    def forward_features(self, x, depth, x_hp=None):
        self.batch += 1
        B = x.shape[0]
        image = x
        embedding1, embedding3 = self.prompt_encoder(image, depth, x_hp=x_hp)
        embedding3 = embedding3.contiguous(memory_format=torch.channels_last)
        ops.mark_flush_point(embedding3)       # backward: Hitnet decoder, all PVT blocks and the prompt decoders are done at this gradient
        if _USE["async_flush"] and embedding3.requires_grad and embedding3.is_cuda:
            # backward: when this gradient arrives, the Hitnet decoder, every PVT block and the prompt decoders are done and the whole
            # ConvNeXt trunk is still to go - their parked weight-gradient work (batched GEMMs, depthwise / 3x3 weight gradients) starts
            # on a side stream now and runs beside the ConvNeXt backward pass instead of after it (bindings.cpp flush_deferred_async)
            nat = ops._native.ops()
            if nat is not None:
                embedding3.register_hook(_async_flush_hook)
        trunks, off = self._prompt_trunks(embedding3), 0
        outs = []
        for i in range(4):
            x, H, W = getattr(self, f"patch_embed{i + 1}")(x)
            d = self.prompt_decoder[i].depth
            prompts = self.prompt_decoder[i].forward_tokens(embedding3, H, W, None if trunks is None else trunks[off:off + d])
            off += d                                                            # prompts: already at (H, W), already tokens
            blocks = getattr(self, f"block{i + 1}")
            with ops.block_run(self, i, len(blocks), x) as run:                 # identical blocks: their weight-gradient GEMMs batch
                for j, blk in enumerate(blocks):
                    run.at(j)
                    x = blk(x + prompts[j], H, W, run)
            x = getattr(self, f"norm{i + 1}")(x)
            x = _tokens_to_nchw(x, H, W)   # channels_last view: feeds the next patch embed and the Hitnet decoder as is
            outs.append(x)
        return embedding1, outs

    def _prompt_trunks(self, embedding):
        """The two conv3x3(24->24)+ReLU layers of ALL prompt decoders (cod.py:1216-1220; 16 for pvt_v2_b2) as two launches of the
        batched NHWC MFMA kernel: the first reads the shared embedding once per tile, the second maps [Z,B,h,w,24] -> same.
        Returns a tuple of Z NHWC [B,h,w,24] maps, or None when the geometry / dtype is outside the kernel (fp32 parity mode, CPU)."""
        decs = [d for pd in self.prompt_decoder for d in pd.decoder]
        c0, c1 = [d.decoder[0] for d in decs], [d.decoder[2] for d in decs]
        w0 = [wb(c)[0] for c in c0]
        B, C, H, W = embedding.shape
        if not (_USE["conv3x3"] and ops.conv3x3_ops.supported(embedding, C, C, H, W) and all(w.dtype == embedding.dtype for w in w0)):
            return None
        x = embedding.permute(0, 2, 3, 1).contiguous()
        h = ops.conv3x3_stack(x, w0, [wb(c)[1] for c in c0], True)
        return ops.unstack(ops.conv3x3_stack(h, [wb(c)[0] for c in c1], [wb(c)[1] for c in c1], True))   # Z views, one gather in the backward

    def forward(self, x, depth, x_hp=None):
        return self.forward_features(x, depth, x_hp)


def _async_flush_hook(grad):
    nat = ops._native.ops()
    if nat is not None:
        nat.flush_deferred_async()
    return None


def _pvt_variant(name, depths, mlp_ratios, doc):
    def __init__(self, **kwargs):
        PyramidVisionTransformerImpr.__init__(self, patch_size=4, embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8],
                                              mlp_ratios=mlp_ratios, qkv_bias=True, depths=depths, sr_ratios=[8, 4, 2, 1],
                                              drop_rate=0.0, drop_path_rate=kwargs.get("drop_path_rate", 0.1))
    return type(name, (PyramidVisionTransformerImpr,), {"__init__": __init__, "__doc__": doc})


# cod.py:1773-1812.  Every variant with head_dim 64 (the attention kernel's specialisation); b0 (head_dim 32) is not on any path.
pvt_v2_b1 = _pvt_variant("pvt_v2_b1", [2, 2, 2, 2], [8, 8, 4, 4], "cod.py:1773-1779")
pvt_v2_b2 = _pvt_variant("pvt_v2_b2", [3, 4, 6, 3], [8, 8, 4, 4], "cod.py:1781-1787 — the backbone Hitnet hard-codes (cod.py:689)")
pvt_v2_b3 = _pvt_variant("pvt_v2_b3", [3, 4, 18, 3], [8, 8, 4, 4], "cod.py:1789-1795")
pvt_v2_b4 = _pvt_variant("pvt_v2_b4", [3, 8, 27, 3], [8, 8, 4, 4], "cod.py:1797-1803")
pvt_v2_b5 = _pvt_variant("pvt_v2_b5", [3, 6, 40, 3], [4, 4, 4, 4], "cod.py:1806-1812")
PVT_VARIANTS = {"pvt_v2_b1": pvt_v2_b1, "pvt_v2_b2": pvt_v2_b2, "pvt_v2_b3": pvt_v2_b3, "pvt_v2_b4": pvt_v2_b4, "pvt_v2_b5": pvt_v2_b5}


# ------------------------------------------------------------------------------------------------ Hitnet decoder
class BasicConv2d(nn.Module):
    """cod.py:355-368 (the ReLU is constructed but never applied)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__()
        self.conv = Conv2d(in_planes, out_planes, kernel_size, stride, padding, dilation, bias=False)
        self.bn = nn.BatchNorm2d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def _bn(self, t):
        """cod.py:366: the fused statistics + apply kernels (two launches each way) where they tile the map, torch's otherwise."""
        if _USE["batchnorm"] and ops.batch_norm_supported(t, self.bn):
            return ops.batch_norm(t, self.bn)
        return self.bn(t)

    def forward(self, x):
        c = self.conv
        w = wb(c)[0]
        if (_USE["conv3x3"] and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1)
                and w.dtype == x.dtype and ops.conv3x3_ops.supported(x, c.in_channels, c.out_channels, x.shape[2], x.shape[3])):
            return self._bn(ops.conv3x3(x, w))        # conv4 (96 -> 32 at S/8, cod.py:713): NHWC bf16 MFMA kernel
        if x.is_cuda and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0):
            # 1x1 conv == GEMM on tokens (cached hipBLASLt plan: a third of the host cost of a MIOpen call); views both ways
            Bc, _, Hc, Wc = x.shape
            return self._bn(_tokens_to_nchw(ops.linear(_nchw_to_tokens(x), w.flatten(1), None), Hc, Wc))
        return self._bn(self.conv(x))


class CALayer(nn.Module):
    """cod.py:415-431."""

    def __init__(self, channel, reduction=16, bias=False):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(Conv2d(channel, channel // reduction, 1, bias=bias), nn.ReLU(inplace=True),
                                     Conv2d(channel // reduction, channel, 1, bias=bias), nn.Sigmoid())
        self.conv_du[0].keep_master = self.conv_du[2].keep_master = True   # tiny fp32 MLP inside ops.ca_gate: no working copy

    def forward(self, x):
        return x * self.conv_du(self.avg_pool(x))


class CAB(nn.Module):
    """cod.py:436-451."""

    def __init__(self, n_feat, kernel_size, reduction, bias, act):
        super().__init__()
        pad = kernel_size // 2
        self.CA = CALayer(n_feat, reduction, bias=bias)
        self.body = nn.Sequential(Conv2d(n_feat, n_feat, kernel_size, padding=pad, bias=bias), act,
                                  Conv2d(n_feat, n_feat, kernel_size, padding=pad, bias=bias))

    def forward(self, x):
        act, du = self.body[1], self.CA.conv_du
        if _USE["cab_glue"] and x.is_cuda and isinstance(act, nn.PReLU) and act.weight.numel() == 1 and du[0].bias is None and du[2].bias is None:
            c0, c1 = self.body[0], self.body[2]
            w0, w1 = wb(c0)[0], wb(c1)[0]
            if (_USE["conv3x3"] and c0.bias is None and c1.bias is None and w0.dtype == x.dtype
                    and ops.conv3x3_ops.supported(x, x.shape[1], x.shape[1], x.shape[2], x.shape[3])):
                if _USE["cab_node"]:             # the whole CAB as one node: PReLU (+ backward) and the skip gradient in conv epilogues
                    out = ops.cab(x, w0, w1, act.weight, du[0].weight, du[2].weight)
                    if out is not None:
                        return out
                res = ops.conv3x3(ops.prelu(ops.conv3x3(x, w0), act.weight), w1)      # NHWC bf16 MFMA kernels
            else:
                res = c1(ops.prelu(c0(x), act.weight))
            return ops.ca_gate(res, x, du[0].weight, du[2].weight)     # gate * res + x in three launches
        return self.CA(self.body(x)) + x


class SAM(nn.Module):
    """cod.py:454-506."""

    def __init__(self, ch_in=32, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(Linear(ch_in, ch_in // reduction, bias=False), nn.ReLU(inplace=True),
                                Linear(ch_in // reduction, ch_in, bias=False), nn.Sigmoid())
        self.fc_wight = nn.Sequential(Linear(ch_in, ch_in // reduction, bias=False), nn.ReLU(inplace=True),
                                      Linear(ch_in // reduction, 1, bias=False), nn.Sigmoid())
        for lin in (self.fc[0], self.fc[2], self.fc_wight[0], self.fc_wight[2]):
            lin.keep_master = True                   # tiny fp32 MLPs inside ops.sam: no working copies

    def _gate(self, x):
        y = x.float().mean((2, 3))
        return x * (self.fc(y) * self.fc_wight(y)).to(x.dtype)[:, :, None, None]

    def forward(self, x_h, x_l):
        if _USE["sam_node"] and x_h.is_cuda and x_l.shape == x_h.shape and ops.sam_supported(x_h, self.fc[0].weight.shape[0]):
            return ops.sam(x_h, x_l, self.fc[0].weight, self.fc[2].weight, self.fc_wight[0].weight, self.fc_wight[2].weight)
        return self._gate(x_h) + self._gate(x_l)


class _ChannelAttention(nn.Module):  # cod.py:371-387; constructed (cod.py:703), never called
    def __init__(self, in_planes):
        super().__init__()
        self.fc1 = Conv2d(in_planes, in_planes // 16, 1, bias=False)
        self.fc2 = Conv2d(in_planes // 16, in_planes, 1, bias=False)


class _SpatialAttention(nn.Module):  # cod.py:390-405; constructed (cod.py:704), never called
    def __init__(self):
        super().__init__()
        self.conv1 = Conv2d(2, 1, 7, padding=3, bias=False)


def _up(x, scale, align):
    """Bilinear resize in the dtype of ``x``: under CUDA autocast the resampler would otherwise widen bf16 maps to fp32 and every
    consumer (cat, conv, CAB) would cast them back."""
    if _USE["bilinear"] and x.is_cuda and x.shape[1] % (16 // x.element_size()) == 0:
        return ops.bilinear_resize(x, int(math.floor(x.shape[2] * scale)), int(math.floor(x.shape[3] * scale)), align)
    if x.is_cuda and x.dtype != torch.float32:
        with torch.autocast("cuda", enabled=False):
            return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=align)
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=align)


class Hitnet(nn.Module):
    """cod.py:685-807."""

    def __init__(self, channel=32, n_feat=32, scale_unetfeats=32, kernel_size=3, reduction=4, bias=False, act=None,
                 drop_path_rate=0.1, backbone: str = "pvt_v2_b2"):
        super().__init__()
        act = act if act is not None else nn.PReLU()  # ONE shared instance (default argument at cod.py:686)
        # the reference hard-codes pvt_v2_b2 (cod.py:689); b1/b3/b4/b5 fit the same 64/128/320/512 Translayers ("tier B")
        self.backbone = PVT_VARIANTS[backbone](drop_path_rate=drop_path_rate)
        if drop_path_rate == 0.0:
            for m in self.backbone.modules():
                if isinstance(m, DropPath):
                    m.drop_prob = 0.0
        self.Translayer2_0 = BasicConv2d(64, channel, 1)
        self.Translayer2_1 = BasicConv2d(128, channel, 1)
        self.Translayer3_1 = BasicConv2d(320, channel, 1)
        self.Translayer4_1 = BasicConv2d(512, channel, 1)
        self.ca = _ChannelAttention(64)
        self.sa = _SpatialAttention()
        self.SAM = SAM()
        self.out_SAM = Conv2d(channel, 1, 1)
        self.out_CFM = Conv2d(channel, 1, 1)
        mk = lambda ch: nn.Sequential(*[CAB(ch, kernel_size, reduction, bias=bias, act=act) for _ in range(2)])
        self.decoder_level4 = mk(n_feat)
        self.decoder_level3 = mk(n_feat + scale_unetfeats)
        self.decoder_level2 = mk(n_feat + scale_unetfeats * 2)
        self.conv4 = BasicConv2d(3 * channel, channel, 3, padding=1)
        self.decoder_level1 = mk(64)
        self.compress_out = BasicConv2d(2 * channel, channel, kernel_size=8, stride=4, padding=2)
        self.compress_out2 = BasicConv2d(2 * channel, channel, kernel_size=1)

    def forward(self, x, pred_normal, x_hp=None, lowres: bool = False):
        """``lowres=True`` returns the five head maps BEFORE their x8 up-sampling (the fused training loss samples them on
        the fly, ops.seg_loss); the default returns the reference's full-resolution maps (cod.py:796, :806)."""
        embedding1, (x1, x2, x3, x4) = self.backbone(x, pred_normal, x_hp)
        cim = self.decoder_level1(x1)
        x2_t, x3_t, x4_t = self.Translayer2_1(x2), self.Translayer3_1(x3), self.Translayer4_1(x4)
        stage_loss, cfm = [], None
        for it in range(4):
            if cfm is not None:
                x4_t = self.compress_out(ops.cat_channels((_up(x4_t, 4, True), cfm)))
            x4f = self.decoder_level4(x4_t)
            x3f = self.decoder_level3(ops.cat_channels((x3_t, _up(x4f, 2, True))))
            if it > 0:
                x2_t = self.compress_out2(ops.cat_channels((x2_t, cfm)))
            x2f = self.decoder_level2(ops.cat_channels((x2_t, _up(x3f, 2, True))))
            cfm = self.conv4(x2f)
            pred = self.out_CFM(cfm).float()
            stage_loss.append(pred if lowres else _up(pred, 8, False))
        T2 = _up(self.Translayer2_0(cim), 0.5, True)
        pred2 = self.out_SAM(self.SAM(cfm, T2)).float()
        return embedding1, stage_loss, (pred2 if lowres else _up(pred2, 8, False))


# ------------------------------------------------------------------------------------------------ losses / top level
def loss_weight(gts):
    """cod.py:77: the 31x31 box-filter edge weight.  It depends on the label only, so one evaluation serves all five
    cal_loss calls of a step (the reference recomputes it five times)."""
    return 1 + 5 * torch.abs(F.avg_pool2d(gts, kernel_size=31, stride=1, padding=15) - gts)


def cal_loss(preds, gts, weit=None):
    """cod.py:76-85."""
    if weit is None:
        weit = loss_weight(gts)
    wbce = F.binary_cross_entropy_with_logits(preds, gts, reduction="none")
    wbce = (weit * wbce).sum(dim=(2, 3)) / weit.sum(dim=(2, 3))
    p = torch.sigmoid(preds)
    inter = ((p * gts) * weit).sum(dim=(2, 3))
    union = ((p + gts) * weit).sum(dim=(2, 3))
    return (wbce + 1 - (inter + 1) / (union - inter + 1)).mean()


def ssim_value(x, y):
    """cod.py:316-351 (value only; it has no gradient path to any parameter)."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    x, y = F.pad(x, (1, 1, 1, 1), mode="reflect"), F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sx = F.avg_pool2d(x * x, 3, 1) - mu_x ** 2
    sy = F.avg_pool2d(y * y, 3, 1) - mu_y ** 2
    sxy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sxy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sx + sy + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1).mean(1, True).mean()


def _stack(t):
    return torch.stack(list(t), dim=0) if isinstance(t, (tuple, list)) else t


class cod(nn.Module):
    """cod.py:36-224: ``forward(raw, input, label, depth, mode)`` with mode in {'loss', 'predict'}.

    Extra keyword arguments (ignored by the reference, cod.py:38-46) are accepted and ignored.
    ``compute_dtype``: torch.float32 = exact-fp32 kernels (parity mode); torch.bfloat16 = bf16 MFMA
    kernels + library GEMM/conv under bf16 autocast with fp32 master weights (throughput mode); torch.float16 = the same
    with IEEE half (fp16 MFMA) - the reference's own AMP recipe (AmpOptimWrapper, config/sod.yml:57): pair it with
    ``runner.LossScaler`` + ``runner.FlatAdamW(scaler=...)``."""

    def __init__(self, win_size=None, filter_ratio=None, using_depth=None, using_sam=None, finetune=None,
                 binary_thresh=None, pretrain_sam=None, head=None, img_size: int = 384,
                 compute_dtype: torch.dtype = torch.float32, drop_path_rate: float = 0.1, backbone: str = "pvt_v2_b2"):
        super().__init__()
        self.hitnet = Hitnet(drop_path_rate=drop_path_rate, backbone=backbone)
        self.batch = 0
        self.compute_dtype = compute_dtype
        self._dp_plan = {"masks": None}
        self._dp_layers = [m for m in self.modules() if isinstance(m, DropPath) and m.drop_prob > 0.0]
        for i, m in enumerate(self._dp_layers):
            m.index, m.plan = i, self._dp_plan
        self.register_buffer("_dp_keep", torch.tensor([[1.0 - m.drop_prob] for m in self._dp_layers] or [[1.0]]), persistent=False)

    @staticmethod
    def grad_readiness_rank(name: str) -> int:
        """Order in which parameter groups finish in the BACKWARD pass (dist.GradReducer buckets follow it): Hitnet decoder and heads,
        then the PVT stages, then the prompt decoders, then the texture diffuser (ConvNeXt trunk, regressor, depth embedding) - the
        diffuser is upstream of everything (cod.py:1455-1472) although it is registered between the PVT stages and the prompt decoders."""
        if not name.startswith("hitnet.backbone."):
            return 0
        if name.startswith("hitnet.backbone.prompt_encoder."):
            return 3
        if name.startswith("hitnet.backbone.prompt_decoder."):
            return 2
        return 1

    def _draw_drop_path(self, batch: int) -> None:
        """All stochastic-depth masks of the step in one shot: [n_layers, B] of {0, 1/keep} (cod.py:935, :1102 per layer)."""
        if self.training and self._dp_layers:
            keep = self._dp_keep
            self._dp_plan["masks"] = (torch.rand(keep.shape[0], batch, device=keep.device) < keep).float() / keep
        else:
            self._dp_plan["masks"] = None

    def _run(self, input, depth, x_hp=None, lowres=False):
        self._draw_drop_path(input.shape[0])
        if self.compute_dtype in (torch.bfloat16, torch.float16):
            with torch.autocast("cuda", dtype=self.compute_dtype):
                return self.hitnet(input, depth, x_hp, lowres)
        return self.hitnet(input, depth, x_hp, lowres)

    def high_pass(self, input):
        """The FFT high-pass image (cod.py:1288) on its own (it depends on the input only, so a data pipeline can
        precompute it and pass it as ``x_hp``)."""
        return fft_highpass(_stack(input).float(), self.hitnet.backbone.prompt_encoder.freq_nums)

    def forward(self, raw, input, label, depth, mode="loss", x_hp=None):
        input, label, depth = _stack(input), _stack(label), _stack(depth)
        if not input.is_cuda:
            raise RuntimeError("dgtd.nn.cod runs on the MI355X HIP device only; the CPU restatement is oracle/cod_cpu.py")
        if mode == "loss":
            embedding1, P1, P2 = self._run(input, depth, x_hp, lowres=True)
            # sum_it 0.2*it*cal_loss(P1[it]) + cal_loss(P2) (cod.py:137-142), up-sampling fused into the loss kernel
            loss = ops.seg_loss([*P1, P2], label, weights=(0.0, 0.2, 0.4, 0.6, 1.0))
            with torch.no_grad():
                if input.shape[-1] == input.shape[-2]:
                    loss3 = ops.ssim_value(embedding1, input)      # min-max normalisation + SSIM map + mean in one fused pass
                else:
                    e = (embedding1 - embedding1.min()) / (embedding1.max() - embedding1.min() + 1e-8)
                    loss3 = ssim_value(e, input.float())
            return {"loss": loss + loss3}
        embedding1, P1, P2 = self._run(input, depth, x_hp)
        if mode == "predict":  # cod.py:152-153 + :219 (the PNG dumps of cod.py:156-217 are dropped)
            out = F.interpolate(P1[-1] + P2, size=label.shape[-2:], mode="bilinear", align_corners=False)
            return out.sigmoid(), label
        raise NotImplementedError(f"Unsupported mode {mode}")
