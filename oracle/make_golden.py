"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the REAL reference
(/root/reference/twig/model/cod.py, imported through oracle/ref_loader.py) on deterministic
inputs with filler weights.  Runs only in the build container; the .npz files are data
(inputs + expected outputs), never reference source.

    python -m oracle.make_golden            # regenerate everything
    python -m oracle.make_golden model64    # one case

Every case is described by ``CASES[name]`` so tests/ can rebuild the same inputs for the
oracle restatement and for the HIP path.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from . import filler, ref_loader

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BB = "hitnet.backbone."


def tensor(key: str, shape, scale: float = 1.0) -> torch.Tensor:
    return (torch.from_numpy(filler.normal("input/" + key, int(np.prod(shape))).reshape(shape)).float() * scale)


def utensor(key: str, shape) -> torch.Tensor:
    return torch.from_numpy(filler.uniform("input/" + key, int(np.prod(shape))).reshape(shape)).float()


def digest(t: torch.Tensor, nsamp: int = 4096) -> dict:
    """Checksums + strided samples for tensors too big to commit whole."""
    f = t.detach().double().flatten()
    step = max(f.numel() // nsamp, 1)
    return {"sum": np.float64(f.sum().item()), "abssum": np.float64(f.abs().sum().item()),
            "sqsum": np.float64((f * f).sum().item()), "step": np.int64(step),
            "samples": f[::step].float().numpy()}


GRAD_SAMPLES = 16


def grad_samples(g: torch.Tensor) -> np.ndarray:
    """GRAD_SAMPLES evenly strided elements of the gradient in logical (contiguous) element order, zero padded."""
    f = g.detach().contiguous().flatten().float()
    step = max(f.numel() // GRAD_SAMPLES, 1)
    v = f[::step][:GRAD_SAMPLES].cpu().numpy()
    return np.pad(v, (0, GRAD_SAMPLES - v.size))


def sign_pattern(n: int, device="cpu") -> torch.Tensor:
    """Deterministic +-1 pattern over n elements (integer hash of the element index; no RNG)."""
    i = torch.arange(n, dtype=torch.int64, device=device)
    return (1 - 2 * (((i * 2654435761) >> 15) & 1)).double()


def grad_projection(g: torch.Tensor) -> float:
    f = g.detach().contiguous().flatten().double()
    return float((f * sign_pattern(f.numel(), f.device)).sum())


# (name, dim, heads, sr, H=W, B) — the four PVT-b2 stage shapes (cod.py:1785-1786) at S=128 and stage 1 at S=512
ATTN_CASES = [("attn_s1", 64, 1, 8, 32, 2), ("attn_s2", 128, 2, 4, 16, 2), ("attn_s3", 320, 5, 2, 8, 2),
              ("attn_s4", 512, 8, 1, 4, 2), ("attn_s1_512", 64, 1, 8, 128, 1)]


def gen_modules(ref) -> dict:
    out = {}
    torch.manual_seed(0)
    # ---- Attention (cod.py:862-921)
    for name, dim, heads, sr, hw, B in ATTN_CASES:
        m = ref.Attention(dim, num_heads=heads, qkv_bias=True, sr_ratio=sr).eval()
        stage = {64: 1, 128: 2, 320: 3, 512: 4}[dim]
        filler.fill_module(m, BB + f"block{stage}.0.attn.")
        x = tensor(name, (B, hw * hw, dim))
        y = m(x, hw, hw)
        if y.numel() <= 1 << 17:
            out[name] = y.detach().numpy()
        else:
            for k, v in digest(y).items():
                out[f"{name}.{k}"] = v
    # ---- OverlapPatchEmbed (cod.py:964-1004)
    m = ref.OverlapPatchEmbed(img_size=64, patch_size=7, stride=4, in_chans=3, embed_dim=64).eval()
    filler.fill_module(m, BB + "patch_embed1.")
    out["patch_embed1"] = m(tensor("patch_embed1", (2, 3, 64, 64)))[0].detach().numpy()
    m = ref.OverlapPatchEmbed(img_size=16, patch_size=3, stride=2, in_chans=64, embed_dim=128).eval()
    filler.fill_module(m, BB + "patch_embed2.")
    out["patch_embed2"] = m(tensor("patch_embed2", (2, 64, 16, 16)))[0].detach().numpy()
    # ---- Mlp + DWConv (cod.py:824-859, 1520-1531)
    m = ref.Mlp(in_features=64, hidden_features=512).eval()
    filler.fill_module(m, BB + "block1.0.mlp.")
    out["mlp1"] = m(tensor("mlp1", (2, 256, 64)), 16, 16).detach().numpy()
    # ---- Block (cod.py:924-961)
    m = ref.Block(dim=128, num_heads=2, mlp_ratio=8, qkv_bias=True, sr_ratio=4,
                  norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6)).eval()
    filler.fill_module(m, BB + "block2.1.")
    out["block2"] = m(tensor("block2", (2, 256, 128)), 16, 16).detach().numpy()
    # ---- convnext_Block (cod.py:1082-1117), layer scale 1.0 overwritten by the filler
    m = ref.convnext_Block(dim=128, drop_path=0.0, layer_scale_init_value=1.0).eval()
    filler.fill_module(m, BB + "prompt_encoder.encoder2.stages.0.1.")
    out["convnext128"] = m(tensor("convnext128", (2, 128, 16, 16))).detach().numpy()
    # ---- LayerNorm channels_first (cod.py:1044-1049)
    m = ref.LayerNorm(128, eps=1e-6, data_format="channels_first").eval()
    filler.fill_module(m, BB + "prompt_encoder.encoder2.downsample_layers.1.0.")
    out["ln_cf"] = m(tensor("ln_cf", (2, 128, 8, 8))).detach().numpy()
    # ---- fft high-pass (cod.py:1256-1271); odd and even `line`
    pe = ref.prompt_encoder(24, [64, 128, 320, 512], [3, 4, 6, 3], True).eval()
    filler.fill_module(pe, BB + "prompt_encoder.")
    for s in (64, 96):
        out[f"fft{s}"] = pe.fft(tensor(f"fft{s}", (1, 3, s, s)), 0.3).numpy()
    # ---- ShapePropWeightRegressor + nearest resize (cod.py:1295-1296, 1051-1060)
    xhp = pe.fft(tensor("diffuser_img", (2, 3, 64, 64)), 0.3)
    W = pe.propagation_weight_regressor(F.interpolate(xhp, size=[12, 12]))
    for k, v in digest(W, 8192).items():  # 1.35 MB whole; the full tensor is re-derived in tests
        out[f"regressor.{k}"] = v
    # ---- depth branch (cod.py:1297-1298)
    depth = utensor("diffuser_depth", (2, 1, 64, 64))
    e1 = F.interpolate(pe.encoder1(depth), size=(12, 12), mode="bilinear")
    out["depth_embed12"] = e1.detach().numpy()
    # ---- MessagePassing (cod.py:1180-1208), full output at img_size 64
    pe.message_passing.img_size = 64
    out["message_passing"] = pe.message_passing(e1, W).detach().numpy()
    # ---- ShapePropDecoder + prompt injection (cod.py:1210-1226, 1471-1472) for a stage-2 shape
    m = ref.ShapePropDecoder(128, 24).eval()
    filler.fill_module(m, BB + "prompt_decoder.1.decoder.0.")
    emb = tensor("spd_emb", (2, 24, 32, 32))
    p = m(emb)
    tok = tensor("spd_tok", (2, 256, 128))
    out["prompt_inject"] = (tok + F.interpolate(p, size=(16, 16), mode="bilinear").flatten(2).permute(0, 2, 1)).detach().numpy()
    # ---- CAB, SAM (cod.py:436-451, 454-506)
    m = ref.CAB(64, 3, 4, bias=False, act=torch.nn.PReLU()).eval()
    filler.fill_module(m, "hitnet.decoder_level1.0.")
    out["cab64"] = m(tensor("cab64", (2, 64, 16, 16))).detach().numpy()
    m = ref.SAM().eval()
    filler.fill_module(m, "hitnet.SAM.")
    out["sam"] = m(tensor("sam_h", (2, 32, 8, 8)), tensor("sam_l", (2, 32, 8, 8))).detach().numpy()
    # ---- losses (cod.py:76-85, 316-351)
    net = ref.cod()
    logits = tensor("loss_logits", (2, 1, 64, 64), 2.0)
    label = (utensor("loss_label", (2, 1, 64, 64)) > 0.5).float()
    out["cal_loss"] = np.float64(net.cal_loss(logits, label).item())
    out["ssim"] = np.float64(net.ssim(utensor("ssim_x", (2, 3, 32, 32)), tensor("ssim_y", (2, 3, 32, 32))).item())
    return out


def gen_model(S: int, B: int, with_grads: bool) -> dict:
    out = {}
    x, d, l = filler.synthetic_batch(B, S)
    out["input"], out["depth"], out["label"] = x.numpy(), d.numpy(), l.numpy()
    net = ref_loader.build_reference_model(S, train=False)
    filler.fill_module(net)
    with torch.no_grad():
        x_hp, P1, P2 = net.hitnet(x, d)
        out["eval.loss"] = np.float64(net(None, x, l, list(d), mode="loss")["loss"].item())
    for k, v in digest(x_hp, 1024).items():
        out[f"eval.x_hp.{k}"] = v
    out["eval.P1"] = torch.stack(P1).numpy()
    out["eval.P2"] = P2.numpy()
    if with_grads:
        net = ref_loader.build_reference_model(S, train=True)  # train mode, DropPath prob 0, BN batch stats
        filler.fill_module(net)
        x_hp, P1, P2 = net.hitnet(x, d)
        out["train.P1"] = torch.stack(P1).detach().numpy()
        out["train.P2"] = P2.detach().numpy()
        filler.fill_module(net)  # reset BN running stats touched by the probe forward above
        loss = net(None, x, l, list(d), mode="loss")["loss"]
        out["train.loss"] = np.float64(loss.item())
        loss.backward()
        names, norms = [], []
        for k, p in net.named_parameters():
            names.append(k)
            norms.append(-1.0 if p.grad is None else p.grad.double().norm().item())
        out["train.grad_names"] = np.array(names)
        out["train.grad_norms"] = np.array(norms, dtype=np.float64)          # the reference as it runs: fp32
        # The same reference in float64.  Its fp32 gradients are 0.2-0.9 % (some ConvNeXt tensors 2 %) away from these for everything
        # upstream of the trunks (oneDNN's fp32 convolutions + cancelling sums), so element-level checks use the fp64 values:
        # norms, 16 strided samples in logical (O,I,H,W) element order and the dot product with a fixed +-1 pattern (a transposed /
        # permuted / sign-flipped gradient keeps its norm but not these)
        net64 = ref_loader.build_reference_model(S, train=True)
        filler.fill_module(net64)
        net64 = net64.double()
        # Hitnet's default argument `act=nn.PReLU()` (cod.py:686) is ONE module shared by every Hitnet of the process: the fp32 model
        # above has already left its gradient on that shared parameter, and backward() accumulates
        net64.zero_grad(set_to_none=True)
        loss64 = net64(None, x.double(), l.double(), list(d.double()), mode="loss")["loss"]
        loss64.backward()
        out["train.loss64"] = np.float64(loss64.item())
        norms64, samples, projs = [], [], []
        for k, p in net64.named_parameters():
            norms64.append(-1.0 if p.grad is None else p.grad.norm().item())
            samples.append(np.zeros(GRAD_SAMPLES, np.float32) if p.grad is None else grad_samples(p.grad))
            projs.append(0.0 if p.grad is None else grad_projection(p.grad))
        out["train.grad_norms64"] = np.array(norms64, dtype=np.float64)
        out["train.grad_samples"] = np.stack(samples)
        out["train.grad_proj"] = np.array(projs, dtype=np.float64)
        bn = {k: v for k, v in net.state_dict().items() if "running_" in k}
        out["train.bn_names"] = np.array(list(bn))
        out["train.bn_values"] = np.concatenate([v.flatten().numpy() for v in bn.values()])
    return out


FP64_KEYS = ("prompt_encoder.propagation_weight_regressor.reg.", "prompt_encoder.encoder1.", "prompt_encoder.message_passing.conv.",
             "prompt_encoder.encoder2.downsample_layers.0.", "patch_embed1.proj.")


def gen_grads_fp64(S: int = 64, B: int = 2) -> dict:
    """The REAL reference in float64 (train mode, DropPath 0) on the model64 inputs: gradients of the parameters that sit upstream of
    both trunks (diffuser front end, ConvNeXt stem, first patch embed).  fp64 truth for the question "which fp32 implementation is
    the inaccurate one" (VERDICT r1 weak #4): the same reference in fp32 is stored beside it."""
    out = {}
    x, d, l = filler.synthetic_batch(B, S)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        net = ref_loader.build_reference_model(S, train=True)
        filler.fill_module(net)
        net = net.to(dt)
        net.zero_grad(set_to_none=True)        # the shared default-argument PReLU keeps gradients across model instances
        loss = net(None, x.to(dt), l.to(dt), list(d.to(dt)), mode="loss")["loss"]
        loss.backward()
        out[f"loss.{tag}"] = np.float64(loss.item())
        for k, p in net.named_parameters():
            if p.grad is not None and any(f in k for f in FP64_KEYS):
                out[f"{tag}.{k}"] = p.grad.numpy()
    return out


def gen_msda() -> dict:
    """The reference's own ms_deform_attn_core_pytorch (twig/ops/functions/ms_deform_attn_func.py:49-71), imported with an empty stub
    for the compiled MultiScaleDeformableAttention module, on the deterministic cases of oracle/ms_deform_attn_cpu.py."""
    import importlib.util
    import types
    from . import ms_deform_attn_cpu as mc
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    path = os.path.join(ref_loader.REFERENCE_ROOT, "twig", "ops", "functions", "ms_deform_attn_func.py")
    spec = importlib.util.spec_from_file_location("ref_msda_func", path)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    out = {}
    for name, (N, M, D, Lq, shapes, P) in mc.CASES.items():
        value, shp, loc, attn, grad = mc.case_inputs(name, N, M, D, Lq, shapes, P)
        value.requires_grad_(); loc.requires_grad_(); attn.requires_grad_()
        y = mod.ms_deform_attn_core_pytorch(value, shp, loc, attn)
        gv, gl, ga = torch.autograd.grad(y, (value, loc, attn), grad)
        out[f"{name}.out"] = y.detach().numpy()
        if gv.numel() <= 1 << 16:
            out[f"{name}.grad_value"] = gv.numpy()
        else:
            for k, v in digest(gv).items():
                out[f"{name}.grad_value.{k}"] = v
        out[f"{name}.grad_loc"] = gl.numpy()
        out[f"{name}.grad_attn"] = ga.numpy()
    return out


PREPROCESS_CASES = [("rgb_down", 97, 131, 3, 64), ("rgb_up", 20, 30, 3, 48), ("gray_mixed", 50, 33, 1, 40), ("rgb_same_w", 90, 64, 3, 64)]


def gen_preprocess() -> dict:
    """Real Pillow (Image.resize BILINEAR, what torchvision's Resize calls on PIL images, sod_train.py:33) on deterministic uint8
    images; ToTensor / Normalize are float32 one-liners restated in oracle/preprocess_cpu.py."""
    from PIL import Image
    out = {}
    for name, h, w, c, s in PREPROCESS_CASES:
        x = (filler.uniform("preprocess/" + name, h * w * c) * 256).astype(np.uint8).reshape(h, w, c)
        xi = x[:, :, 0] if c == 1 else x
        out[name + ".in"] = x
        out[name + ".resized"] = np.asarray(Image.fromarray(xi).resize((s, s), Image.BILINEAR))
        out[name + ".resized_flip"] = np.asarray(Image.fromarray(np.ascontiguousarray(xi[:, ::-1])).resize((s, s), Image.BILINEAR))
    return out


def main(argv):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    if not argv or "preprocess" in argv:
        np.savez_compressed(os.path.join(GOLDEN_DIR, "preprocess.npz"), **gen_preprocess())
    want = set(argv) or {"modules", "model64", "model256", "msda", "grads64_fp64"}
    if "grads64_fp64" in want:
        np.savez_compressed(os.path.join(GOLDEN_DIR, "grads64_fp64.npz"), **gen_grads_fp64())
    if "msda" in want:
        np.savez_compressed(os.path.join(GOLDEN_DIR, "msda.npz"), **gen_msda())
    if "modules" in want:
        np.savez_compressed(os.path.join(GOLDEN_DIR, "modules.npz"), **gen_modules(ref_loader.load_reference_cod()))
    if "model64" in want:
        np.savez_compressed(os.path.join(GOLDEN_DIR, "model64.npz"), **gen_model(64, 2, True))
    if "model256" in want:  # BASELINE.json configs[0]: 256x256, batch 2, forward + loss
        g = gen_model(256, 2, False)
        keep = {k: v for k, v in g.items() if k in ("eval.loss",) or k.startswith("eval.x_hp")}
        full = torch.from_numpy(g["eval.P1"][-1] + g["eval.P2"])
        for k, v in digest(full, 4096).items():
            keep[f"eval.logit.{k}"] = v
        np.savez_compressed(os.path.join(GOLDEN_DIR, "model256.npz"), **keep)
    for f in sorted(os.listdir(GOLDEN_DIR)):
        print(f, os.path.getsize(os.path.join(GOLDEN_DIR, f)))


if __name__ == "__main__":
    main(sys.argv[1:])
