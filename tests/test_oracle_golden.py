"""CPU: the oracle restatement (oracle/cod_cpu.py) against the golden vectors that
oracle/make_golden.py produced from the real reference (tests/golden/*.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cod_cpu, filler
from oracle.make_golden import ATTN_CASES, BB, GOLDEN_DIR, tensor, utensor

# same-machine runs are bit-exact; a different host CPU may pick other oneDNN kernels
RTOL, ATOL = 2e-5, 2e-5


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN_DIR, "modules.npz"))


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def close_digest(t, G, name, rtol=1e-4):
    f = t.detach().double().flatten()
    np.testing.assert_allclose(f[::int(G[f"{name}.step"])].float().numpy(), G[f"{name}.samples"], rtol=RTOL, atol=ATOL)
    d = {"sum": f.sum().item(), "abssum": f.abs().sum().item(), "sqsum": (f * f).sum().item()}
    for k in ("sum", "abssum", "sqsum"):
        np.testing.assert_allclose(d[k], G[f"{name}.{k}"], rtol=rtol, atol=1e-3)


@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention(G, case):
    name, dim, heads, sr, hw, B = case
    m = cod_cpu.Attention(dim, heads, sr).eval()
    stage = {64: 1, 128: 2, 320: 3, 512: 4}[dim]
    filler.fill_module(m, BB + f"block{stage}.0.attn.")
    with torch.no_grad():
        y = m(tensor(name, (B, hw * hw, dim)), hw, hw)
    if name in G:
        close(y, G[name])
    else:
        close_digest(y, G, name)


def test_patch_embed(G):
    m = cod_cpu.OverlapPatchEmbed(7, 4, 3, 64).eval()
    filler.fill_module(m, BB + "patch_embed1.")
    with torch.no_grad():
        close(m(tensor("patch_embed1", (2, 3, 64, 64)))[0], G["patch_embed1"])
    m = cod_cpu.OverlapPatchEmbed(3, 2, 64, 128).eval()
    filler.fill_module(m, BB + "patch_embed2.")
    with torch.no_grad():
        close(m(tensor("patch_embed2", (2, 64, 16, 16)))[0], G["patch_embed2"])


def test_mlp_block(G):
    m = cod_cpu.Mlp(64, 512).eval()
    filler.fill_module(m, BB + "block1.0.mlp.")
    with torch.no_grad():
        close(m(tensor("mlp1", (2, 256, 64)), 16, 16), G["mlp1"])
    m = cod_cpu.Block(128, 2, 8, 4).eval()
    filler.fill_module(m, BB + "block2.1.")
    with torch.no_grad():
        close(m(tensor("block2", (2, 256, 128)), 16, 16), G["block2"])


def test_convnext_and_ln(G):
    m = cod_cpu.ConvNeXtBlock(128).eval()
    filler.fill_module(m, BB + "prompt_encoder.encoder2.stages.0.1.")
    with torch.no_grad():
        close(m(tensor("convnext128", (2, 128, 16, 16))), G["convnext128"])
    m = cod_cpu.LayerNorm2(128, channels_first=True).eval()
    filler.fill_module(m, BB + "prompt_encoder.encoder2.downsample_layers.1.0.")
    with torch.no_grad():
        close(m(tensor("ln_cf", (2, 128, 8, 8))), G["ln_cf"])


def test_diffuser_front_end(G):
    for s in (64, 96):
        close(cod_cpu.fft_highpass(tensor(f"fft{s}", (1, 3, s, s))), G[f"fft{s}"])
    pe = cod_cpu.PromptEncoder(64).eval()
    filler.fill_module(pe, BB + "prompt_encoder.")
    with torch.no_grad():
        xhp = cod_cpu.fft_highpass(tensor("diffuser_img", (2, 3, 64, 64)))
        W = pe.propagation_weight_regressor(F.interpolate(xhp, size=[12, 12]))
        close_digest(W, G, "regressor")
        e1 = F.interpolate(pe.encoder1(utensor("diffuser_depth", (2, 1, 64, 64))), size=(12, 12), mode="bilinear")
        close(e1, G["depth_embed12"])
        close(pe.message_passing(e1, W), G["message_passing"])


def test_prompt_inject(G):
    m = cod_cpu.ShapePropDecoder(128).eval()
    filler.fill_module(m, BB + "prompt_decoder.1.decoder.0.")
    with torch.no_grad():
        p = m(tensor("spd_emb", (2, 24, 32, 32)))
        y = tensor("spd_tok", (2, 256, 128)) + F.interpolate(p, size=(16, 16), mode="bilinear").flatten(2).transpose(1, 2)
    close(y, G["prompt_inject"])


def test_cab_sam_losses(G):
    m = cod_cpu.CAB(64, torch.nn.PReLU()).eval()
    filler.fill_module(m, "hitnet.decoder_level1.0.")
    with torch.no_grad():
        close(m(tensor("cab64", (2, 64, 16, 16))), G["cab64"])
    m = cod_cpu.SAM().eval()
    filler.fill_module(m, "hitnet.SAM.")
    with torch.no_grad():
        close(m(tensor("sam_h", (2, 32, 8, 8)), tensor("sam_l", (2, 32, 8, 8))), G["sam"])
    logits = tensor("loss_logits", (2, 1, 64, 64), 2.0)
    label = (utensor("loss_label", (2, 1, 64, 64)) > 0.5).float()
    close(cod_cpu.cal_loss(logits, label), G["cal_loss"])
    close(cod_cpu.ssim_value(utensor("ssim_x", (2, 3, 32, 32)), tensor("ssim_y", (2, 3, 32, 32))), G["ssim"])


@pytest.fixture(scope="module")
def model64():
    g = np.load(os.path.join(GOLDEN_DIR, "model64.npz"))
    net = cod_cpu.cod(64)
    filler.fill_module(net)
    return g, net


def test_state_dict_contract(model64):
    g, net = model64
    sd = net.state_dict()
    assert len(sd) == 879
    assert sum(p.numel() for p in net.parameters()) == 114_413_839
    assert tuple(sd["hitnet.backbone.block3.5.attn.kv.weight"].shape) == (640, 320)
    assert tuple(sd["hitnet.backbone.prompt_encoder.encoder2.stages.2.26.pwconv1.weight"].shape) == (2048, 512)
    assert tuple(sd["hitnet.backbone.prompt_decoder.2.decoder.5.decoder.4.weight"].shape) == (320, 24, 3, 3)
    assert sd["hitnet.decoder_level2.0.body.1.weight"].data_ptr() == sd["hitnet.decoder_level4.1.body.1.weight"].data_ptr()
    assert set(g["train.grad_names"].tolist()) == {k for k, _ in net.named_parameters()}


def test_whole_model_eval(model64):
    g, net = model64
    net.eval()
    x, d, l = (torch.from_numpy(g[k]) for k in ("input", "depth", "label"))
    fx, fd, fl = filler.synthetic_batch(2, 64)
    close(fx, g["input"]); close(fd, g["depth"], atol=1e-6); close(fl, g["label"])
    with torch.no_grad():
        x_hp, P1, P2 = net.hitnet(x, d)
        loss = net(None, list(x), list(l), list(d), mode="loss")["loss"]
        prob, _ = net(None, x, l, d, mode="predict")
    close(torch.stack(P1), g["eval.P1"])
    close(P2, g["eval.P2"])
    close_digest(x_hp, g, "eval.x_hp")
    close(loss, g["eval.loss"])
    logit = P1[-1] + P2
    assert torch.equal(prob > 0.5, logit > 0)  # "argmax label" == (logit > 0), SURVEY §0
    ref_logit = torch.from_numpy(g["eval.P1"][-1] + g["eval.P2"])
    band = ref_logit.abs() < 1e-4
    assert torch.equal((logit > 0)[~band], (ref_logit > 0)[~band])


def test_whole_model_train_grads(model64):
    g, net = model64
    filler.fill_module(net)
    net.train()
    x, d, l = (torch.from_numpy(g[k]) for k in ("input", "depth", "label"))
    net.zero_grad(set_to_none=True)
    loss = net(None, x, l, d, mode="loss")["loss"]
    close(loss, g["train.loss"])
    loss.backward()
    want = dict(zip(g["train.grad_names"].tolist(), g["train.grad_norms"].tolist()))
    unused = sorted(k for k, v in want.items() if v < 0)
    assert unused == sorted(["hitnet.backbone.prompt_encoder.adaptor.weight", "hitnet.backbone.prompt_encoder.adaptor.bias",
                             "hitnet.ca.fc1.weight", "hitnet.ca.fc2.weight", "hitnet.sa.conv1.weight"])
    for k, p in net.named_parameters():
        if want[k] < 0:
            assert p.grad is None, k
        else:
            np.testing.assert_allclose(p.grad.double().norm().item(), want[k], rtol=2e-4, atol=1e-7, err_msg=k)
    bn = {k: v for k, v in net.state_dict().items() if "running_" in k}
    assert list(bn) == g["train.bn_names"].tolist()
    close(torch.cat([v.flatten() for v in bn.values()]), g["train.bn_values"])
    filler.fill_module(net)


def test_config1_loss_256():
    """BASELINE.json configs[0]: sod.yml model, 256x256, batch 2, forward + loss on CPU."""
    g = np.load(os.path.join(GOLDEN_DIR, "model256.npz"))
    net = cod_cpu.cod(256).eval()
    filler.fill_module(net)
    x, d, l = filler.synthetic_batch(2, 256)
    with torch.no_grad():
        x_hp, P1, P2 = net.hitnet(x, d)
        loss = cod_cpu.total_loss(x_hp, P1, P2, x, l)
    close(loss, g["eval.loss"], rtol=1e-5)
    close_digest(P1[-1] + P2, g, "eval.logit")
    close_digest(x_hp, g, "eval.x_hp")
