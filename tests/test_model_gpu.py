"""GPU: the HIP-backed modules against the oracle (oracle/cod_cpu.py) and the committed goldens.
Tolerances: BASELINE.json north_star — <= 1e-3 fp32 on the logit map, label indices identical
(outside a reported |logit| < 1e-3 band)."""
import os

import numpy as np
import pytest
import torch

from oracle import cod_cpu, filler
from oracle.make_golden import ATTN_CASES, BB, GOLDEN_DIR, tensor

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def dgtd():
    import dgtd as m
    m._lib.load()
    return m


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN_DIR, "modules.npz"))


@pytest.mark.parametrize("case", ATTN_CASES[:4], ids=[c[0] for c in ATTN_CASES[:4]])
def test_attention_module_vs_golden(dgtd, G, case):
    name, dim, heads, sr, hw, B = case
    m = dgtd.nn.Attention(dim, num_heads=heads, qkv_bias=True, sr_ratio=sr).eval()
    stage = {64: 1, 128: 2, 320: 3, 512: 4}[dim]
    filler.fill_module(m, BB + f"block{stage}.0.attn.")
    m = m.cuda()
    with torch.no_grad():
        y = m(tensor(name, (B, hw * hw, dim)).cuda(), hw, hw)
    np.testing.assert_allclose(y.cpu().numpy(), G[name], rtol=1e-4, atol=1e-4)


def test_block_and_patch_embed_vs_golden(dgtd, G):
    m = dgtd.nn.Block(dim=128, num_heads=2, mlp_ratio=8, qkv_bias=True, sr_ratio=4).eval()
    filler.fill_module(m, BB + "block2.1.")
    with torch.no_grad():
        y = m.cuda()(tensor("block2", (2, 256, 128)).cuda(), 16, 16)
    np.testing.assert_allclose(y.cpu().numpy(), G["block2"], rtol=1e-4, atol=1e-4)
    m = dgtd.nn.OverlapPatchEmbed(img_size=64, patch_size=7, stride=4, in_chans=3, embed_dim=64).eval()
    filler.fill_module(m, BB + "patch_embed1.")
    with torch.no_grad():
        y, H, W = m.cuda()(tensor("patch_embed1", (2, 3, 64, 64)).cuda())
    assert (H, W) == (16, 16)
    np.testing.assert_allclose(y.cpu().numpy(), G["patch_embed1"], rtol=1e-4, atol=1e-4)
    m = dgtd.nn.convnext_Block(dim=128, drop_path=0.0, layer_scale_init_value=1.0).eval()
    filler.fill_module(m, BB + "prompt_encoder.encoder2.stages.0.1.")
    with torch.no_grad():
        y = m.cuda()(tensor("convnext128", (2, 128, 16, 16)).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), G["convnext128"], rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def pair64(dgtd):
    g = np.load(os.path.join(GOLDEN_DIR, "model64.npz"))
    net = dgtd.nn.cod(drop_path_rate=0.0)
    filler.fill_module(net)
    return g, net.cuda()


def test_whole_model_eval_logits_and_labels(pair64):
    g, net = pair64
    net.eval()
    x, d, l = (torch.from_numpy(g[k]).cuda() for k in ("input", "depth", "label"))
    with torch.no_grad():
        x_hp, P1, P2 = net.hitnet(x, d)
        loss = net(None, list(x), list(l), list(d), mode="loss")["loss"]
        prob, _ = net(None, x, l, d, mode="predict")
    P1 = torch.stack(P1).cpu().numpy()
    P2 = P2.cpu().numpy()
    assert np.abs(P1 - g["eval.P1"]).max() <= LOGIT_TOL
    assert np.abs(P2 - g["eval.P2"]).max() <= LOGIT_TOL
    logit, ref = P1[-1] + P2, g["eval.P1"][-1] + g["eval.P2"]
    assert np.abs(logit - ref).max() <= LOGIT_TOL
    band = np.abs(ref) < LOGIT_TOL
    assert np.array_equal((logit > 0)[~band], (ref > 0)[~band])
    assert np.array_equal((prob.cpu().numpy() > 0.5)[~band], (ref > 0)[~band])
    assert abs(loss.item() - float(g["eval.loss"])) <= LOGIT_TOL
    f = x_hp.double().flatten().cpu()
    np.testing.assert_allclose(f[::int(g["eval.x_hp.step"])].float().numpy(), g["eval.x_hp.samples"], rtol=1e-4, atol=1e-5)


def test_whole_model_train_loss_and_grads(pair64):
    g, net = pair64
    filler.fill_module(net)
    net.train()
    x, d, l = (torch.from_numpy(g[k]).cuda() for k in ("input", "depth", "label"))
    net.zero_grad(set_to_none=True)
    loss = net(None, x, l, d, mode="loss")["loss"]
    assert abs(loss.item() - float(g["train.loss"])) <= LOGIT_TOL
    loss.backward()
    want = dict(zip(g["train.grad_names"].tolist(), g["train.grad_norms"].tolist()))
    # Three diffuser parameters sit upstream of the whole ConvNeXt trunk and their gradient is a signed sum over every
    # pixel of d(fused image): the terms cancel to ~1e-3 of their magnitude, so fp32 re-association anywhere downstream
    # (MIOpen/hipBLASLt vs oneDNN) shows up ~100x amplified.  They get 2e-2; everything else 2e-3.
    cancelling = ("prompt_encoder.encoder1.", "prompt_encoder.message_passing.conv.")
    bad = []
    for k, p in net.named_parameters():
        if want[k] < 0:
            assert p.grad is None, k
            continue
        got = p.grad.double().norm().item()
        rtol = 2e-2 if any(c in k for c in cancelling) else 2e-3
        if abs(got - want[k]) > rtol * want[k] + 1e-6:
            bad.append((k, got, want[k]))
    assert not bad, bad[:10]
    bn = torch.cat([v.flatten() for k, v in net.state_dict().items() if "running_" in k]).cpu().numpy()
    np.testing.assert_allclose(bn, g["train.bn_values"], rtol=1e-3, atol=1e-4)
    filler.fill_module(net)


def test_bf16_mode_close_to_oracle(dgtd):
    """Throughput mode (bf16 MFMA kernels + bf16 GEMMs): logits stay close, labels agree away from 0."""
    g = np.load(os.path.join(GOLDEN_DIR, "model64.npz"))
    net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=torch.bfloat16)
    filler.fill_module(net)
    net = net.cuda().eval()
    x, d = (torch.from_numpy(g[k]).cuda() for k in ("input", "depth"))
    with torch.no_grad():
        _, P1, P2 = net.hitnet(x, d) if False else net._run(x, d)
    logit = (P1[-1] + P2).float().cpu().numpy()
    ref = g["eval.P1"][-1] + g["eval.P2"]
    # bf16 (8-bit mantissa) through 16 + 36 blocks; the logits span [-2.9, 2.9].  Measured: max 0.10, mean 0.032, and no
    # label flips for |logit| >= 0.05.
    # budgets = 1.5 x the measured values; a regression of half that size fails
    emax, emean = float(np.abs(logit - ref).max()), float(np.abs(logit - ref).mean())
    print(f"bf16 eval logits vs reference golden: max |diff| {emax:.4f} (budget 0.095), mean {emean:.4f} (budget 0.03)")
    assert emax < 0.095      # measured 0.063
    assert emean < 0.03      # measured 0.020
    band = np.abs(ref) < 0.1
    assert np.array_equal((logit > 0)[~band], (ref > 0)[~band])


@pytest.mark.parametrize("S,B", [(96, 2)])
def test_whole_model_vs_oracle_other_sizes(dgtd, S, B):
    """Sizes without a committed golden: run the oracle on the host and compare directly (fp32 mode, eval).
    S=96 gives N_kv = 9 (ragged attention tiles) and non-power-of-two maps everywhere."""
    ref = cod_cpu.cod(S).eval()
    filler.fill_module(ref)
    net = dgtd.nn.cod(drop_path_rate=0.0)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    x, d, l = filler.synthetic_batch(B, S, seed=S)
    with torch.no_grad():
        _, P1r, P2r = ref.hitnet(x, d)
        want_loss = ref(None, x, l, d, mode="loss")["loss"].item()
        _, P1, P2 = net.hitnet(x.cuda(), d.cuda())
        got_loss = net(None, x.cuda(), l.cuda(), d.cuda(), mode="loss")["loss"].item()
    ref_logit = (P1r[-1] + P2r).numpy()
    logit = (P1[-1] + P2).cpu().numpy()
    assert np.abs(logit - ref_logit).max() <= LOGIT_TOL
    band = np.abs(ref_logit) < LOGIT_TOL
    assert np.array_equal((logit > 0)[~band], (ref_logit > 0)[~band])
    assert abs(got_loss - want_loss) <= LOGIT_TOL


def test_batch_of_one_and_list_inputs(dgtd):
    """mmengine's pseudo_collate hands cod.forward LISTS of per-sample tensors (cod.py:120-124); batch 1 also exercises the
    per-sample DropPath plan and the BatchNorm batch statistics with a single sample."""
    net = dgtd.nn.cod(compute_dtype=torch.bfloat16).cuda().train()
    x, d, l = filler.synthetic_batch(1, 64, seed=3)
    out = net(["a.png"], [x[0].cuda()], [l[0].cuda()], [d[0].cuda()], mode="loss")
    out["loss"].backward()
    assert torch.isfinite(out["loss"])
    g = net.hitnet.backbone.prompt_encoder.propagation_weight_regressor.reg.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0


def test_tier_b_backbone_builds_and_trains(dgtd):
    """Config 3's "tier B" backbone: pvt_v2_b3 swapped into Hitnet (cod.py:1789-1795)."""
    net = dgtd.nn.cod(compute_dtype=torch.bfloat16, backbone="pvt_v2_b3").cuda().train()
    assert len(net.hitnet.backbone.block3) == 18
    x, d, l = filler.synthetic_batch(2, 64, seed=5)
    loss = net(None, x.cuda(), l.cuda(), d.cuda(), mode="loss")["loss"]
    loss.backward()
    assert torch.isfinite(loss) and net.hitnet.backbone.block3[17].attn.q.weight.grad is not None


def test_flat_adamw_matches_torch_adamw():
    """dgtd.runner.FlatAdamW over the reducer's flat buckets == torch.optim.AdamW (fused) with the same lr multipliers, after
    4 steps on a small module, including the bf16 working copies it rewrites."""
    import copy
    import dgtd
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = dgtd.nn.modules.Linear(64, 96)
            self.c = dgtd.nn.modules.Conv2d(8, 8, 3, padding=1)
            self.n = torch.nn.LayerNorm(96)
            self.hitnet_backbone = dgtd.nn.modules.Linear(96, 33)   # odd sizes: unaligned run boundaries

        def forward(self, x, img):
            return self.hitnet_backbone(self.n(self.a(x))).float().sum() + self.c(img).float().sum()

    keys = {"hitnet_backbone": 0.2}
    n1 = Net().cuda()
    n2 = copy.deepcopy(n1)
    r1 = dgtd.dist.GradReducer(n1, working_dtype=torch.bfloat16, exclude_prefixes=())
    r2 = dgtd.dist.GradReducer(n2, working_dtype=torch.bfloat16, exclude_prefixes=())
    o1 = dgtd.runner.FlatAdamW(r1, lr=1e-2, weight_decay=0.1, custom_keys=keys)
    o2 = dgtd.runner.build_optimizer(n2, lr=1e-2, weight_decay=0.1, custom_keys=keys)
    # identical gradients are written into both reducers' flat buckets (a model in the loop would amplify 1-ulp differences
    # through the bf16 working copies); torch's AdamW reads them through the masters' .grad views
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(4):
        for b1, b2 in zip(r1.buckets, r2.buckets):
            grad = torch.randn(b1["flat"].shape, device="cuda", generator=g) * 0.1
            b1["flat"].copy_(grad)
            b2["flat"].copy_(grad)
        o1.step()
        o2.step()
        r2.refresh_working()
    for (k, p1), (_, p2) in zip(n1.named_parameters(), n2.named_parameters()):
        torch.testing.assert_close(p1, p2, atol=1e-6, rtol=1e-5, msg=lambda m: f"{k}: {m}")
    for b1, b2 in zip(r1.buckets, r2.buckets):   # working copies: bf16 of (nearly) the same masters, at most one bf16 ulp apart
        for off, n in zip(b1["offsets"][:b1["k_work"]], b1["sizes"][:b1["k_work"]]):   # (this test also wrote "gradients" into the alignment padding)
            torch.testing.assert_close(b1["wflat"][off:off + n].float(), b2["wflat"][off:off + n].float(), atol=1e-3, rtol=8e-3)
    sd = o1.state_dict()
    o1.load_state_dict(sd)
    assert o1.steps == 4


def test_config1_256_eval_vs_reference_golden(dgtd):
    """BASELINE.json configs[0] (256x256, batch 2, forward + loss): the HIP model in fp32 against the REFERENCE's own outputs
    committed in tests/golden/model256.npz (loss scalar, strided samples and checksums of the logit map and of x_hp)."""
    g = np.load(os.path.join(GOLDEN_DIR, "model256.npz"))
    net = dgtd.nn.cod(drop_path_rate=0.0)
    filler.fill_module(net)
    net = net.cuda().eval()
    x, d, l = filler.synthetic_batch(2, 256)
    with torch.no_grad():
        x_hp, P1, P2 = net.hitnet(x.cuda(), d.cuda())
        loss = net(None, x.cuda(), l.cuda(), d.cuda(), mode="loss")["loss"]
    assert abs(loss.item() - float(g["eval.loss"])) <= LOGIT_TOL
    logit = (P1[-1] + P2).double().flatten().cpu()
    samples = logit[::int(g["eval.logit.step"])].float().numpy()
    assert np.abs(samples - g["eval.logit.samples"]).max() <= LOGIT_TOL
    assert abs(logit.sum().item() - float(g["eval.logit.sum"])) <= LOGIT_TOL * logit.numel() * 0.05     # mean error well inside the budget
    xs = x_hp.double().flatten().cpu()[::int(g["eval.x_hp.step"])].float().numpy()
    np.testing.assert_allclose(xs, g["eval.x_hp.samples"], rtol=1e-4, atol=1e-5)


def test_config2_full_size_properties(dgtd):
    """BASELINE.json configs[1] (512x512, batch 8): the CPU oracle needs ~47 s per step at this size, so parity is checked through
    size-independent properties of the path: (i) eval outputs are reproducible (bit-identical reruns once MIOpen is restricted to
    deterministic algorithms; its default fp32 strided-conv forward uses split-K atomics), (ii) samples are independent
    in eval mode (no cross-sample op: BatchNorm uses running statistics), i.e. row i of the batch-8 result equals the batch-1 result
    of sample i within fp32 GEMM re-tiling noise, (iii) the batch-8 loss is the mean of the per-sample cal_loss terms, (iv) one bf16
    training step at full size yields a finite loss and finite, non-zero gradients in every trunk."""
    S, B = 512, 8
    net = dgtd.nn.cod(drop_path_rate=0.0)
    filler.fill_module(net)
    net = net.cuda().eval()
    x, d, l = (t.cuda() for t in filler.synthetic_batch(B, S, seed=11))
    old_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        with torch.no_grad():
            _, P1, P2 = net.hitnet(x, d)
            _, Q1, Q2 = net.hitnet(x, d)
    finally:
        torch.backends.cudnn.deterministic = old_det
    full = P1[-1] + P2
    rerun = (full - (Q1[-1] + Q2)).abs().max().item()
    assert rerun <= 1e-5, rerun
    with torch.no_grad():
        losses = []
        for i in (0, 5):
            _, R1, R2 = net.hitnet(x[i:i + 1], d[i:i + 1])
            one = R1[-1] + R2
            assert (one - full[i:i + 1]).abs().max().item() <= 1e-4
            band = full[i:i + 1].abs() < 1e-4
            assert torch.equal((one > 0)[~band], (full[i:i + 1] > 0)[~band])
        batch_loss = net(None, x, l, d, mode="loss")["loss"].item()
        half = [net(None, x[i:i + 4], l[i:i + 4], d[i:i + 4], mode="loss")["loss"].item() for i in (0, 4)]
    # cal_loss is a mean over samples (cod.py:85) and the SSIM term a global mean of equally sized maps: the batch loss is the mean
    # of the losses of any equal split of the batch
    assert abs(batch_loss - float(np.mean(half))) <= 1e-3
    tr = dgtd.nn.cod(compute_dtype=torch.bfloat16).cuda().train()
    out = tr(None, x, l, d, mode="loss")["loss"]
    out.backward()
    assert torch.isfinite(out)
    for name in ("hitnet.backbone.block1.0.attn.q.weight", "hitnet.backbone.prompt_encoder.encoder2.stages.2.13.pwconv1.weight",
                 "hitnet.backbone.prompt_decoder.2.decoder.3.decoder.0.weight", "hitnet.decoder_level2.0.body.0.weight"):
        gr = dict(tr.named_parameters())[name].grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().sum() > 0, name


# ---------------------------------------------------------------------------------------------- element-level gradient parity
_ORACLE_CACHE = {}


def _oracle_grads(S, B, seed, dtype=torch.float32):
    """Loss and every parameter gradient of the CPU oracle in train mode (DropPath 0, BatchNorm batch statistics) in ``dtype``."""
    key = (S, B, seed, dtype)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = _oracle_grads_uncached(S, B, seed, dtype)
    return _ORACLE_CACHE[key]


def _oracle_grads_uncached(S, B, seed, dtype):
    ref = cod_cpu.cod(S).train()
    filler.fill_module(ref)
    ref = ref.to(dtype)
    x, d, l = filler.synthetic_batch(B, S, seed=seed)
    loss = ref(None, x.to(dtype), l.to(dtype), d.to(dtype), mode="loss")["loss"]
    loss.backward()
    return (x, d, l), float(loss.detach()), {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}


def test_fp32_gradients_match_reference_element_fingerprints(pair64):
    """gradient NORMS cannot see a transposed / permuted weight gradient (O,H,W,I vs O,I,H,W, q/kv halves, ky/kx).
    The golden file carries 16 strided samples and a fixed +-1 projection of every reference gradient in logical element order."""
    from oracle.make_golden import GRAD_SAMPLES, grad_projection, grad_samples
    g, net = pair64
    filler.fill_module(net)
    net.train()
    x, d, l = (torch.from_numpy(g[k]).cuda() for k in ("input", "depth", "label"))
    net.zero_grad(set_to_none=True)
    net(None, x, l, d, mode="loss")["loss"].backward()
    names = g["train.grad_names"].tolist()
    norms, samples, projs = g["train.grad_norms64"], g["train.grad_samples"], g["train.grad_proj"]
    params = dict(net.named_parameters())
    bad = []
    for i, k in enumerate(names):
        if norms[i] < 0:
            continue
        gr = params[k].grad
        # fingerprints come from the reference run in float64 (the fp32 reference itself is up to 2 % away from it); the HIP fp32 path
        # sits within ~1e-5 of the fp64 truth (test_upstream_gradients_against_fp64_reference), so 1e-3 on the scale of the tensor's
        # rms element / norm is generous for rounding and far below what a permuted / transposed / sign-flipped gradient shows (O(1))
        got_s = grad_samples(gr)
        got_p = grad_projection(gr)
        rms = norms[i] / np.sqrt(gr.numel())
        e_s, e_p = float(np.abs(got_s - samples[i]).max()), abs(got_p - float(projs[i]))
        # 1e-6 absolute: a few gradients are exactly 0 in exact arithmetic (a bias in front of a train-mode BatchNorm: norm4.bias) and
        # pure rounding noise in any floating-point run
        if e_s > 1e-3 * max(float(np.abs(samples[i]).max()), rms) + 1e-6 or e_p > 1e-3 * norms[i] + 1e-6 \
                or abs(float(gr.double().norm()) - norms[i]) > 1e-3 * norms[i] + 1e-6:
            bad.append(f"{k}: sample err {e_s:.3e} (rms element {rms:.3e}), projection {got_p:.6e} vs {float(projs[i]):.6e}, norm {norms[i]:.3e}")
    assert GRAD_SAMPLES == samples.shape[1]
    assert not bad, "\n".join(bad[:8])
    filler.fill_module(net)


def test_upstream_gradients_against_fp64_reference(dgtd):
    """the diffuser parameters got 2e-2 instead of 2e-3 in the norm test.  Settled with the REAL reference run in
    float64 (tests/golden/grads64_fp64.npz, oracle/make_golden.py): element-wise relative L2 of the HIP fp32 gradients of every
    parameter upstream of both trunks vs the fp64 truth, beside the same figure for the reference's own fp32 CPU run.  Finding
    (tools/debug_fp32_grad_error.py): the gap was MIOpen's fp32 Winograd convolutions (median error over all tensors 1.6e-2 with,
    1.7e-3 without); with them off (tests/conftest.py) the HIP path is as close to the truth as the CPU fp32 path is."""
    g = np.load(os.path.join(GOLDEN_DIR, "grads64_fp64.npz"))
    net = dgtd.nn.cod(drop_path_rate=0.0)
    filler.fill_module(net)
    net = net.cuda().train()
    x, d, l = filler.synthetic_batch(2, 64)
    loss = net(None, x.cuda(), l.cuda(), d.cuda(), mode="loss")["loss"]
    loss.backward()
    assert abs(loss.item() - float(g["loss.f64"])) < 1e-4
    params = dict(net.named_parameters())
    keys = [k[4:] for k in g.files if k.startswith("f64.")]
    assert len(keys) >= 10
    report = {}
    for k in keys:
        truth = torch.from_numpy(g["f64." + k])
        e_hip = float((params[k].grad.double().cpu() - truth).norm() / truth.norm())
        e_cpu = float((torch.from_numpy(g["f32." + k]).double() - truth).norm() / truth.norm())
        report[k] = (e_hip, e_cpu)
    print("relative L2 vs fp64 (hip fp32, reference cpu fp32):")
    for k, v in report.items():
        print(f"  {k}: {v[0]:.3e} {v[1]:.3e}")
    for k, (e_hip, e_cpu) in report.items():
        assert e_hip <= max(10.0 * e_cpu, 2e-3), (k, e_hip, e_cpu)


POOLED_BUDGET = {torch.bfloat16: 0.115, torch.float16: 0.031}      # measured 0.0768 / 0.0207      # <= 1.5 x the measured pooled error (printed by the test)


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16], ids=str)
def test_16bit_training_gradients_vs_oracle(dgtd, half):
    """the benchmarked precision had no gradient check.  The PRODUCTION configuration (16-bit working copies in
    the gradient reducer's buckets, channels_last KxK kernels, flat fp32 gradient buckets) at 64^2 against the fp32 CPU oracle,
    per tensor: cosine similarity and relative L2 of the master .grad.  Budgets: every tensor above the noise floor must point the
    same way (cos > 0.9), 90 % of the gradient mass within 15 % relative L2 (bf16) / 5 % (fp16); measured: the heavy ConvNeXt tensors sit
    at 10 % / 3 % (cos 0.995 / 0.9995)."""
    S, B = 64, 2
    (x, d, l), loss_ref, gref = _oracle_grads(S, B, seed=0)
    net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=half)
    filler.fill_module(net)
    net = net.cuda().train()
    red = dgtd.dist.GradReducer(net, working_dtype=half)
    red.zero_grad()
    loss = net(None, x.cuda(), l.cuda(), d.cuda(), mode="loss")["loss"]
    loss.backward()
    red.finish()
    assert abs(loss.item() - loss_ref) < (0.05 if half == torch.bfloat16 else 0.02), (loss.item(), loss_ref)
    total = sum(float(v.double().norm()) ** 2 for v in gref.values()) ** 0.5
    rows = []
    for k, p in net.named_parameters():
        if k not in gref:
            continue
        a, b_ = p.grad.double().cpu().flatten(), gref[k].double().flatten()
        nb = float(b_.norm())
        rel = float((a - b_).norm()) / max(nb, 1e-30)
        cos = float(torch.dot(a, b_) / (a.norm() * b_.norm() + 1e-30))
        rows.append((k, nb, rel, cos))
    mass = sum(nb * nb for _, nb, _, _ in rows)
    # one number for the whole gradient: relative L2 of all tensors pooled (dominated by the heavy ConvNeXt / PVT weights)
    pooled = (sum((rel * nb) ** 2 for _, nb, rel, _ in rows) / mass) ** 0.5
    pooled_budget = POOLED_BUDGET[half]
    print(f"{half}: pooled relative L2 of the whole gradient {pooled:.4f} (budget {pooled_budget})")
    assert pooled < pooled_budget, pooled
    budget = 0.15 if half == torch.bfloat16 else 0.05
    good = sum(nb * nb for _, nb, rel, _ in rows if rel < budget)
    worst = sorted(rows, key=lambda r: r[3])[:5]
    print(f"{half}: {len(rows)} tensors, gradient mass within {budget}: {good / mass:.4f}; worst cosines: {worst}")
    heavy = sorted([r for r in rows if r[2] >= budget], key=lambda r: -r[1])[:12]
    assert good / mass > 0.90, "fraction %.4f; heaviest tensors over budget (name, |g|, rel L2, cos):\n%s" % (good / mass, "\n".join(map(str, heavy)))
    floor = 1e-4 * total                                     # tensors whose whole gradient is below 1e-4 of the total are rounding noise
    lowest = min(cos for _, nb, _, cos in rows if nb > floor)
    print(f"{half}: lowest cosine above the noise floor {lowest:.4f} (budget > 0.9)")
    assert all(cos > 0.9 for _, nb, _, cos in rows if nb > floor), [r for r in rows if r[1] > floor and r[3] <= 0.9][:5]


def test_fp16_mode_close_to_oracle(dgtd):
    """fp16 compute (the reference's AMP dtype): eval logits against the reference golden, tighter than the bf16 budget."""
    g = np.load(os.path.join(GOLDEN_DIR, "model64.npz"))
    net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=torch.float16)
    filler.fill_module(net)
    net = net.cuda().eval()
    x, d = (torch.from_numpy(g[k]).cuda() for k in ("input", "depth"))
    with torch.no_grad():
        _, P1, P2 = net._run(x, d)
    logit = (P1[-1] + P2).float().cpu().numpy()
    ref = g["eval.P1"][-1] + g["eval.P2"]
    assert np.isfinite(logit).all()
    emax, emean = float(np.abs(logit - ref).max()), float(np.abs(logit - ref).mean())
    print(f"fp16 eval logits vs reference golden: max |diff| {emax:.4f} (budget 0.011), mean {emean:.4f} (budget 0.0023)")
    assert emax < 0.011, emax        # measured 0.0071
    assert emean < 0.0023, emean     # measured 0.0015
    band = np.abs(ref) < 0.03
    assert np.array_equal((logit > 0)[~band], (ref > 0)[~band])


def test_config3_tier_b_full_size_properties(dgtd):
    """BASELINE.json configs[2] per-GPU workload: tier-B backbone (pvt_v2_b3, cod.py:1789-1795) at 512x512, batch 8.  Same
    size-independent properties as config 2: reproducible eval, sample independence, batch loss = mean of per-sample losses, and a
    full-size bf16 training step through the gradient reducer with finite, non-zero gradients in every trunk."""
    S, B = 512, 8
    net = dgtd.nn.cod(drop_path_rate=0.0, backbone="pvt_v2_b3")
    filler.fill_module(net)
    net = net.cuda().eval()
    assert len(net.hitnet.backbone.block3) == 18
    x, d, l = (t.cuda() for t in filler.synthetic_batch(B, S, seed=13))
    old_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        with torch.no_grad():
            _, P1, P2 = net.hitnet(x, d)
            _, Q1, Q2 = net.hitnet(x, d)
    finally:
        torch.backends.cudnn.deterministic = old_det
    full = P1[-1] + P2
    assert (full - (Q1[-1] + Q2)).abs().max().item() <= 1e-5
    with torch.no_grad():
        _, R1, R2 = net.hitnet(x[3:4], d[3:4])
        one = R1[-1] + R2
        assert (one - full[3:4]).abs().max().item() <= 1e-4
        batch_loss = net(None, x, l, d, mode="loss")["loss"].item()
        per = [net(None, x[i:i + 1], l[i:i + 1], d[i:i + 1], mode="loss")["loss"].item() for i in range(B)]
    assert abs(batch_loss - float(np.mean(per))) <= 1e-3
    del net
    tr = dgtd.nn.cod(compute_dtype=torch.bfloat16, backbone="pvt_v2_b3").cuda().train()
    red = dgtd.dist.GradReducer(tr, working_dtype=torch.bfloat16)
    red.zero_grad()
    out = tr(None, x, l, d, mode="loss")["loss"]
    out.backward()
    red.finish()
    assert torch.isfinite(out)
    params = dict(tr.named_parameters())
    for name in ("hitnet.backbone.block3.17.attn.q.weight", "hitnet.backbone.block3.9.mlp.fc1.weight", "hitnet.backbone.block1.0.attn.q.weight",
                 "hitnet.backbone.prompt_encoder.encoder2.stages.2.13.pwconv1.weight", "hitnet.backbone.prompt_decoder.2.decoder.17.decoder.0.weight",
                 "hitnet.decoder_level2.0.body.0.weight"):
        gr = params[name].grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().sum() > 0, name
    assert all(not b["missing"] for b in red.buckets)      # every bucketed parameter received a gradient


def test_config4_1024_batch4_inference_vs_oracle(dgtd):
    """BASELINE.json configs[3] at its stated batch: 1024x1024, batch 4, predict mode (N = 65536 queries x N_kv = 1024 keys in stage 1:
    the multi-chunk online-softmax path of the attention kernel inside the whole model).  The CPU oracle needs ~15 s per 1024^2
    image, so it checks ONE sample of the batch (eval mode: BatchNorm uses running statistics, samples are independent); the other
    three are tied to it through the batch-1 HIP result of another sample."""
    S, B = 1024, 4
    ref = cod_cpu.cod(S).eval()
    filler.fill_module(ref)
    net = dgtd.nn.cod(drop_path_rate=0.0)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    x, d, l = filler.synthetic_batch(B, S, seed=9)
    with torch.no_grad():
        got, _ = net(None, x.cuda(), l.cuda(), d.cuda(), mode="predict")
        one, _ = net(None, x[3:4].cuda(), l[3:4].cuda(), d[3:4].cuda(), mode="predict")
        want, _ = ref(None, x[1:2], l[1:2], d[1:2], mode="predict")
    assert got.shape == (B, 1, S, S) and torch.isfinite(got).all()
    assert (one - got[3:4]).abs().max().item() <= 1e-4
    want, got1 = want.numpy(), got[1:2].cpu().numpy()
    assert np.abs(got1 - want).max() <= LOGIT_TOL            # probabilities: the 1e-3 logit budget maps to <= 2.5e-4 here
    band = np.abs(want - 0.5) < LOGIT_TOL
    assert np.array_equal((got1 > 0.5)[~band], (want > 0.5)[~band])


def test_miou_parity_hip_vs_oracle(dgtd):
    """north_star: "mIoU within 0.1 of reference on identical weights" (SURVEY 8(d)): the binary 2-class mIoU of (sigmoid > 0.5) vs
    the label, computed by ONE routine on oracle and HIP outputs over the same synthetic samples with filler weights, plus the
    reference's own mean_iou restated (identically 1.0 for the single-channel head on both sides)."""
    S, n, bs = 64, 32, 16
    ref = cod_cpu.cod(S).eval()
    filler.fill_module(ref)
    net = dgtd.nn.cod(drop_path_rate=0.0)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    M = dgtd.runner.metrics
    a, b_, r1, r2 = [], [], [], []
    for i in range(0, n, bs):
        x, d, l = filler.synthetic_batch(bs, S, seed=100 + i)
        with torch.no_grad():
            pw, _ = ref(None, x, l, d, mode="predict")
            pg, _ = net(None, x.cuda(), l.cuda(), d.cuda(), mode="predict")
        pg = pg.cpu()
        for j in range(bs):
            a.append(M.binary_miou(pw[j:j + 1], l[j:j + 1]))
            b_.append(M.binary_miou(pg[j:j + 1], l[j:j + 1]))
        r1.append(M.mean_iou_reference(pw, l))
        r2.append(M.mean_iou_reference(pg, l))
    miou_ref, miou_hip = 100 * float(np.mean(a)), 100 * float(np.mean(b_))
    print(f"binary mIoU: oracle {miou_ref:.4f}  hip {miou_hip:.4f}")
    assert abs(miou_ref - miou_hip) <= 0.1
    assert all(v == 1.0 for v in r1 + r2)
