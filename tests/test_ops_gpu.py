"""GPU: every HIP op through the C ABI against a plain PyTorch fp32 reference of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["python-bindings", "cpp-bindings"])
def dgtd(request):
    """Every op test runs through both host binding layers over the same C ABI: the Python autograd.Functions and the
    C++ torch bindings (libdgtd_torch.so)."""
    import dgtd as m
    m._lib.load()
    old = m.ops._native.ENABLED
    m.ops._native.ENABLED = request.param == "cpp-bindings"
    if m.ops._native.ENABLED:
        assert m.ops._native.ops() is not None, "libdgtd_torch.so missing: run python __graft_entry__.py"
    yield m
    m.ops._native.ENABLED = old


# fp32 = exact kernels (parity mode); bf16 = throughput mode; fp16 = the reference's own AMP dtype (config/sod.yml:57), same kernels on the
# f16 MFMA / conversions.  Tolerances below treat both 16-bit types alike (fp16 has 3 more mantissa bits than bf16).
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
HALVES = [torch.bfloat16, torch.float16]


def _rand(*shape, seed=0, dtype=torch.float32, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)


# ---------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("C", [64, 128, 320, 512, 256, 1024])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_layernorm_fwd_bwd(dgtd, C, dtype):
    rows = 777  # ragged vs rows-per-block
    x = _rand(rows, C, seed=C, dtype=dtype)
    w = (1 + 0.1 * _rand(C, seed=1)).requires_grad_()
    b = (0.1 * _rand(C, seed=2)).requires_grad_()
    dy = _rand(rows, C, seed=3, dtype=dtype)
    xr = x.float().requires_grad_()
    ref = F.layer_norm(xr, (C,), w, b, 1e-6)
    gx, gw, gb = torch.autograd.grad(ref, (xr, w, b), dy.float())
    xs = x.clone().requires_grad_()
    y = dgtd.ops.layer_norm(xs, w, b, 1e-6)
    hx, hw, hb = torch.autograd.grad(y, (xs, w, b), dy)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert y.dtype == dtype
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(hx.float(), gx, atol=tol, rtol=tol)
    gtol = 2e-4 if dtype == torch.float32 else 5e-2
    torch.testing.assert_close(hw, gw, atol=gtol * math.sqrt(rows), rtol=gtol)
    torch.testing.assert_close(hb, gb, atol=gtol * math.sqrt(rows), rtol=gtol)


# ---------------------------------------------------------------------------------------------- attention
def _attn_ref(q, kv, heads, scale):
    B, N, C = q.shape
    d = C // heads
    qh = q.view(B, N, heads, d).transpose(1, 2)
    kvh = kv.view(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    p = torch.softmax((qh @ kvh[0].transpose(-2, -1)) * scale, dim=-1)
    return (p @ kvh[1]).transpose(1, 2).reshape(B, N, C)


# (B, N, Nkv, heads): the four PVT stages at S=256 and S=512, S=384's N_kv=144, ragged/tiny cases,
# and N_kv=1024 (S=1024) which exercises the multi-chunk online softmax and the key-slice loop of the backward.
ATTN_SHAPES = [(2, 4096, 64, 1), (2, 1024, 64, 2), (2, 256, 64, 5), (2, 64, 64, 8),
               (1, 16384, 256, 1), (1, 4096, 256, 2), (2, 1024, 256, 5), (2, 256, 256, 8),
               (1, 2304, 144, 2), (2, 4, 4, 8), (1, 1, 1, 8), (3, 100, 37, 2), (1, 2048, 1024, 2)]


@pytest.mark.parametrize("shape", ATTN_SHAPES, ids=[str(s) for s in ATTN_SHAPES])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_sra_attention_fwd_bwd(dgtd, shape, dtype):
    B, N, Nkv, heads = shape
    C = heads * 64
    q = _rand(B, N, C, seed=1, dtype=dtype)
    kv = _rand(B, Nkv, 2 * C, seed=2, dtype=dtype)
    do = _rand(B, N, C, seed=3, dtype=dtype)
    scale = 64 ** -0.5
    qr, kvr = q.float().requires_grad_(), kv.float().requires_grad_()
    ref = _attn_ref(qr, kvr, heads, scale)
    gq, gkv = torch.autograd.grad(ref, (qr, kvr), do.float())
    qs, kvs = q.clone().requires_grad_(), kv.clone().requires_grad_()
    out = dgtd.ops.sra_attention(qs, kvs, heads, scale)
    hq, hkv = torch.autograd.grad(out, (qs, kvs), do)
    if dtype == torch.float32:
        # exact-fp32 MFMA: the 1e-3 logit budget of BASELINE.json needs op error well below it
        torch.testing.assert_close(out, ref, atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(hq, gq, atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(hkv, gkv, atol=2e-4 * math.sqrt(N / 64), rtol=1e-3)
    else:
        torch.testing.assert_close(out.float(), ref, atol=3e-2, rtol=3e-2)
        torch.testing.assert_close(hq.float(), gq, atol=5e-2, rtol=5e-2)
        rel = (hkv.float() - gkv).norm() / gkv.norm()
        assert rel < 2e-2, rel


def test_attention_spiked_scores_online_softmax(dgtd):
    """Force the running-max rescale across K/V chunks (guide §5.4 rule 26): one key in the LAST chunk
    dominates one query row."""
    B, N, Nkv, heads = 1, 64, 1024, 1
    q = _rand(B, N, 64, seed=5)
    kv = _rand(B, Nkv, 128, seed=6)
    kv[0, 1000, :64] = q[0, 7] * 4.0
    ref = _attn_ref(q, kv, heads, 0.125)
    out = dgtd.ops.sra_attention(q, kv, heads, 0.125)
    torch.testing.assert_close(out, ref, atol=2e-5, rtol=2e-5)
    for half in HALVES:
        out16 = dgtd.ops.sra_attention(q.to(half), kv.to(half), heads, 0.125)
        ref16 = _attn_ref(q.to(half).float(), kv.to(half).float(), heads, 0.125)
        torch.testing.assert_close(out16.float(), ref16, atol=3e-2, rtol=3e-2)


# ---------------------------------------------------------------------------------------------- texture diffuser
@pytest.mark.parametrize("S", [64, 96, 512])
def test_diffuser_front_end_vs_oracle(dgtd, S):
    """diffuser_state + diffuse_tail against the oracle's PromptEncoder front end (cod.py:1295-1302), values and
    parameter gradients.  S=96 exercises a non-integer nearest/bilinear scale (96/12 = 8 exact, 64/12 is not)."""
    from oracle import cod_cpu, filler
    B = 2
    pe = cod_cpu.PromptEncoder(S)
    pe.encoder2 = torch.nn.Identity()  # only the front end is under test
    filler.fill_module(pe, "hitnet.backbone.prompt_encoder.")
    g = torch.Generator().manual_seed(S)
    image = torch.randn(B, 3, S, S, generator=g)
    depth = torch.rand(B, 1, S, S, generator=g)
    gout = torch.randn(B, 3, S, S, generator=g)
    x_hp, fused = pe(image, depth)
    names = ["propagation_weight_regressor.reg.weight", "propagation_weight_regressor.reg.bias", "encoder1.weight",
             "encoder1.bias", "message_passing.conv.weight", "message_passing.conv.bias"]
    params = dict(pe.named_parameters())
    want = torch.autograd.grad(fused, [params[n] for n in names], gout)

    dev = [params[n].detach().cuda().requires_grad_() for n in names]
    x4 = dgtd.ops.diffuser_state(x_hp.cuda(), depth.cuda(), dev[0], dev[1], dev[2], dev[3])
    out = dgtd.ops.diffuse_tail(x4, dev[4], dev[5], image.cuda())
    torch.testing.assert_close(out.cpu(), fused.detach(), atol=2e-5, rtol=2e-5)
    got = torch.autograd.grad(out, dev, gout.cuda())
    for n, a, b in zip(names, got, want):
        torch.testing.assert_close(a.cpu(), b, atol=2e-4 * max(1.0, b.abs().max().item()), rtol=2e-3, msg=lambda m, n=n: f"{n}: {m}")


# ---------------------------------------------------------------------------------------------- fused epilogues
@pytest.mark.parametrize("C", [64, 320, 1024, 2048])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("with_s,with_g", [(True, True), (True, False), (False, True)])
def test_scale_residual_fwd_bwd(dgtd, C, dtype, with_s, with_g):
    B, N = 3, 173
    x, y, g = (_rand(B, N, C, seed=i, dtype=dtype) for i in (1, 2, 3))
    s = torch.tensor([0.0, 1.25, 1.25], device="cuda") if with_s else None
    gamma = (1 + 0.1 * _rand(C, seed=4)).requires_grad_() if with_g else None
    xr, yr = x.float().requires_grad_(), y.float().requires_grad_()
    t = yr * (gamma if with_g else 1.0)
    if with_s:
        t = t * s.view(B, 1, 1)
    ref = xr + t
    wants = torch.autograd.grad(ref, [xr, yr] + ([gamma] if with_g else []), g.float())
    xs, ys = x.clone().requires_grad_(), y.clone().requires_grad_()
    out = dgtd.ops.scale_residual(xs, ys, s, gamma)
    gots = torch.autograd.grad(out, [xs, ys] + ([gamma] if with_g else []), g)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(out.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(gots[0].float(), wants[0], atol=tol, rtol=tol)
    torch.testing.assert_close(gots[1].float(), wants[1], atol=tol, rtol=tol)
    if with_g:
        torch.testing.assert_close(gots[2], wants[2], atol=(2e-4 if dtype == torch.float32 else 0.3), rtol=(1e-4 if dtype == torch.float32 else 3e-2))


@pytest.mark.parametrize("rows,K,N", [(1000, 64, 512), (77, 320, 1280), (4, 2048, 512)])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_linear_with_colsum_bias_grad(dgtd, rows, K, N, dtype):
    x = _rand(2, rows, K, seed=1, dtype=dtype)
    w = _rand(N, K, seed=2, dtype=dtype, scale=K ** -0.5)
    b = _rand(N, seed=3, dtype=dtype, scale=0.1)
    dy = _rand(2, rows, N, seed=4, dtype=dtype)
    xr, wr, br = (t.float().requires_grad_() for t in (x, w, b))
    ref = F.linear(xr, wr, br)
    wants = torch.autograd.grad(ref, (xr, wr, br), dy.float())
    xs, ws, bs = (t.clone().requires_grad_() for t in (x, w, b))
    out = dgtd.ops.linear(xs, ws, bs)
    gots = torch.autograd.grad(out, (xs, ws, bs), dy)
    if dtype == torch.float32:
        tols = [(2e-4, 2e-4)] * 4  # library fp32 GEMM vs fp32 GEMM: different blocking only
    else:
        tols = [(5e-2, 5e-2)] * 4
    torch.testing.assert_close(out.float(), ref, atol=tols[0][0], rtol=tols[0][1])
    torch.testing.assert_close(gots[0].float(), wants[0], atol=tols[1][0], rtol=tols[1][1])
    assert (gots[1].float() - wants[1]).norm() / wants[1].norm() < (1e-4 if dtype == torch.float32 else 2e-2)
    assert (gots[2].float() - wants[2]).norm() / wants[2].norm() < (1e-5 if dtype == torch.float32 else 2e-2)
    cs = dgtd.ops.colsum(dy.reshape(-1, N))
    torch.testing.assert_close(cs, dy.float().reshape(-1, N).sum(0), atol=1e-3 if dtype == torch.float32 else 1e-2, rtol=1e-4)


# ---------------------------------------------------------------------------------------------- fused structure loss
@pytest.mark.parametrize("S,hs,B", [(64, 8, 2), (96, 12, 3), (512, 64, 2), (64, 16, 2), (160, 20, 1)])     # x8 (tiled backward), x4 (generic gather)
def test_seg_loss_vs_oracle(dgtd, S, hs, B):
    from oracle import cod_cpu
    g = torch.Generator().manual_seed(S)
    maps = [(torch.randn(B, 1, hs, hs, generator=g) * 2).requires_grad_() for _ in range(5)]
    lo = torch.rand(B, 1, max(S // 16, 2), max(S // 16, 2), generator=g)
    label = (F.interpolate(lo, size=(S, S), mode="bilinear") > 0.5).float()
    mix = (0.0, 0.2, 0.4, 0.6, 1.0)
    up = [F.interpolate(m, scale_factor=S // hs, mode="bilinear", align_corners=False) for m in maps]
    want = sum(w * cod_cpu.cal_loss(u, label) for w, u in zip(mix, up))
    gw = torch.autograd.grad(want, maps)
    dev = [m.detach().cuda().requires_grad_() for m in maps]
    got = dgtd.ops.seg_loss(dev, label.cuda(), mix)
    gg = torch.autograd.grad(got * 1.5, dev)   # non-unit upstream gradient
    assert abs(got.item() - want.item()) < 2e-5 * max(1.0, abs(want.item()))
    for a, b, w in zip(gg, gw, mix):
        torch.testing.assert_close(a.cpu() / 1.5, b, atol=2e-7, rtol=2e-4)


# ---------------------------------------------------------------------------------------------- Hitnet CAB glue
@pytest.mark.parametrize("shape", [(2, 32, 16, 16), (3, 96, 20, 12), (2, 64, 128, 128)], ids=str)
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("channels_last", [True, False])
def test_prelu_shared_slope(dgtd, shape, dtype, channels_last):
    x = _rand(*shape, seed=5, dtype=dtype)
    g = _rand(*shape, seed=6, dtype=dtype)
    if channels_last:
        x, g = x.contiguous(memory_format=torch.channels_last), g.contiguous(memory_format=torch.channels_last)
    a = torch.full((1,), 0.25, device="cuda").requires_grad_()
    xr = x.float().requires_grad_()
    ref = F.prelu(xr, a)
    gx, ga = torch.autograd.grad(ref, (xr, a), g.float())
    xs = x.clone().requires_grad_()
    y = dgtd.ops.prelu(xs, a)
    hx, ha = torch.autograd.grad(y, (xs, a), g)
    assert y.dtype == dtype and y.stride() == x.stride()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(hx.float(), gx, atol=tol, rtol=tol)
    torch.testing.assert_close(ha, ga, atol=1e-3 * math.sqrt(x.numel()) * (1 if dtype == torch.float32 else 30), rtol=1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 16, 16), (3, 64, 12, 20), (2, 96, 64, 64), (1, 64, 128, 128)])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_ca_gate_matches_calayer_plus_residual(dgtd, B, C, H, W, dtype):
    """CALayer (cod.py:415-431) + the CAB residual (cod.py:451) against the plain torch composition in fp32."""
    R = C // 4
    res = _rand(B, C, H, W, seed=1, dtype=dtype).contiguous(memory_format=torch.channels_last)
    x = _rand(B, C, H, W, seed=2, dtype=dtype).contiguous(memory_format=torch.channels_last)
    g = _rand(B, C, H, W, seed=3, dtype=dtype).contiguous(memory_format=torch.channels_last)
    w1 = (_rand(R, C, 1, 1, seed=4) / math.sqrt(C)).requires_grad_()
    w2 = (_rand(C, R, 1, 1, seed=5) / math.sqrt(R)).requires_grad_()
    rr, xr = res.float().requires_grad_(), x.float().requires_grad_()
    gate = torch.sigmoid(F.conv2d(F.relu(F.conv2d(rr.mean((2, 3), keepdim=True), w1)), w2))
    ref = rr * gate + xr
    gres, gxx, gw1, gw2 = torch.autograd.grad(ref, (rr, xr, w1, w2), g.float())
    rs, xs = res.clone().requires_grad_(), x.clone().requires_grad_()
    out = dgtd.ops.ca_gate(rs, xs, w1, w2)
    hres, hx, hw1, hw2 = torch.autograd.grad(out, (rs, xs, w1, w2), g)
    assert out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(out.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(hres.float(), gres, atol=tol, rtol=tol)
    torch.testing.assert_close(hx.float(), gxx, atol=tol, rtol=tol)
    wtol = 1e-4 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(hw1, gw1, atol=wtol * gw1.abs().max().item() + 1e-6, rtol=wtol)
    torch.testing.assert_close(hw2, gw2, atol=wtol * gw2.abs().max().item() + 1e-6, rtol=wtol)


@pytest.mark.parametrize("dtype", HALVES, ids=str)
def test_transpose_batched_mixed_shapes(dgtd, dtype):
    """dgtd_transpose_batched (the once-per-step transposed weight copies of csrc/gemm.hip): 70 matrices in one call (two launches of
    <= 64), full and partial 64x64 tiles on the 16-byte path, shapes that are not multiples of 8 and a misaligned view on the 2-byte path."""
    import ctypes as C
    L = dgtd._lib
    shapes = [(320, 1280), (1280, 320), (64, 64), (8, 8), (72, 40), (512, 2048), (30, 50), (7, 64), (64, 9)] + [(128, 64 + 8 * i) for i in range(61)]
    srcs = [_rand(r, c, seed=10 + i).to(dtype) for i, (r, c) in enumerate(shapes)]
    odd = _rand(16 * 24 + 1, seed=99).to(dtype)[1:].view(16, 24)           # 2 bytes off a 16-byte boundary
    srcs.append(odd)
    dsts = [torch.full((t.shape[1], t.shape[0]), float("nan"), device="cuda", dtype=dtype) for t in srcs]
    n = len(srcs)
    P, I = C.c_void_p * n, C.c_int * n
    L.call("dgtd_transpose_batched", P(*[t.data_ptr() for t in srcs]), P(*[t.data_ptr() for t in dsts]), I(*[t.shape[0] for t in srcs]),
           I(*[t.shape[1] for t in srcs]), n, L.dtype_code(srcs[0]), L.stream_ptr())
    for a, b in zip(srcs, dsts):
        assert torch.equal(b, a.t().contiguous()), tuple(a.shape)


@pytest.mark.parametrize("B,C,H,W", [(8, 96, 64, 64), (3, 64, 12, 20), (40, 32, 16, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=str)
def test_ca_gate_bwd_rows_sum_to_the_batched_gradients(dgtd, B, C, H, W, dtype):
    """dgtd_ca_gate_bwd_rows leaves one { dw1 | dw2 } row per sample (the deferred flush sums them with the rows of the module's other
    calls): same dres bit for bit, and the rows add up to what dgtd_ca_gate_bwd sums inside its launch (also beyond its 32-sample cap)."""
    L = dgtd._lib
    R = C // 4
    res = _rand(B, C, H, W, seed=1, dtype=dtype).contiguous(memory_format=torch.channels_last)
    x = _rand(B, C, H, W, seed=2, dtype=dtype).contiguous(memory_format=torch.channels_last)
    g = _rand(B, C, H, W, seed=3, dtype=dtype).contiguous(memory_format=torch.channels_last)
    w1 = (_rand(R, C, seed=4) / math.sqrt(C)).contiguous()
    w2 = (_rand(C, R, seed=5) / math.sqrt(R)).contiguous()
    stats = torch.empty(2 * B * C + B * R + 64 * B * C, dtype=torch.float32, device="cuda")
    out = torch.empty_like(res)
    L.call("dgtd_ca_gate_fwd", L.ptr(res), L.ptr(x), L.ptr(w1), L.ptr(w2), L.ptr(out), L.ptr(stats), B, H * W, C, R, L.dtype_code(res), L.stream_ptr())
    small = torch.empty(2 * R * C + B * C + 64 * B * C + B * 2 * R * C, dtype=torch.float32, device="cuda")
    d_a, d_b = torch.empty_like(res), torch.empty_like(res)
    L.call("dgtd_ca_gate_bwd", L.ptr(g), L.ptr(res), L.ptr(w1), L.ptr(w2), L.ptr(stats), L.ptr(d_a), small[:R * C].data_ptr(), small[R * C:].data_ptr(),
           small[2 * R * C:].data_ptr(), B, H * W, C, R, L.dtype_code(res), L.stream_ptr())
    ref = small[:2 * R * C].clone()
    rows = torch.full((B, 2 * R * C), float("nan"), dtype=torch.float32, device="cuda")
    L.call("dgtd_ca_gate_bwd_rows", L.ptr(g), L.ptr(res), L.ptr(w1), L.ptr(w2), L.ptr(stats), L.ptr(d_b), L.ptr(rows), small[2 * R * C:].data_ptr(),
           B, H * W, C, R, L.dtype_code(res), L.stream_ptr())
    assert torch.equal(d_a, d_b)
    torch.testing.assert_close(rows.sum(0), ref, atol=2e-5 * ref.abs().max().item() + 1e-7, rtol=1e-4)


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 16, 16), (8, 32, 64, 64), (3, 64, 12, 20), (1, 32, 7, 9)])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("layer", ["native", "ctypes"])
def test_sam_matches_torch_composition(dgtd, B, C, H, W, dtype, layer):
    """SAM (cod.py:454-506) in two launches each way against the plain torch composition in fp32: output, both input gradients and
    the four weight gradients (the weights gate BOTH inputs, so each gradient sums two paths)."""
    from dgtd.ops import hitnet as H_
    R = max(C // 16, 1)
    xh = _rand(B, C, H, W, seed=1, dtype=dtype).contiguous(memory_format=torch.channels_last)
    xl = _rand(B, C, H, W, seed=2, dtype=dtype).contiguous(memory_format=torch.channels_last)
    g = _rand(B, C, H, W, seed=3, dtype=dtype).contiguous(memory_format=torch.channels_last)
    ws = [(_rand(R, C, seed=4) / math.sqrt(C)).requires_grad_(), (_rand(C, R, seed=5) / math.sqrt(R)).requires_grad_(),
          (_rand(R, C, seed=6) / math.sqrt(C)).requires_grad_(), (_rand(1, R, seed=7) / math.sqrt(R)).requires_grad_()]

    def gate(x):
        y = x.mean((2, 3))
        a = torch.sigmoid(F.linear(F.relu(F.linear(y, ws[0])), ws[1]))
        b = torch.sigmoid(F.linear(F.relu(F.linear(y, ws[2])), ws[3]))
        return x * a[:, :, None, None] * b[:, :, None, None]

    hr, lr = xh.float().requires_grad_(), xl.float().requires_grad_()
    ref = gate(hr) + gate(lr)
    gref = torch.autograd.grad(ref, (hr, lr, *ws), g.float())
    hs, ls = xh.clone().requires_grad_(), xl.clone().requires_grad_()
    out = dgtd.ops.sam(hs, ls, *ws) if layer == "native" else H_._SamFn.apply(hs, ls, *ws)
    got = torch.autograd.grad(out, (hs, ls, *ws), g)
    assert out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(out.float(), ref, atol=tol, rtol=tol)
    for a, b in zip(got[:2], gref[:2]):
        torch.testing.assert_close(a.float(), b, atol=tol, rtol=tol)
    wtol = 2e-4 if dtype == torch.float32 else 3e-2
    for a, b in zip(got[2:], gref[2:]):
        assert a.shape == b.shape and a.dtype == b.dtype
        torch.testing.assert_close(a, b, atol=wtol * b.abs().max().item() + 1e-6, rtol=wtol)


@pytest.mark.parametrize("B,C,H,W", [(8, 32, 16, 16), (8, 32, 64, 64), (2, 32, 128, 128), (3, 64, 12, 20), (1, 8, 5, 7), (2, 128, 9, 9)])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("layer", ["native", "ctypes"])
def test_batch_norm_matches_torch(dgtd, B, C, H, W, dtype, layer):
    """nn.BatchNorm2d of BasicConv2d (cod.py:359, :366), training mode: output, saved running statistics, num_batches_tracked and the
    three gradients against torch's own BatchNorm2d on the fp32 copy of the same input; then the eval-mode map of the moved statistics."""
    from dgtd.ops import hitnet as H_
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * _rand(C, seed=11)); bn.bias.copy_(0.2 * _rand(C, seed=12))
        bn.running_mean.copy_(0.1 * _rand(C, seed=13)); bn.running_var.copy_(1 + 0.2 * _rand(C, seed=14).abs())
    import copy
    ref_bn = copy.deepcopy(bn)
    x = (1.5 * _rand(B, C, H, W, seed=1) + 0.7).to(dtype).contiguous(memory_format=torch.channels_last)
    g = _rand(B, C, H, W, seed=2, dtype=dtype).contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_()
    ref = ref_bn(xr)
    gref = torch.autograd.grad(ref, (xr, ref_bn.weight, ref_bn.bias), g.float())
    xs = x.clone().requires_grad_()
    assert dgtd.ops.batch_norm_supported(xs, bn)
    if layer == "native":
        out = dgtd.ops.batch_norm(xs, bn)
    else:
        out = H_._BatchNormFn.apply(xs, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, True, bn.momentum, bn.eps)
    got = torch.autograd.grad(out, (xs, bn.weight, bn.bias), g)
    assert out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last)
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(out.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(bn.running_var, ref_bn.running_var, atol=1e-5, rtol=1e-4)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    torch.testing.assert_close(got[0].float(), gref[0], atol=tol, rtol=tol)
    wtol = 2e-4 if dtype == torch.float32 else 2e-2
    for a, b in zip(got[1:], gref[1:]):
        assert a.dtype == torch.float32
        torch.testing.assert_close(a, b, atol=wtol * b.abs().max().item() + 1e-5, rtol=wtol)
    bn.eval(); ref_bn.eval()
    with torch.no_grad():
        assert dgtd.ops.batch_norm_supported(x, bn)
        torch.testing.assert_close(dgtd.ops.batch_norm(x, bn).float(), ref_bn(x.float()), atol=tol, rtol=tol)
    assert int(bn.num_batches_tracked) == 1
    assert not dgtd.ops.batch_norm_supported(x.clone().requires_grad_(), bn)      # eval mode + gradient wanted: torch's path


# ---------------------------------------------------------------------------------------------- dense 3x3 convolution (NHWC bf16 MFMA)
def _conv_ref(x, w, b, relu):
    y = F.conv2d(x.float(), w.float(), b.float() if b is not None else None, padding=1)
    return F.relu(y) if relu else y


@pytest.mark.parametrize("B,C,Co,H,W", [(2, 24, 24, 32, 32), (1, 24, 24, 40, 48), (2, 32, 32, 16, 16), (2, 64, 64, 32, 32),
                                         (1, 96, 96, 64, 64), (2, 24, 64, 128, 128), (3, 64, 32, 24, 32), (8, 24, 24, 128, 128)])
@pytest.mark.parametrize("relu,bias", [(True, True), (False, False)])
@pytest.mark.parametrize("half", HALVES, ids=str)
def test_conv3x3_single_fwd_bwd(dgtd, B, C, Co, H, W, relu, bias, half):
    """One convolution: forward, input gradient (through the fused ReLU mask), weight and bias gradients vs fp32 torch."""
    x = _rand(B, C, H, W, seed=1, dtype=half).contiguous(memory_format=torch.channels_last)
    w = (_rand(Co, C, 3, 3, seed=2) / math.sqrt(9 * C)).to(half)
    b = (0.1 * _rand(Co, seed=3)).to(half) if bias else None
    g = _rand(B, Co, H, W, seed=4, dtype=half).contiguous(memory_format=torch.channels_last)
    xr, wr = x.float().requires_grad_(), w.float().requires_grad_()
    br = b.float().requires_grad_() if bias else None
    ref = _conv_ref(xr, wr, br, relu)
    grads = torch.autograd.grad(ref, (xr, wr) + ((br,) if bias else ()), g.float())
    xs, wsn = x.clone().requires_grad_(), w.clone().requires_grad_()
    bs = b.clone().requires_grad_() if bias else None
    y = dgtd.ops.conv3x3(xs, wsn, bs, relu)
    assert y.shape == ref.shape and y.dtype == half
    torch.testing.assert_close(y.float(), ref, atol=2e-2, rtol=2e-2)
    # the ReLU mask is taken from the bf16 output: compare gradients where the fp32 reference is not within rounding of 0
    got = torch.autograd.grad(y, (xs, wsn) + ((bs,) if bias else ()), g)
    if relu:   # reference gradients with the mask of the bf16 output, so a pre-activation that rounds across 0 does not count as an error
        ref2 = F.conv2d(xr, wr, br, padding=1)
        grads = torch.autograd.grad(ref2, (xr, wr) + ((br,) if bias else ()), g.float() * (y.float() > 0))
    torch.testing.assert_close(got[0].float(), grads[0], atol=3e-2, rtol=3e-2)
    n = B * H * W
    torch.testing.assert_close(got[1].float(), grads[1], atol=2e-2 * math.sqrt(n), rtol=3e-2)
    if bias:
        torch.testing.assert_close(got[2].float(), grads[2], atol=2e-2 * math.sqrt(n), rtol=3e-2)


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("half", HALVES, ids=str)
def test_conv3x3_stack_matches_separate_convs(dgtd, shared, half):
    """Z = 5 convolutions in one launch (shared or own inputs) == Z separate fp32 convolutions, incl. the summed shared-input grad."""
    Z, B, C, H, W = 5, 2, 24, 32, 48
    xs = [_rand(B, H, W, C, seed=10 + z, dtype=half) for z in range(1 if shared else Z)]
    ws = [(_rand(C, C, 3, 3, seed=20 + z) / math.sqrt(9 * C)).to(half).requires_grad_() for z in range(Z)]
    bs = [(0.1 * _rand(C, seed=30 + z)).to(half).requires_grad_() for z in range(Z)]
    g = _rand(Z, B, H, W, C, seed=5, dtype=half)
    xin = (xs[0] if shared else torch.stack(xs)).clone().requires_grad_()
    y = dgtd.ops.conv3x3_stack(xin, ws, bs, True)
    assert y.shape == (Z, B, H, W, C)
    got = torch.autograd.grad(y, [xin] + ws + bs, g)
    dx_ref = torch.zeros_like(xin, dtype=torch.float32)
    for z in range(Z):
        xr = (xs[0] if shared else xs[z]).float().permute(0, 3, 1, 2).requires_grad_()
        wr, br = ws[z].detach().float().requires_grad_(), bs[z].detach().float().requires_grad_()
        pre = F.conv2d(xr, wr, br, padding=1)
        torch.testing.assert_close(y[z].float(), F.relu(pre).permute(0, 2, 3, 1), atol=2e-2, rtol=2e-2)
        gz = g[z].float().permute(0, 3, 1, 2) * (y[z].float().permute(0, 3, 1, 2) > 0)
        dx, dw, db = torch.autograd.grad(pre, (xr, wr, br), gz)
        if shared:
            dx_ref += dx.permute(0, 2, 3, 1)
        else:
            dx_ref[z] = dx.permute(0, 2, 3, 1)
        torch.testing.assert_close(got[1 + z].float(), dw, atol=2e-2 * math.sqrt(B * H * W), rtol=3e-2)
        torch.testing.assert_close(got[1 + Z + z].float(), db, atol=2e-2 * math.sqrt(B * H * W), rtol=3e-2)
    torch.testing.assert_close(got[0].float(), dx_ref, atol=6e-2 if shared else 3e-2, rtol=3e-2)


@pytest.mark.parametrize("B,C,Hi,Wi,Ho,Wo", [(2, 32, 16, 16, 32, 32), (2, 32, 16, 16, 64, 64), (1, 64, 24, 20, 48, 40), (2, 32, 128, 128, 64, 64),
                                             (2, 8, 7, 9, 19, 13), (1, 96, 32, 32, 64, 64), (2, 24, 2, 2, 16, 16), (1, 24, 16, 16, 128, 128), (2, 8, 3, 5, 40, 33)])
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_bilinear_resize_nhwc(dgtd, B, C, Hi, Wi, Ho, Wo, align, dtype):
    """Forward and gather-backward vs F.interpolate in fp32 (x2, x4, x0.5 and a ragged non-integer ratio)."""
    x = _rand(B, C, Hi, Wi, seed=1, dtype=dtype).contiguous(memory_format=torch.channels_last)
    g = _rand(B, C, Ho, Wo, seed=2, dtype=dtype).contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_()
    ref = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=align)
    gx, = torch.autograd.grad(ref, xr, g.float())
    xs = x.clone().requires_grad_()
    y = dgtd.ops.bilinear_resize(xs, Ho, Wo, align)
    hx, = torch.autograd.grad(y, xs, g)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(hx.float(), gx, atol=tol * 8, rtol=tol)


# ---------------------------------------------------------------------------------------------- Linear fused with its consumer
@pytest.mark.parametrize("rows,K,N", [(1024, 128, 512), (8192, 512, 2048), (300, 64, 256)])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_linear_gelu_fused_backward(dgtd, rows, K, N, dtype):
    """gelu(x W^T + b): value, dx, dW and the bias gradient (GELU' fused with the column sum) vs fp32 torch."""
    x = _rand(2, rows // 2, K, seed=1, dtype=dtype)
    w = (_rand(N, K, seed=2) / math.sqrt(K)).to(dtype).requires_grad_()
    b = (0.1 * _rand(N, seed=3)).to(dtype).requires_grad_()
    g = _rand(2, rows // 2, N, seed=4, dtype=dtype)
    xr, wr, br = x.float().requires_grad_(), w.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    ref = F.gelu(F.linear(xr, wr, br))
    gx, gw, gb = torch.autograd.grad(ref, (xr, wr, br), g.float())
    xs = x.clone().requires_grad_()
    y = dgtd.ops.linear_gelu(xs, w, b)
    hx, hw, hb = torch.autograd.grad(y, (xs, w, b), g)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(hx.float(), gx, atol=tol, rtol=tol)
    torch.testing.assert_close(hw.float(), gw, atol=tol * math.sqrt(rows), rtol=tol)
    torch.testing.assert_close(hb.float(), gb, atol=tol * math.sqrt(rows), rtol=tol)


@pytest.mark.parametrize("rows,K,N", [(1024, 512, 128), (8192, 2048, 512), (300, 256, 64)])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("with_s,with_g", [(True, True), (True, False), (False, False)])
def test_linear_residual_fused_backward(dgtd, rows, K, N, dtype, with_s, with_g):
    """x + s*gamma*(h W^T + b): value and every gradient (dh, dW, db, dx, dgamma) vs fp32 torch."""
    B = 2
    h = _rand(B, rows // B, K, seed=1, dtype=dtype)
    x = _rand(B, rows // B, N, seed=5, dtype=dtype)
    w = (_rand(N, K, seed=2) / math.sqrt(K)).to(dtype).requires_grad_()
    b = (0.1 * _rand(N, seed=3)).to(dtype).requires_grad_()
    s = torch.tensor([0.0, 1.25], device="cuda") if with_s else None
    gamma = (0.5 + 0.1 * _rand(N, seed=6)).requires_grad_() if with_g else None
    g = _rand(B, rows // B, N, seed=4, dtype=dtype)
    hr, xr = h.float().requires_grad_(), x.float().requires_grad_()
    wr, br = w.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    y = F.linear(hr, wr, br)
    if with_g:
        y = y * gamma
    if with_s:
        y = y * s.view(B, 1, 1)
    ref = xr + y
    ins = [hr, xr, wr, br] + ([gamma] if with_g else [])
    rg = torch.autograd.grad(ref, ins, g.float())
    hs, xs = h.clone().requires_grad_(), x.clone().requires_grad_()
    out = dgtd.ops.linear_residual(hs, w, b, xs, s, gamma)
    got = torch.autograd.grad(out, [hs, xs, w, b] + ([gamma] if with_g else []), g)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(out.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(got[0].float(), rg[0], atol=tol, rtol=tol)
    torch.testing.assert_close(got[1].float(), rg[1], atol=tol, rtol=tol)
    for a_, r_ in zip(got[2:], rg[2:]):
        torch.testing.assert_close(a_.float(), r_, atol=tol * math.sqrt(rows), rtol=tol)


@pytest.mark.parametrize("C", [64, 320, 512])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_layernorm_fork_adds_skip_gradient(dgtd, C, dtype):
    """(LN(x), x): the gradient of the skip branch is added inside the LayerNorm backward; result == autograd's separate add."""
    rows = 1000
    x = _rand(rows, C, seed=1, dtype=dtype)
    w = (1 + 0.1 * _rand(C, seed=2)).requires_grad_()
    b = (0.1 * _rand(C, seed=3)).requires_grad_()
    g1, g2 = _rand(rows, C, seed=4, dtype=dtype), _rand(rows, C, seed=5, dtype=dtype)
    xr = x.float().requires_grad_()
    ref = F.layer_norm(xr, (C,), w, b, 1e-6)
    gx, gw, gb = torch.autograd.grad([ref, xr * 1.0], (xr, w, b), [g1.float(), g2.float()])
    xs = x.clone().requires_grad_()
    y, skip = dgtd.ops.layer_norm_fork(xs, w, b, 1e-6)
    hx, hw, hb = torch.autograd.grad([y, skip * 1.0], (xs, w, b), [g1, g2])
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(hx.float(), gx, atol=tol, rtol=tol)
    torch.testing.assert_close(hw, gw, atol=(2e-4 if dtype == torch.float32 else 5e-2) * math.sqrt(rows), rtol=5e-2)
    # only the norm output used: the skip gradient is absent
    y2, _ = dgtd.ops.layer_norm_fork(xs, w, b, 1e-6)
    h2, = torch.autograd.grad(y2, xs, g1)
    g2x, = torch.autograd.grad(F.layer_norm(xr, (C,), w, b, 1e-6), xr, g1.float())
    torch.testing.assert_close(h2.float(), g2x, atol=tol, rtol=tol)


@pytest.mark.parametrize("K,C,H", [(7, 128, 16), (7, 512, 12), (3, 256, 10)])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_dwconv_fork_adds_skip_gradient(dgtd, K, C, H, dtype):
    """(dwconv(x), x): the skip-branch gradient is added inside the input-gradient convolution (mode 3)."""
    x = _rand(2, H, H, C, seed=1, dtype=dtype)
    w = (_rand(C, 1, K, K, seed=2) / K).to(dtype).requires_grad_()
    b = (0.1 * _rand(C, seed=3)).to(dtype).requires_grad_()
    g1, g2 = _rand(2, H, H, C, seed=4, dtype=dtype), _rand(2, H, H, C, seed=5, dtype=dtype)
    xr, wr, br = x.float().requires_grad_(), w.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=K // 2, groups=C).permute(0, 2, 3, 1)
    gx, gw, gb = torch.autograd.grad([ref, xr * 1.0], (xr, wr, br), [g1.float(), g2.float()])
    xs = x.clone().requires_grad_()
    y, skip = dgtd.ops.dwconv_fork(xs, w, b)
    hx, hw, hb = torch.autograd.grad([y, skip * 1.0], (xs, w, b), [g1, g2])
    tol = 1e-4 if dtype == torch.float32 else 5e-2
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)
    torch.testing.assert_close(hx.float(), gx, atol=tol, rtol=tol)
    torch.testing.assert_close(hw.float(), gw, atol=tol * 8, rtol=5e-2)
    torch.testing.assert_close(hb.float(), gb, atol=tol * 8, rtol=5e-2)


# ---------------------------------------------------------------------------------------------- depthwise convs at the benchmarked shapes
# the depthwise family is the largest dgtd time category of the bf16 step and had no direct op test at the shapes the
# bench runs.  (K, gelu, B, H, W, C): ConvNeXt dw7x7 at stage 3 (LDS-tiled path with the XCD remap) and stage 1, Mlp dw3x3 + bias +
# erf-GELU (modes 1 and 2) at PVT stage 1 (hidden 512 @128^2), stage 2 (1024 @64^2) and stage 4 (2048 @16^2), plus a ragged map.
DW_SHAPES = [(7, False, 8, 32, 32, 512), (7, False, 8, 128, 128, 128), (3, True, 8, 128, 128, 512), (3, True, 8, 64, 64, 1024),
             (3, True, 8, 16, 16, 2048), (7, False, 2, 20, 12, 256), (3, True, 3, 10, 14, 128)]


@pytest.mark.parametrize("K,gelu,B,H,W,C", DW_SHAPES, ids=[f"k{s[0]}{'g' if s[1] else ''}_{s[2]}x{s[3]}x{s[4]}x{s[5]}" for s in DW_SHAPES])
@pytest.mark.parametrize("half", HALVES, ids=str)
def test_dwconv_benchmarked_shapes_vs_fp32_torch(dgtd, K, gelu, B, H, W, C, half):
    """dgtd_dwconv_fwd modes 0/1/2 (bias + polynomial-erfc GELU and its backward), the input gradient (flipped filter) and
    dgtd_dwconv_bwd_weight for both K, in the 16-bit dtypes, against fp32 F.conv2d (+ exact erf GELU) on the same rounded inputs."""
    if half == torch.float16 and not dgtd.ops._native.ENABLED:
        pytest.skip("fp16 at the large shapes runs through one binding layer (same C ABI)")
    x = _rand(B, H, W, C, seed=1, dtype=half)
    w = (_rand(C, 1, K, K, seed=2) / K).to(half).requires_grad_()
    b = (0.1 * _rand(C, seed=3)).to(half).requires_grad_()
    g = _rand(B, H, W, C, seed=4, dtype=half)
    xr, wr, br = x.float().requires_grad_(), w.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    pre = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=K // 2, groups=C).permute(0, 2, 3, 1)
    ref = F.gelu(pre) if gelu else pre
    gx, gw, gb = torch.autograd.grad(ref, (xr, wr, br), g.float())
    xs = x.clone().requires_grad_()
    y = dgtd.ops.dwconv_nhwc(xs, w, b, gelu)
    hx, hw, hb = torch.autograd.grad(y, (xs, w, b), g)
    assert y.dtype == half and hw.dtype == half
    torch.testing.assert_close(y.float(), ref, atol=3e-2, rtol=2e-2)
    # dx sums K*K taps of 16-bit-rounded du: absolute error grows with sqrt(K*K)
    torch.testing.assert_close(hx.float(), gx, atol=2e-2 * K, rtol=3e-2)
    n = B * H * W
    assert (hw.float() - gw).norm() / gw.norm() < 2e-2, "weight gradient"
    assert (hb.float() - gb).norm() / gb.norm() < 2e-2, "bias gradient"
    torch.testing.assert_close(hw.float(), gw, atol=3e-2 * math.sqrt(n), rtol=5e-2)


DWB = [(7, 5, 8, 32, 32, 512), (7, 3, 2, 20, 12, 256), (3, 4, 2, 16, 16, 1024), (7, 35, 1, 9, 8, 128)]


@pytest.mark.parametrize("K,L,B,H,W,C", DWB, ids=[f"k{s[0]}_n{s[1]}_{s[2]}x{s[3]}x{s[4]}x{s[5]}" for s in DWB])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_dwconv_weight_gradients_deferred_and_batched(dgtd, K, L, B, H, W, C, dtype):
    """Deferred weight-gradient phase: with deferral on, the depthwise nodes only park their (input, output-gradient) pairs; one
    dgtd_dwconv_bwd_weight_batched launch per shape at the flush (n = 35 crosses the 32-entry table) + one transposing multi-reduce
    must give every layer's weight / bias gradient, against fp32 F.conv2d.  A second shape parked in between keeps its own group."""
    nat = dgtd.ops._native.ops()
    if nat is None:
        pytest.skip("deferral lives in the C++ binding layer")
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    layers = []
    for i in range(L):
        x = _rand(B, H, W, C, seed=10 + i, dtype=dtype)
        w = (_rand(C, 1, K, K, seed=50 + i) / K).to(dtype).requires_grad_()
        b = (0.1 * _rand(C, seed=90 + i)).to(dtype).requires_grad_()
        g = _rand(B, H, W, C, seed=130 + i, dtype=dtype)
        layers.append((x, w, b, g))
    other = (_rand(2, 8, 8, 128, seed=3, dtype=dtype), (_rand(128, 1, 3, 3, seed=4) / 3).to(dtype).requires_grad_(), None, _rand(2, 8, 8, 128, seed=5, dtype=dtype))
    nat.set_deferred(True)
    try:
        # ONE backward pass over all layers: everything it parks is flushed together by the engine callback at its end
        ys, params, gs = [], [], []
        for i, (x, w, b, g) in enumerate(layers + [other]):
            ys.append(dgtd.ops.dwconv_nhwc(x, w, b, False))
            params.append((w, b) if b is not None else (w,))
            gs.append(g)
        done = nat.flushed_reductions()
        flat = torch.autograd.grad(ys, [p for ps in params for p in ps], gs)
        assert nat.flushed_reductions() - done == L + 1 and nat.pending_reductions() == 0
        got, k = [], 0
        for ps in params:
            got.append(flat[k:k + len(ps)])
            k += len(ps)
    finally:
        nat.set_deferred(False)
    assert nat.pending_reductions() == 0
    for (x, w, b, g), gr in zip(layers + [other], got):
        wr = w.detach().float().requires_grad_()
        br = b.detach().float().requires_grad_() if b is not None else None
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wr, br, padding=w.shape[-1] // 2, groups=w.shape[0]).permute(0, 2, 3, 1)
        rw = torch.autograd.grad(ref, (wr, br) if b is not None else (wr,), g.float())
        assert gr[0].dtype == dtype and gr[0].shape == w.shape
        assert (gr[0].float() - rw[0]).norm() / rw[0].norm() < tol, "weight gradient"
        if b is not None:
            assert (gr[1].float() - rw[1]).norm() / rw[1].norm() < tol, "bias gradient"


@pytest.mark.parametrize("half", HALVES, ids=str)
def test_conv3x3_weight_gradients_deferred_batched_and_shared(dgtd, half):
    """Deferred weight-gradient phase of the single 3x3 convolutions: weights used 4x / 4x / 1x per step at one geometry, one weight at a
    second geometry and one weight at TWO geometries (must keep the immediate path).  Every weight's accumulated gradient (autograd
    semantics: sum over its calls) against fp32 F.conv2d; the shared weights receive exactly one gradient tensor."""
    nat = dgtd.ops._native.ops()
    if nat is None:
        pytest.skip("deferral lives in the C++ binding layer")
    def leaf(co, ci, seed, bias=False):
        w = (_rand(co, ci, 3, 3, seed=seed) * 0.05).to(half).contiguous(memory_format=torch.channels_last).requires_grad_()
        b = (0.1 * _rand(co, seed=seed + 1)).to(half).requires_grad_() if bias else None
        return w, b
    wa, wb_, wc, wd, we = leaf(96, 96, 1), leaf(96, 96, 3), leaf(96, 96, 5, True), leaf(64, 64, 7), leaf(32, 32, 9)
    calls = [(wa, 16)] * 4 + [(wb_, 16)] * 4 + [(wc, 16)] + [(wd, 32)] * 2 + [(we, 16), (we, 32)]
    xs = [(_rand(2, w.shape[1], S, S, seed=20 + i).to(half).contiguous(memory_format=torch.channels_last).requires_grad_(),
           _rand(2, w.shape[0], S, S, seed=60 + i).to(half).contiguous(memory_format=torch.channels_last)) for i, ((w, _), S) in enumerate(calls)]
    nat.set_deferred(True)
    try:
        ys = [dgtd.ops.conv3x3(x, w, b) for ((w, b), S), (x, g) in zip(calls, xs)]
        done = nat.flushed_reductions()
        torch.autograd.backward(ys, [g for _, g in xs])
        assert nat.flushed_reductions() - done == 4 + 4 + 1 + 2       # `we` runs at two geometries: not parked; flushed at the end of the pass
    finally:
        nat.set_deferred(False)
    assert nat.pending_reductions() == 0
    for w, b in (wa, wb_, wc, wd, we):
        wr = w.detach().float().requires_grad_()
        br = b.detach().float().requires_grad_() if b is not None else None
        refs = [F.conv2d(x.detach().float(), wr, br, padding=1) for ((w2, _), S), (x, g) in zip(calls, xs) if w2 is w]
        gs = [g.float() for ((w2, _), S), (x, g) in zip(calls, xs) if w2 is w]
        torch.autograd.backward(refs, gs)
        assert (w.grad.float() - wr.grad).norm() / wr.grad.norm() < 2e-2, f"weight {tuple(w.shape)} used {len(refs)}x"
        if b is not None:
            assert (b.grad.float() - br.grad).norm() / br.grad.norm() < 2e-2
    for (x, g), ((w, b), S) in zip(xs, calls):                       # input gradients are immediate either way
        xr = x.detach().float().requires_grad_()
        F.conv2d(xr, w.detach().float(), None if b is None else b.detach().float(), padding=1).backward(g.float())
        torch.testing.assert_close(x.grad.float(), xr.grad, atol=5e-2, rtol=5e-2)


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_dwconv_prepared_weights_follow_the_parameters(dgtd, dtype):
    """Under deferral a depthwise layer packs its weights once into a persistent buffer and every later step start re-packs all registered
    layers in one launch: the outputs must follow in-place weight updates between steps (the optimizer), a bias that appears, and a NEW
    weight tensor that reuses the address of a freed one."""
    nat = dgtd.ops._native.ops()
    if nat is None:
        pytest.skip("prepared weights live in the C++ binding layer")
    tol = 1e-4 if dtype == torch.float32 else 3e-2

    def ref(x, w, b):
        return F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None if b is None else b.float(), padding=w.shape[-1] // 2, groups=w.shape[0]).permute(0, 2, 3, 1)

    def step(x, w, b):
        nat.set_deferred(True)                      # = reducer.zero_grad(): refreshes every registered layer
        try:
            return dgtd.ops.dwconv_nhwc(x, w, b, False)
        finally:
            nat.set_deferred(False)

    for K, C in ((7, 128), (3, 256)):
        x = _rand(2, 12, 12, C, seed=1, dtype=dtype)
        w = (_rand(C, 1, K, K, seed=2) / K).to(dtype).requires_grad_()
        b = (0.1 * _rand(C, seed=3)).to(dtype).requires_grad_()
        for it in range(3):
            torch.testing.assert_close(step(x, w, b).float(), ref(x, w, b), atol=tol, rtol=tol)
            with torch.no_grad():                   # what the optimizer does between steps
                w.mul_(-1.5); b.add_(0.25)
        torch.testing.assert_close(step(x, w, None).float(), ref(x, w, None), atol=tol, rtol=tol)      # same weight, now without bias
        ptr = w.data_ptr()
        del w
        w2 = (_rand(C, 1, K, K, seed=9) / K).to(dtype).requires_grad_()                                 # usually lands on the freed address
        torch.testing.assert_close(step(x, w2, b).float(), ref(x, w2, b), atol=tol, rtol=tol), ptr


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_prelu_shared_slope_gradient_under_deferral(dgtd, dtype):
    """One nn.PReLU() slope shared by every activation (cod.py:686): under deferral all backward calls add into one fp32 accumulator
    and autograd receives a single gradient, converted at the flush.  Against fp32 torch.prelu summed over the calls; with shared
    deferral off (eager overlap mode of the reducer) the per-call path must give the same value."""
    nat = dgtd.ops._native.ops()
    if nat is None:
        pytest.skip("deferral lives in the C++ binding layer")
    res = {}
    for shared in (True, False):
        a = torch.full((1,), 0.25, device="cuda", dtype=dtype).requires_grad_()
        xs = [_rand(2, 32, 16, 16, seed=30 + i, dtype=dtype).requires_grad_() for i in range(5)]
        gs = [_rand(2, 32, 16, 16, seed=40 + i, dtype=dtype) for i in range(5)]
        nat.set_shared_deferral(shared)
        nat.set_deferred(True)
        try:
            torch.autograd.backward([dgtd.ops.prelu(x, a) for x in xs], gs)
        finally:
            nat.set_deferred(False)
            nat.set_shared_deferral(True)
        ar = a.detach().float().requires_grad_()
        per_call = [torch.autograd.grad(torch.prelu(x.detach().float(), ar), ar, g.float())[0] for x, g in zip(xs, gs)]
        ref, scale = sum(per_call), sum(p.abs() for p in per_call).item()     # the calls cancel: the error scale is that of the terms
        tol = (1e-5 if dtype == torch.float32 else 8e-3) * scale              # 16-bit: the per-call path rounds every term to 16 bits
        assert abs(a.grad.float().item() - ref.item()) <= tol, (shared, a.grad.item(), ref.item(), scale)
        res[shared] = a.grad.float().item()
        if shared and dtype != torch.float32:                                 # one fp32 accumulator, one rounding at the end
            assert abs(res[True] - ref.item()) <= 2 ** -8 * abs(ref.item()) + 1e-4 * scale


@pytest.mark.parametrize("scale", [2, 4, 8])
@pytest.mark.parametrize("half", [torch.float32] + HALVES, ids=str)
def test_prompt_tail_weight_gradient_with_channels_last_weights(dgtd, scale, half):
    """Round-2 finding: the folded prompt-decoder tail (ShapePropDecoder.forward_tokens, cod.py:1224-1226 + :1471) used
    F.avg_pool2d on the 3x3 weight, whose BACKWARD is wrong on ROCm for channels_last inputs - exactly the storage the gradient
    reducer gives these weights.  The weight gradient of the production formulation, with the weight as a channels_last leaf, against
    the literal fp32 sequence conv3x3 -> bilinear."""
    C, S = 128, 32
    dec = dgtd.nn.ShapePropDecoder(C, 24).cuda()
    conv = dec.decoder[4]
    h = _rand(2, 24, S, S, seed=1).relu()
    H = S // scale
    w32 = conv.weight.detach().clone().requires_grad_()
    ref = F.interpolate(F.conv2d(h, w32, conv.bias, padding=1), size=(H, H), mode="bilinear", align_corners=False)
    gout = _rand(*ref.shape, seed=2)
    gw_ref, = torch.autograd.grad(ref, w32, gout)
    w = conv.weight.detach().to(half).contiguous(memory_format=torch.channels_last).clone(memory_format=torch.preserve_format).requires_grad_()
    conv._w, conv._b = w, conv.bias.detach().to(half)
    hh = h.to(half).contiguous(memory_format=torch.channels_last)
    y = dec.forward_tokens(None, H, H, trunk=hh.permute(0, 2, 3, 1))
    g, = torch.autograd.grad(y, w, gout.flatten(2).transpose(1, 2).to(y.dtype))
    torch.testing.assert_close(y.float(), ref.flatten(2).transpose(1, 2), atol=2e-5 if half == torch.float32 else 3e-2, rtol=3e-2)
    rel = float((g.float() - gw_ref).norm() / gw_ref.norm())
    assert rel < (1e-4 if half == torch.float32 else 1e-2), rel


# ---------------------------------------------------------------------------------------------- cat / stack replacements
@pytest.mark.parametrize("src,dst", [(torch.bfloat16, torch.float32), (torch.float16, torch.float32), (torch.float32, torch.float32),
                                     (torch.bfloat16, torch.bfloat16), (torch.float32, torch.float16)], ids=str)
def test_multi_copy_gathers_and_scatters(dgtd, src, dst):
    """dgtd_multi_copy: 300 tensors of ragged sizes (odd lengths, unaligned slots, one empty) into one flat buffer with conversion,
    bit-exact against per-tensor copy_, more than one 128-entry table per call; and back (to_tensors)."""
    g = torch.Generator().manual_seed(0)
    sizes = [int(v) for v in torch.randint(1, 5000, (300,), generator=g)] + [0, 8, 2048, 2049, 100003]
    ts = [_rand(n, seed=i, dtype=src) for i, n in enumerate(sizes)]
    offs, off = [], 3
    for n in sizes:
        offs.append(off)
        off += n + (n % 5)                    # gaps: slots are neither contiguous nor aligned
    flat = torch.full((off + 7,), -7.0, device="cuda", dtype=dst)
    want = flat.clone()
    for t, o in zip(ts, offs):
        want[o:o + t.numel()].copy_(t)
    dgtd._lib.multi_copy(ts, offs, flat)
    assert torch.equal(flat, want)            # including the untouched gaps
    back = [torch.zeros_like(t) for t in ts]
    dgtd._lib.multi_copy(back, offs, flat, to_tensors=True)
    for t, b in zip(ts, back):
        assert torch.equal(b, t.to(dst).to(src))


def test_stack_and_cat_channels_match_torch(dgtd):
    xs = [_rand(6, 3, 3, 24, seed=i, dtype=torch.bfloat16).requires_grad_() for i in range(16)]
    g = _rand(16, 6, 3, 3, 24, seed=99, dtype=torch.bfloat16)
    y = dgtd.ops.stack(xs)
    assert torch.equal(y, torch.stack([x.detach() for x in xs]))
    grads = torch.autograd.grad(y, xs, g)
    assert all(torch.equal(a, g[i]) for i, a in enumerate(grads))
    maps = [_rand(2, c, 12, 20, seed=c, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_() for c in (32, 64, 24)]
    out = dgtd.ops.cat_channels(maps)
    ref = torch.cat([m.detach() for m in maps], dim=1)
    assert torch.equal(out, ref) and out.is_contiguous(memory_format=torch.channels_last)
    go = _rand(*ref.shape, seed=5, dtype=torch.bfloat16)
    gs = torch.autograd.grad(out, maps, go)
    assert torch.equal(gs[1], go[:, 32:96]) and torch.equal(gs[2], go[:, 96:])


# ---------------------------------------------------------------------------------------------- dense conv = patch gather + GEMM
CONV_CASES = [  # (B, Ci, H, W, Co, K, stride, pad, layout)
    (2, 3, 64, 64, 64, 7, 4, 3, "nchw"),          # patch_embed1 on the fp32 NCHW image
    (2, 64, 32, 32, 128, 3, 2, 1, "nhwc"),        # patch_embed2..4
    (2, 24, 32, 32, 40, 4, 2, 1, "nhwc"),         # folded prompt tail, stage 2 (padding 1)
    (2, 24, 32, 32, 40, 4, 4, 0, "nhwc"),         # stage 3: windows tile the map
    (2, 24, 30, 30, 40, 4, 8, 0, "offset"),       # stage 4: offset view h[:, :, 2:, 2:]
    (2, 64, 16, 16, 32, 8, 4, 2, "nhwc"),         # Hitnet.compress_out
    (2, 32, 12, 20, 1, 1, 1, 0, "nhwc"),          # 1-channel head
    (1, 96, 24, 24, 96, 3, 1, 1, "nhwc"),         # fp32-mode CAB conv
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[f"ci{c[1]}_k{c[5]}s{c[6]}p{c[7]}_{c[8]}" for c in CONV_CASES])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
def test_conv2d_gemm_matches_conv2d(dgtd, case, dtype):
    """ops.conv2d_gemm (dgtd_im2col + library GEMM, dgtd_col2im in the backward) against F.conv2d in fp32: value, input gradient,
    weight and bias gradients, for every geometry the model uses, with the weight stored O,H,W,I like the reducer stores it."""
    B, Ci, H, W, Co, K, S, P, layout = case
    full = _rand(B, Ci, H + (2 if layout == "offset" else 0), W + (2 if layout == "offset" else 0), seed=1, dtype=torch.float32 if layout == "nchw" else dtype)
    if layout != "nchw":
        full = full.contiguous(memory_format=torch.channels_last)
    full.requires_grad_(layout != "nchw")
    x = full[:, :, 2:, 2:] if layout == "offset" else full
    w = (_rand(Co, Ci, K, K, seed=2) / math.sqrt(Ci * K * K)).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = (0.1 * _rand(Co, seed=3)).to(dtype).requires_grad_()
    xr = x.detach().float().requires_grad_()
    wr, br = w.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    ref = F.conv2d(xr, wr, br, stride=S, padding=P)
    g = _rand(*ref.shape, seed=4, dtype=dtype)
    with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
        y = dgtd.ops.conv2d_gemm(x, w, b, S, P)
    assert y.shape == ref.shape and y.dtype == dtype
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)
    ins = ([full] if layout != "nchw" else []) + [w, b]
    got = torch.autograd.grad(y, ins, g)
    want = torch.autograd.grad(ref, ([xr] if layout != "nchw" else []) + [wr, br], g.float())
    if layout != "nchw":
        gx = got[0][:, :, 2:, 2:] if layout == "offset" else got[0]
        torch.testing.assert_close(gx.float(), want[0], atol=tol * 4, rtol=tol)
        if layout == "offset":
            assert float(got[0][:, :, :2].abs().sum()) == 0.0            # nothing flows into the skipped border
    n = B * ref.shape[2] * ref.shape[3]
    assert (got[-2].float() - want[-2]).norm() / want[-2].norm() < (1e-4 if dtype == torch.float32 else 2e-2)
    torch.testing.assert_close(got[-1].float(), want[-1], atol=tol * math.sqrt(n), rtol=tol)


@pytest.mark.parametrize("B,S", [(2, 64), (3, 96), (8, 512)])
def test_ssim_value_vs_oracle(dgtd, B, S):
    """dgtd_ssim_value against the oracle's SSIM module on the min-max normalised high-pass image (cod.py:143-144, :316-351)."""
    from oracle import cod_cpu
    g = torch.Generator().manual_seed(S)
    x_hp = torch.rand(B, 3, S, S, generator=g) * 3.0
    img = torch.randn(B, 3, S, S, generator=g)
    e = (x_hp - x_hp.min()) / (x_hp.max() - x_hp.min() + 1e-8)
    want = cod_cpu.ssim_value(e, img).item()
    got = dgtd.ops.ssim_value(x_hp.cuda(), img.cuda()).item()
    assert abs(got - want) < 2e-6, (got, want)


# ---------------------------------------------------------------------------------------------- the package's own MFMA GEMM (csrc/gemm.hip)
GEMM_SHAPES = [(128, 64, 64), (256, 128, 128), (1024, 512, 128), (8192, 2048, 512), (8192, 512, 2048), (2048, 320, 1280), (2048, 1280, 320),
               (32768, 64, 512)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=str)
def test_own_gemm_entries_vs_fp32_torch(dgtd, M, N, K, dtype):
    """dgtd_gemm_bias / _bias_gelu / _bias_residual / _gelu_bwd and dgtd_transpose_batched through the C ABI against fp32 torch on the same
    16-bit operands: 128- and 64-wide column tiles, one to 32 k-steps, both 16-bit types (cod.py:852-859, :900-921, :1097-1116)."""
    import ctypes as C
    L = dgtd._lib
    code, st = L.dtype_code(torch.empty(1, dtype=dtype)), L.stream_ptr()
    assert L.load().dgtd_gemm_supported(M, N, K, code) == 1 and L.load().dgtd_gemm_supported(M + 1, N, K, code) == 0
    x = (_rand(M, K, seed=1) * 0.5).to(dtype)
    w = (_rand(N, K, seed=2) / math.sqrt(K)).to(dtype)
    b = (0.1 * _rand(N, seed=3)).to(dtype)
    tol = dict(atol=2e-2, rtol=2e-2)
    ref = x.float() @ w.float().t() + b.float()
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    L.call("dgtd_gemm_bias", x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, code, st)
    torch.testing.assert_close(out.float(), ref, **tol)
    L.call("dgtd_gemm_bias", x.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, code, st)
    torch.testing.assert_close(out.float(), ref - b.float(), **tol)
    pre, h = torch.empty_like(out), torch.empty_like(out)
    L.call("dgtd_gemm_bias_gelu", x.data_ptr(), w.data_ptr(), b.data_ptr(), pre.data_ptr(), h.data_ptr(), M, N, K, code, st)
    torch.testing.assert_close(pre.float(), ref, **tol)
    torch.testing.assert_close(h.float(), F.gelu(pre.float()), atol=1e-2, rtol=1e-2)          # GELU of the STORED pre-activation
    h2 = torch.empty_like(out)
    L.call("dgtd_gemm_bias_gelu", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, h2.data_ptr(), M, N, K, code, st)
    assert torch.equal(h, h2)
    B = 2
    res = _rand(M, N, seed=4, dtype=dtype)
    s = torch.tensor([0.0, 1.25], device="cuda")
    gamma = 0.5 + 0.1 * _rand(N, seed=5)
    y, o = torch.empty_like(out), torch.empty_like(out)
    L.call("dgtd_gemm_bias_residual", x.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), s.data_ptr(), gamma.data_ptr(), y.data_ptr(),
           o.data_ptr(), M, N, K, M // B, code, st)
    torch.testing.assert_close(y.float(), ref, **tol)
    want = res.float() + s.repeat_interleave(M // B)[:, None] * gamma[None, :] * y.float()
    torch.testing.assert_close(o.float(), want, **tol)
    L.call("dgtd_gemm_bias_residual", x.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), None, None, None, o.data_ptr(), M, N, K, 1, code, st)
    torch.testing.assert_close(o.float(), res.float() + ref, **tol)
    # input gradient through the GELU: dpre = (dy W) * gelu'(pre) on the transposed copy of W, + column partials
    dy = (0.1 * _rand(M, N, seed=6)).to(dtype)            # gradient of a Linear [N -> ...] whose INPUT is gelu(pre [M, K2]); here K2 := K
    w2 = (_rand(N, K, seed=7) / math.sqrt(N)).to(dtype)   # that Linear's weight [out = N, in = K]
    w2t = torch.empty(K, N, device="cuda", dtype=dtype)
    P, I = C.c_void_p * 1, C.c_int * 1
    L.call("dgtd_transpose_batched", P(w2.data_ptr()), P(w2t.data_ptr()), I(N), I(K), 1, code, st)
    assert torch.equal(w2t, w2.t().contiguous())
    prek = (_rand(M, K, seed=8)).to(dtype)
    dpre = torch.empty(M, K, device="cuda", dtype=dtype)
    if L.load().dgtd_gemm_supported(M, K, N, code):
        ws = torch.empty(L.load().dgtd_gemm_gelu_bwd_workspace(M, K) // 4, device="cuda", dtype=torch.float32)
        nb = C.c_int(0)
        L.call("dgtd_gemm_gelu_bwd", dy.data_ptr(), w2t.data_ptr(), prek.data_ptr(), dpre.data_ptr(), ws.data_ptr(), C.byref(nb), M, K, N, code, st)
        p32 = prek.float().requires_grad_()
        F.gelu(p32).backward(dy.float() @ w2.float())
        torch.testing.assert_close(dpre.float(), p32.grad, atol=2e-3, rtol=2e-2)
        torch.testing.assert_close(ws.view(nb.value, K).sum(0), dpre.float().sum(0), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("rows,C", [(1024, 128), (8192, 512), (2048, 1024)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=str)
@pytest.mark.parametrize("with_s", [True, False])
def test_mlp_residual_fused_node(dgtd, rows, C, dtype, with_s):
    """x + s*gamma*(gelu(v W1^T + b1) W2^T + b2) as ONE node (convnext_Block's pointwise half, cod.py:1097-1116): value and every
    gradient (dv, dW1, db1, dW2, db2, dx, dgamma) vs fp32 torch; the same tolerances as the two-node form it replaces."""
    B, H4 = 2, 4 * C
    v = _rand(B, rows // B, C, seed=1, dtype=dtype)
    x = _rand(B, rows // B, C, seed=5, dtype=dtype)
    w1 = (_rand(H4, C, seed=2) / math.sqrt(C)).to(dtype).requires_grad_()
    b1 = (0.1 * _rand(H4, seed=3)).to(dtype).requires_grad_()
    w2 = (_rand(C, H4, seed=7) / math.sqrt(H4)).to(dtype).requires_grad_()
    b2 = (0.1 * _rand(C, seed=8)).to(dtype).requires_grad_()
    s = torch.tensor([0.0, 1.25], device="cuda") if with_s else None
    gamma = (0.5 + 0.1 * _rand(C, seed=6)).requires_grad_()
    g = _rand(B, rows // B, C, seed=4, dtype=dtype)
    vr, xr = v.float().requires_grad_(), x.float().requires_grad_()
    p32 = [t.detach().float().requires_grad_() for t in (w1, b1, w2, b2)]
    y = F.linear(F.gelu(F.linear(vr, p32[0], p32[1])), p32[2], p32[3]) * gamma
    if with_s:
        y = y * s.view(B, 1, 1)
    ref = xr + y
    rg = torch.autograd.grad(ref, [vr, xr, *p32, gamma], g.float())
    vs, xs = v.clone().requires_grad_(), x.clone().requires_grad_()
    assert dgtd.ops._native.ops().gemm_ok(rows, H4, C, dgtd._lib.dtype_code(v)), "shape must take the fused node"
    out = dgtd.ops.mlp_residual(vs, w1, b1, w2, b2, xs, s, gamma)
    got = torch.autograd.grad(out, [vs, xs, w1, b1, w2, b2, gamma], g)
    tol = 3e-2
    torch.testing.assert_close(out.float(), ref, atol=tol, rtol=tol)
    names = ["dv", "dx", "dW1", "db1", "dW2", "db2", "dgamma"]
    for n, a, b_ in zip(names, got, rg):
        scale = math.sqrt(rows) if n in ("dW1", "db1", "dW2", "db2", "dgamma") else 1.0
        torch.testing.assert_close(a.float(), b_, atol=tol * scale, rtol=tol, msg=lambda m, n=n: f"{n}: {m}")


# ---------------------------------------------------------------------------------------------- CAB: conv3x3 epilogues + the one-node form
@pytest.mark.parametrize("C,S", [(32, 16), (64, 32), (96, 64)])
@pytest.mark.parametrize("half", HALVES, ids=str)
def test_conv3x3_fwd_ex_epilogues(dgtd, C, S, half):
    """dgtd_conv3x3_fwd_ex through the C ABI: PReLU forward (pre-activation + activation from one launch), PReLU backward in the
    epilogue of an input-gradient convolution (+ the slope gradient) and the skip-gradient add, against fp32 torch (cod.py:441-451)."""
    L = dgtd._lib
    code, st = L.dtype_code(torch.empty(1, dtype=half)), L.stream_ptr()
    B = 2
    x = _rand(B, S, S, C, seed=1).to(half)                                   # NHWC
    w = (_rand(C, 3, 3, C, seed=2) * 0.05).to(half)                          # O,H,W,I
    slope = torch.tensor([0.25], device="cuda")
    xr, wr = x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2)
    pre_ref = F.conv2d(xr, wr, padding=1).permute(0, 2, 3, 1)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    L.call("dgtd_conv3x3_fwd_ex", x.data_ptr(), None, w.data_ptr(), None, y.data_ptr(), y2.data_ptr(), None, None, slope.data_ptr(), None,
           1, B, S, S, C, C, 2, 0, code, st)
    torch.testing.assert_close(y2.float(), pre_ref, atol=3e-2, rtol=3e-2)
    want = torch.where(y2.float() > 0, y2.float(), 0.25 * y2.float())
    torch.testing.assert_close(y.float(), want, atol=1e-2, rtol=1e-2)         # PReLU of the stored pre-activation
    # act 3 + add: v = conv(x, w); out = (ref > 0 ? v : slope v) + add; sgrad += sum(v * ref | ref <= 0)
    ref = _rand(B, S, S, C, seed=3).to(half)
    add = _rand(B, S, S, C, seed=4).to(half)
    sg = torch.zeros(1, device="cuda")
    out = torch.empty_like(x)
    L.call("dgtd_conv3x3_fwd_ex", x.data_ptr(), None, w.data_ptr(), None, out.data_ptr(), None, add.data_ptr(), ref.data_ptr(), slope.data_ptr(),
           sg.data_ptr(), 1, B, S, S, C, C, 3, 0, code, st)
    v = pre_ref.to(half).float()
    want = torch.where(ref.float() > 0, v, 0.25 * v) + add.float()
    torch.testing.assert_close(out.float(), want, atol=4e-2, rtol=3e-2)
    sg_ref = (v * ref.float())[ref.float() <= 0].sum()
    assert abs(sg.item() - sg_ref.item()) <= 2e-2 * max(1.0, abs(sg_ref.item()), float((v * ref.float()).abs().sum()) ** 0.5)
    # plain add only
    L.call("dgtd_conv3x3_fwd_ex", x.data_ptr(), None, w.data_ptr(), None, out.data_ptr(), None, add.data_ptr(), None, None, None,
           1, B, S, S, C, C, 0, 0, code, st)
    torch.testing.assert_close(out.float(), pre_ref + add.float(), atol=4e-2, rtol=3e-2)


@pytest.mark.parametrize("C,S,red", [(32, 16, 4), (96, 64, 4)])
@pytest.mark.parametrize("half", HALVES, ids=str)
def test_cab_node_vs_composition_and_fp32(dgtd, C, S, red, half):
    """The CAB as ONE C++ node (ops.cab) against the composition of the separate ops it replaces and against fp32 torch: output and the
    gradients of the input, both convolution weights, the shared PReLU slope and the two channel-attention weights."""
    M = dgtd.nn.modules
    act = torch.nn.PReLU()
    cab = dgtd.nn.CAB(C, 3, red, bias=False, act=act).cuda()
    with torch.no_grad():
        for p in cab.parameters():
            p.copy_((0.2 * _rand(*p.shape, seed=int(p.numel()) % 97)).to(p.dtype))
        act.weight.fill_(0.25)
    x = _rand(2, C, S, S, seed=1).to(half).contiguous(memory_format=torch.channels_last)
    g = _rand(2, C, S, S, seed=2).to(half).contiguous(memory_format=torch.channels_last)
    params = [cab.body[0].weight, cab.body[2].weight, act.weight, cab.CA.conv_du[0].weight, cab.CA.conv_du[2].weight]

    def run(node):
        old = M._USE["cab_node"]
        M._USE["cab_node"] = node
        try:
            xs = x.clone().requires_grad_()
            with torch.autocast("cuda", dtype=half):
                y = cab(xs)
            grads = torch.autograd.grad(y, [xs] + params, g)
        finally:
            M._USE["cab_node"] = old
        return y.detach().float(), [t.float() for t in grads]

    y1, g1 = run(True)
    y0, g0 = run(False)
    torch.testing.assert_close(y1, y0, atol=2e-2, rtol=2e-2)
    names = ["dx", "dw0", "dw1", "da", "dcw1", "dcw2"]
    for n, a, b_ in zip(names, g1, g0):
        tol = 3e-2 * max(1.0, float(b_.abs().max()))
        torch.testing.assert_close(a, b_, atol=tol, rtol=3e-2, msg=lambda m, n=n: f"node vs composition {n}: {m}")
    # fp32 torch reference of the module's reference semantics (cod.py:436-451)
    xr = x.float().requires_grad_()
    pr = [p.detach().float().requires_grad_() for p in params]
    t = F.conv2d(F.prelu(F.conv2d(xr, pr[0], padding=1), pr[2]), pr[1], padding=1)
    gate = torch.sigmoid(F.conv2d(F.relu(F.conv2d(t.mean((2, 3), keepdim=True), pr[3])), pr[4]))
    ref = t * gate + xr
    gr = torch.autograd.grad(ref, [xr] + pr, g.float())
    # 16-bit vs fp32: a pre-activation that rounds across zero flips the PReLU branch of that element (a factor 4 in its gradient), so
    # the comparison is per tensor in relative L2, not per element
    rel = lambda a, b_: float((a - b_).norm() / b_.norm().clamp_min(1e-12))
    assert rel(y1, ref.detach()) < 2e-2, rel(y1, ref.detach())
    for n, a, b_ in zip(names, g1, gr):
        assert rel(a, b_) < 6e-2, (n, rel(a, b_))
