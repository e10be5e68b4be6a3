"""Multi-scale deformable attention (the reference's own native op, twig/ops): oracle vs the vectors generated from the reference's
ms_deform_attn_core_pytorch (CPU), and the HIP kernels vs the oracle with the reference's acceptance thresholds
(twig/ops/test.py:43 double allclose, :68 float rtol 1e-2 / atol 1e-3, :96-99 gradcheck; channel counts of test.py:108)."""
import os

import numpy as np
import pytest
import torch

from oracle import ms_deform_attn_cpu as mc
from oracle.make_golden import GOLDEN_DIR


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN_DIR, "msda.npz"))


def _run_oracle(name, dtype=torch.float64):
    N, M, D, Lq, shapes, P = mc.CASES[name]
    value, shp, loc, attn, grad = mc.case_inputs(name, N, M, D, Lq, shapes, P, dtype)
    value.requires_grad_(); loc.requires_grad_(); attn.requires_grad_()
    y = mc.ms_deform_attn(value, shp, loc, attn)
    gv, gl, ga = torch.autograd.grad(y, (value, loc, attn), grad)
    return (value, shp, loc, attn, grad), y.detach(), gv, gl, ga


@pytest.mark.parametrize("name", list(mc.CASES))
def test_oracle_matches_reference_vectors(G, name):
    _, y, gv, gl, ga = _run_oracle(name)
    np.testing.assert_allclose(y.numpy(), G[f"{name}.out"], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(gl.numpy(), G[f"{name}.grad_loc"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(ga.numpy(), G[f"{name}.grad_attn"], rtol=1e-9, atol=1e-14)
    if f"{name}.grad_value" in G:
        np.testing.assert_allclose(gv.numpy(), G[f"{name}.grad_value"], rtol=1e-9, atol=1e-14)
    else:
        f = gv.double().flatten()
        np.testing.assert_allclose(f[::int(G[f"{name}.grad_value.step"])].float().numpy(), G[f"{name}.grad_value.samples"], rtol=1e-6, atol=1e-9)
        assert abs(f.sum().item() - float(G[f"{name}.grad_value.sum"])) <= 1e-9 * max(1.0, abs(float(G[f"{name}.grad_value.abssum"])))


def _level_start(shp):
    return torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mc.CASES))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_hip_forward_backward_vs_oracle(name, dtype):
    import dgtd
    (value, shp, loc, attn, grad), y, gv, gl, ga = _run_oracle(name)
    v, l_, a = (t.detach().to(dtype).cuda().requires_grad_() for t in (value, loc, attn))
    out = dgtd.ops.MSDeformAttnFunction.apply(v, shp.cuda(), _level_start(shp).cuda(), l_, a, 2)
    hv, hl, ha = torch.autograd.grad(out, (v, l_, a), grad.to(dtype).cuda())
    if dtype == torch.float64:      # twig/ops/test.py:43: torch.allclose defaults
        kw = dict(rtol=1e-5, atol=1e-8)
    else:                           # twig/ops/test.py:68
        kw = dict(rtol=1e-2, atol=1e-3)
    assert out.shape == y.shape and out.dtype == dtype
    assert torch.allclose(out.cpu().double(), y, **kw)
    gkw = kw if dtype == torch.float64 else dict(rtol=1e-2, atol=1e-4)
    assert torch.allclose(hv.cpu().double(), gv, **gkw)
    assert torch.allclose(hl.cpu().double(), gl, **gkw)
    assert torch.allclose(ha.cpu().double(), ga, **gkw)


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [4, 16, 30, 32, 64, 71, 128, 256, 512, 1025])
def test_hip_gradients_by_channel_count(channels):
    """twig/ops/test.py:108 sweeps the channel count through the backward's reduction variants; here one kernel covers them."""
    import dgtd
    N, M, Lq, shapes, P = 1, 2, 2, [(6, 4), (3, 2)], 2
    value, shp, loc, attn, grad = mc.case_inputs(f"ch{channels}", N, M, channels, Lq, shapes, P)
    value.requires_grad_(); loc.requires_grad_(); attn.requires_grad_()
    y = mc.ms_deform_attn(value, shp, loc, attn)
    gv, gl, ga = torch.autograd.grad(y, (value, loc, attn), grad)
    v, l_, a = (t.detach().cuda().requires_grad_() for t in (value, loc, attn))
    out = dgtd.ops.ms_deform_attn(v, shp.cuda(), _level_start(shp).cuda(), l_, a, 2)
    hv, hl, ha = torch.autograd.grad(out, (v, l_, a), grad.cuda())
    for got, want in ((out, y.detach()), (hv, gv), (hl, gl), (ha, ga)):
        assert torch.allclose(got.cpu(), want, rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_hip_gradcheck_small_case():
    """torch.autograd.gradcheck in double on the shapes of twig/ops/test.py:15-20, as check_gradient_numerical does (:96-99)."""
    import dgtd
    N, M, D, Lq, shapes, P = mc.CASES["ref_test"]
    value, shp, loc, attn, _ = mc.case_inputs("gradcheck", N, M, 4, Lq, shapes, P)
    loc = loc.clamp(0.05, 0.95)     # keep finite differences away from the zero-padding kinks
    args = (value.cuda().requires_grad_(), shp.cuda(), _level_start(shp).cuda(), loc.cuda().requires_grad_(), attn.cuda().requires_grad_(), 2)
    assert torch.autograd.gradcheck(dgtd.ops.MSDeformAttnFunction.apply, args, eps=1e-6, atol=1e-6, rtol=1e-4, nondet_tol=1e-12)


@pytest.mark.gpu
def test_half_inputs_compute_in_float32():
    import dgtd
    N, M, D, Lq, shapes, P = mc.CASES["d30"]
    value, shp, loc, attn, _ = mc.case_inputs("half", N, M, D, Lq, shapes, P, torch.float32)
    out = dgtd.ops.ms_deform_attn(value.cuda().bfloat16(), shp.cuda(), _level_start(shp).cuda(), loc.cuda().bfloat16(), attn.cuda().bfloat16(), 2)
    assert out.dtype == torch.float32      # custom_fwd(cast_inputs=torch.float32), ms_deform_attn_func.py:21
    ref = mc.ms_deform_attn(value.bfloat16().float(), shp, loc.bfloat16().float(), attn.bfloat16().float())
    assert torch.allclose(out.cpu(), ref, rtol=1e-2, atol=1e-3)


# ------------------------------------------------------------------------------------------------ the module-level native binding
REF_FUNC = "/root/reference/twig/ops/functions/ms_deform_attn_func.py"


def test_module_binding_exports_the_reference_callables():
    """twig/ops/src/vision.cpp:13-16: a module named MultiScaleDeformableAttention with ms_deform_attn_forward / _backward."""
    import inspect
    import MultiScaleDeformableAttention as MSDA
    f, b = inspect.signature(MSDA.ms_deform_attn_forward), inspect.signature(MSDA.ms_deform_attn_backward)
    assert list(f.parameters)[:6] == ["value", "value_spatial_shapes", "value_level_start_index", "sampling_locations", "attention_weights", "im2col_step"]
    assert list(b.parameters)[:7] == ["value", "value_spatial_shapes", "value_level_start_index", "sampling_locations", "attention_weights", "grad_output", "im2col_step"]
    with pytest.raises(Exception, match="HIP device|CPU|cpu"):     # no fallback: a CPU tensor is refused loudly
        v = torch.zeros(1, 4, 1, 4)
        MSDA.ms_deform_attn_forward(v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 1, 1, 1, 1, 2), torch.ones(1, 1, 1, 1, 1), 1)


@pytest.mark.skipif(not os.path.exists(REF_FUNC), reason="the reference tree exists in the build container only")
def test_reference_function_file_binds_to_the_module_unchanged():
    """Build container: the reference's own ms_deform_attn_func.py, executed as it is, imports `MultiScaleDeformableAttention` and finds
    this repository's module - its MSDeformAttnFunction then calls MSDA.ms_deform_attn_forward / _backward of libdgtd.so."""
    import MultiScaleDeformableAttention as MSDA
    ns = {"__name__": "ref_ms_deform_attn_func"}
    exec(compile(open(REF_FUNC).read(), REF_FUNC, "exec"), ns)
    assert ns["MSDA"] is MSDA
    fn = ns["MSDeformAttnFunction"]
    assert issubclass(fn, torch.autograd.Function)
    # the file's own pure-PyTorch core (ms_deform_attn_func.py:49-71) is what generated tests/golden/msda.npz: one case through it here
    name = next(iter(mc.CASES))
    (value, shp, loc, attn, _), y, *_ = _run_oracle(name)
    got = ns["ms_deform_attn_core_pytorch"](value.detach(), shp, loc.detach(), attn.detach())
    assert torch.allclose(got, y, rtol=1e-10, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mc.CASES)[:3])
def test_reference_shaped_function_over_the_module(G, name):
    """GPU box (no reference tree): a Function with the body of twig/ops/functions/ms_deform_attn_func.py:19-46 - MSDA.forward in
    forward(), MSDA.backward's 3-tuple in backward(), `None` for the integer inputs - over the module, against the reference vectors."""
    import MultiScaleDeformableAttention as MSDA
    from torch.autograd import Function
    from torch.autograd.function import once_differentiable

    class RefShaped(Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, attn, im2col_step):
            ctx.im2col_step = im2col_step
            out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, ctx.im2col_step)
            ctx.save_for_backward(value, shapes, lsi, loc, attn)
            return out

        @staticmethod
        @once_differentiable
        def backward(ctx, grad_output):
            value, shapes, lsi, loc, attn = ctx.saved_tensors
            gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output, ctx.im2col_step)
            return gv, None, None, gl, ga, None

    N, M, D, Lq, shapes, P = mc.CASES[name]
    value, shp, loc, attn, grad = mc.case_inputs(name, N, M, D, Lq, shapes, P, torch.float64)
    v, l_, a = (t.cuda().requires_grad_() for t in (value, loc, attn))
    out = RefShaped.apply(v, shp.cuda(), _level_start(shp).cuda(), l_, a, 2)
    hv, hl, ha = torch.autograd.grad(out, (v, l_, a), grad.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), G[f"{name}.out"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(hl.cpu().numpy(), G[f"{name}.grad_loc"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(ha.cpu().numpy(), G[f"{name}.grad_attn"], rtol=1e-8, atol=1e-11)
    if f"{name}.grad_value" in G:
        np.testing.assert_allclose(hv.cpu().numpy(), G[f"{name}.grad_value"], rtol=1e-8, atol=1e-11)
