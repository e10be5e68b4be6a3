"""CPU: host-side algebra of the drop-in modules that needs no GPU."""
import pytest
import torch
import torch.nn.functional as F

from oracle import cod_cpu, filler


@pytest.mark.parametrize("scale", [1, 2, 4, 8])
def test_prompt_tail_conv_resize_fusion_matches_reference_sequence(scale):
    """ShapePropDecoder.forward_tokens == bilinear(conv3x3(.)) -> tokens (twig/model/cod.py:1224-1226 + :1471) for the
    four pyramid scales: the 2x2-centre bilinear average folded into one strided 4x4 convolution."""
    import dgtd
    C = 40
    dec = dgtd.nn.ShapePropDecoder(C, 24)
    ref = cod_cpu.ShapePropDecoder(C)
    filler.fill_module(ref, "hitnet.backbone.prompt_decoder.1.decoder.0.")
    dec.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(scale)
    emb = torch.randn(2, 24, 32, 32, generator=g)
    H = 32 // scale
    with torch.no_grad():
        want = F.interpolate(ref(emb), size=(H, H), mode="bilinear", align_corners=False).flatten(2).transpose(1, 2)
        got = dec.forward_tokens(emb.contiguous(memory_format=torch.channels_last), H, H)
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # gradients reach the 3x3 weights through the folded kernel
    emb.requires_grad_()
    dec.forward_tokens(emb, H, H).square().sum().backward()
    gw = dec.decoder[4].weight.grad.clone()
    ref.zero_grad()
    F.interpolate(ref(emb), size=(H, H), mode="bilinear", align_corners=False).square().sum().backward()
    torch.testing.assert_close(gw, ref.decoder[4].weight.grad, rtol=1e-4, atol=1e-4)


def test_tokens_nchw_views_are_free():
    import dgtd
    from dgtd.nn.modules import _nchw_to_tokens, _tokens_to_nchw
    t = torch.randn(2, 12, 8)
    x = _tokens_to_nchw(t, 3, 4)
    assert x.shape == (2, 8, 3, 4) and x.is_contiguous(memory_format=torch.channels_last)
    assert x.data_ptr() == t.data_ptr()
    back = _nchw_to_tokens(x)
    assert back.data_ptr() == t.data_ptr() and torch.equal(back, t)


def test_drop_path_plan_and_state_dict_contract():
    import dgtd
    net = dgtd.nn.cod()
    ref_keys = set(cod_cpu.cod(64).state_dict())
    assert set(net.state_dict()) == ref_keys                      # working copies / plans never leak into the checkpoint
    assert len(net._dp_layers) == 50                              # 15 PVT + 35 ConvNeXt DropPaths
    net.train()
    net._draw_drop_path(4)
    m = net._dp_plan["masks"]
    assert m.shape == (50, 4)
    keep = net._dp_keep
    assert torch.all((m == 0) | torch.isclose(m, (1.0 / keep).expand_as(m)))
    net.eval()
    net._draw_drop_path(4)
    assert net._dp_plan["masks"] is None


def test_fold_2x2_mean_matches_avg_pool_and_its_gradient():
    """The [16,9] tap matrix == F.avg_pool2d(w, 2, 1, 1) (value and gradient; CPU, where avg_pool2d's backward is right), for a
    contiguous and a channels_last weight."""
    from dgtd.nn.modules import _fold_2x2_mean
    w = torch.randn(10, 24, 3, 3)
    g = torch.randn(10, 24, 4, 4)
    for cl in (False, True):
        a = (w.contiguous(memory_format=torch.channels_last) if cl else w).clone(memory_format=torch.preserve_format).requires_grad_()
        b = w.clone().requires_grad_()
        ya, yb = _fold_2x2_mean(a), F.avg_pool2d(b, 2, stride=1, padding=1)
        torch.testing.assert_close(ya, yb, rtol=1e-6, atol=1e-6)
        ga, = torch.autograd.grad(ya, a, g)
        gb, = torch.autograd.grad(yb, b, g)
        torch.testing.assert_close(ga, gb, rtol=1e-6, atol=1e-6)


def test_block_run_is_a_no_op_without_the_native_layer():
    """ops.block_run (arena hints for the deferred weight-gradient GEMMs) must not touch anything on CPU tensors / without the C++
    binding layer: modules call it unconditionally."""
    import torch
    import dgtd
    from dgtd import ops

    class Owner(torch.nn.Module):
        pass

    owner, x = Owner(), torch.zeros(2, 4)
    with ops.block_run(owner, 0, 3, x) as run:
        for j in range(3):
            run.at(j)
            run.roles(out=0, grad_a=1, grad_b=2)
    assert "_dgtd_arena_token" not in owner.__dict__
    ops.NO_RUN.at(0)
    ops.NO_RUN.roles(out=1)
