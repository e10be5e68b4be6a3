"""CPU: host logic of dist.GradReducer that needs no GPU - bucket layout (alignment padding), working-copy refresh on
load_state_dict, parameters without a gradient, collective ordering."""
import copy

import pytest
import torch

from oracle import filler


@pytest.fixture(scope="module")
def cod_with_reducer():
    import dgtd
    net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=torch.bfloat16)
    red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16)
    return dgtd, net, red


def test_every_bucket_slice_is_16_byte_aligned(cod_with_reducer):
    """1-element biases (out_CFM / out_SAM) used to shift every later tensor of their bucket to an odd byte phase.
    Every working copy, master, gradient view starts on a 16-byte boundary now, and the padding is zero everywhere."""
    dgtd, net, red = cod_with_reducer
    n = 0
    for b in red.buckets:
        assert b["flat"].data_ptr() % 16 == 0 and b["mflat"].data_ptr() % 16 == 0
        for leaf, master, gv, off, size, padded in zip(b["leaves"], b["masters"], b["gviews"], b["offsets"], b["sizes"], b["padded"]):
            assert leaf.data_ptr() % 16 == 0, "working copy / leaf"
            assert master.data_ptr() % 16 == 0 and gv.data_ptr() % 16 == 0
            assert off % dgtd.dist.reducer.ALIGN == 0 and padded % dgtd.dist.reducer.ALIGN == 0 and padded >= size
            if padded > size:
                assert float(b["mflat"][off + size:off + padded].abs().sum()) == 0.0
                n += 1
    assert n > 0                                           # the model does contain odd-sized tensors
    # the state_dict contract is untouched by the re-homing
    assert len(net.state_dict()) == 879


def test_load_state_dict_refreshes_working_copies(cod_with_reducer):
    """load_state_dict writes the fp32 masters; the modules compute with the working copies.  Loading through the root
    AND through a sub-module (runner.load_pretrained loads hitnet.backbone directly) must both refresh them."""
    dgtd, net, red = cod_with_reducer
    other = dgtd.nn.cod(drop_path_rate=0.0)
    filler.fill_module(other)
    q = net.hitnet.backbone.block1[0].attn.q
    before = q._w.detach().clone()
    net.load_state_dict(other.state_dict())
    assert not torch.equal(q._w, before)
    for m in (q, net.hitnet.decoder_level2[0].body[0], net.hitnet.backbone.prompt_encoder.encoder2.stages[2][5].pwconv1):
        torch.testing.assert_close(m._w.float(), m.weight.detach().bfloat16().float(), rtol=0, atol=0)
        if m.bias is not None:
            torch.testing.assert_close(m._b.float(), m.bias.detach().bfloat16().float(), rtol=0, atol=0)
    # sub-module load: only the backbone
    sd = {k: v * 0.5 for k, v in other.hitnet.backbone.state_dict().items()}
    net.hitnet.backbone.load_state_dict(sd)
    torch.testing.assert_close(q._w.float(), (other.hitnet.backbone.block1[0].attn.q.weight * 0.5).bfloat16().float(), rtol=0, atol=0)
    # masters stay views of the flat bucket (the optimizer updates them in place)
    b0 = red.buckets[0]
    assert b0["masters"][0].data_ptr() == b0["mflat"].data_ptr() + 4 * b0["offsets"][0]


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        import dgtd
        self.a = dgtd.nn.modules.Linear(8, 5)          # 40 + 5 elements: both need padding
        self.unused = dgtd.nn.modules.Linear(3, 3)
        self.c = dgtd.nn.modules.Conv2d(4, 6, 3, padding=1)
        self.n = torch.nn.LayerNorm(5)

    def forward(self, x, img):
        return self.n(self.a(x)).sum() + self.c(img).sum()


def test_missing_gradients_are_recorded_and_padded_layout_round_trips():
    import dgtd
    torch.manual_seed(0)
    net = _Net()
    ref = copy.deepcopy(net)
    red = dgtd.dist.GradReducer(net, exclude_prefixes=(), working_dtype=torch.bfloat16, bucket_bytes=1 << 30)
    assert len(red.buckets) == 1
    b = red.buckets[0]
    x, img = torch.randn(4, 8), torch.randn(2, 4, 5, 5)
    red.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        net(x, img).float().backward()
    red.finish()
    missing = {b["names"][i] for i in b["missing"]}
    assert missing == {"unused.weight", "unused.bias"}
    ref(x, img).backward()
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if k.startswith("unused"):
            assert float(p.grad.abs().sum()) == 0.0
        else:
            torch.testing.assert_close(p.grad, q.grad, rtol=5e-2, atol=5e-2, msg=lambda m: f"{k}: {m}")
    # element-wise: the KxK conv weight gradient lands in O,H,W,I storage and reads back in the logical O,I,H,W shape
    i = b["names"].index("c.weight")
    assert b["nhwc"][i]
    flat = b["flat"][b["offsets"][i]:b["offsets"][i] + b["sizes"][i]].view(6, 3, 3, 4).permute(0, 3, 1, 2)
    torch.testing.assert_close(flat, net.c.weight.grad, rtol=0, atol=0)
    torch.testing.assert_close(flat, ref.c.weight.grad, rtol=5e-2, atol=5e-2)
    # a second step with every parameter used clears the record
    red.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        (net(x, img) + net.unused(torch.randn(2, 3)).sum()).float().backward()
    red.finish()
    assert red.buckets[0]["missing"] == ()


def test_buckets_launch_in_index_order_whatever_the_hook_order():
    """World > 1 safety: bucket i is gathered/launched only after buckets < i, even when its gradients arrive first."""
    import dgtd
    net = torch.nn.Sequential(dgtd.nn.Linear(4, 4), dgtd.nn.Linear(4, 4), dgtd.nn.Linear(4, 4))
    red = dgtd.dist.GradReducer(net, exclude_prefixes=(), bucket_bytes=1)     # one bucket per parameter
    red.overlap = True
    order = []
    red._launch = lambda b: order.append(b["index"])
    red.zero_grad()
    # fire the hooks by hand in the WRONG order: last bucket first
    for b in reversed(red.buckets):
        for leaf in b["leaves"]:
            leaf.grad = torch.zeros_like(leaf)
            red._make_hook(b)(leaf)
    assert order == sorted(order) and len(order) == len(red.buckets)
    red.zero_grad()
    order.clear()
    b_last = red.buckets[-1]
    for leaf in b_last["leaves"]:
        leaf.grad = torch.zeros_like(leaf)
        red._make_hook(b_last)(leaf)
    assert order == []                                  # blocked behind bucket 0
    for b in red.buckets:
        for leaf in b["leaves"]:
            if leaf.grad is None:
                leaf.grad = torch.zeros_like(leaf)
    red.finish()
    assert order == list(range(len(red.buckets)))
