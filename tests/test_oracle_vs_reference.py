"""CPU, build container only: the oracle restatement against the LIVE reference
(/root/reference/twig/model/cod.py through oracle/ref_loader.py).  Skipped on the GPU box,
where the reference tree does not exist (the committed goldens cover it there)."""
import pytest
import torch

from oracle import cod_cpu, filler, ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present")


def test_keys_forward_backward_bit_exact():
    S = 32
    ref = ref_loader.build_reference_model(S, train=True)
    mine = cod_cpu.cod(S).train()
    sr, sm = ref.state_dict(), mine.state_dict()
    assert list(sr.keys()) == list(sm.keys()) or set(sr) == set(sm)
    assert all(sr[k].shape == sm[k].shape for k in sr)
    filler.fill_module(mine)
    ref.load_state_dict(mine.state_dict())
    x, d, l = filler.synthetic_batch(2, S)
    la = ref(None, x, l, list(d), mode="loss")["loss"]
    lb = mine(None, x, l, list(d), mode="loss")["loss"]
    assert abs(la.item() - lb.item()) < 1e-5
    la.backward(); lb.backward()
    gm = dict(mine.named_parameters())
    for k, p in ref.named_parameters():
        q = gm[k]
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is not None:
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), k
    ref.eval(); mine.eval()
    with torch.no_grad():
        a, b = ref.hitnet(x, d), mine.hitnet(x, d)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert all(torch.equal(p, q) for p, q in zip(a[1], b[1]))
