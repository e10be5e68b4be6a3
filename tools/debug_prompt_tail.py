"""Debug: gradient of the folded prompt-decoder tail with a channels_last 16-bit weight leaf vs the fp32 reference sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import dgtd
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for s_ in (1, 2, 4, 8):
        C = 128
        dec = dgtd.nn.ShapePropDecoder(C, 24).cuda()
        emb = torch.randn(2, 24, 32, 32, device="cuda")
        H = 32 // s_
        conv = dec.decoder[4]
        # reference: fp32 literal sequence on the conv output
        h = F.relu(dec.decoder[2](F.relu(dec.decoder[0](emb)))).detach()
        w32 = conv.weight.detach().clone().requires_grad_()
        ref = F.interpolate(F.conv2d(h, w32, conv.bias, padding=1), size=(H, H), mode="bilinear", align_corners=False)
        gout = torch.randn_like(ref)
        gw_ref, = torch.autograd.grad(ref, w32, gout)
        for cl in (False, True):
            w = conv.weight.detach().to(dtype)
            if cl:
                w = w.contiguous(memory_format=torch.channels_last)
            w = w.clone(memory_format=torch.preserve_format).requires_grad_()
            conv._w, conv._b = w, conv.bias.detach().to(dtype)
            hh = h.to(dtype).contiguous(memory_format=torch.channels_last)
            y = dec.forward_tokens(None, H, H, trunk=hh.permute(0, 2, 3, 1))
            g, = torch.autograd.grad(y, w, gout.flatten(2).transpose(1, 2).to(y.dtype))
            rel = float((g.float() - gw_ref).norm() / gw_ref.norm())
            print(f"dtype {dtype} scale {s_} channels_last_weight {cl}: rel err {rel:.4f} grad strides {g.stride()}")
            conv._w = conv._b = None
# avg_pool2d backward alone
w = torch.randn(128, 24, 3, 3, device="cuda")
for cl in (False, True):
    a = (w.contiguous(memory_format=torch.channels_last) if cl else w).clone(memory_format=torch.preserve_format).requires_grad_()
    o = F.avg_pool2d(a, 2, stride=1, padding=1)
    go = torch.randn(128, 24, 4, 4, device="cuda")
    g, = torch.autograd.grad(o, a, go)
    ac = w.cpu().clone().requires_grad_()
    gc, = torch.autograd.grad(F.avg_pool2d(ac, 2, stride=1, padding=1), ac, go.cpu())
    print("avg_pool2d bwd channels_last", cl, float((g.cpu() - gc).abs().max()))
