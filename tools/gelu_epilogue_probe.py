import torch, torch.nn.functional as F
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(512, 256, device="cuda", dtype=dt); w = torch.randn(384, 256, device="cuda", dtype=dt) / 16; b = torch.randn(384, device="cuda", dtype=dt)
    try:
        y = torch._addmm_activation(b, x, w.t(), use_gelu=True).float()
    except Exception as e:
        print(dt, "failed", e); continue
    pre = torch.addmm(b.float(), x.float(), w.float().t())
    print(dt, "vs erf", (y - F.gelu(pre)).abs().max().item(), "vs tanh", (y - F.gelu(pre, approximate="tanh")).abs().max().item())
