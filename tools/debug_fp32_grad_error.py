"""Debug: where does the fp32 HIP path lose gradient accuracy vs an fp64 oracle?  Run once per MIOpen setting (env)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgtd
from oracle import filler
g64 = torch.load(os.path.join(os.path.dirname(__file__), "_dbg_g64.pt"))
S, B = 32, 2
net = dgtd.nn.cod(drop_path_rate=0.0)
filler.fill_module(net)
net = net.cuda().train()
if os.environ.get("DBG_DET") == "1":
    torch.backends.cudnn.deterministic = True
x, d, l = filler.synthetic_batch(B, S, seed=0)
net(None, x.cuda(), l.cuda(), d.cuda(), mode="loss")["loss"].backward()
rows = []
for k, p in net.named_parameters():
    if k in g64 and p.grad is not None:
        t = g64[k]
        rows.append((float((p.grad.double().cpu() - t).norm() / t.norm()), k))
rows.sort(reverse=True)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("MIOPEN_", "DBG_", "DGTD_")))
print("==", tag or "default")
for e, k in rows[:12]:
    print(f"  {e:.3e} {k}")
import statistics
print("  median", statistics.median(e for e, _ in rows), "n", len(rows))
