import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd
from oracle import filler
net = dgtd.nn.cod(drop_path_rate=0.0); filler.fill_module(net); net = net.cuda().eval()
x, d, l = (t.cuda() for t in filler.synthetic_batch(8, 512, seed=11))
outs = {}
def hook(name):
    def f(m, i, o):
        outs.setdefault(name, []).append(o)
    return f
bb = net.hitnet.backbone
mods = {"prompt_encoder": bb.prompt_encoder, "patch_embed1": bb.patch_embed1, "block1.0": bb.block1[0], "block1.0.attn": bb.block1[0].attn,
        "block1.0.mlp": bb.block1[0].mlp, "block1.2": bb.block1[2], "block2.3": bb.block2[3], "block3.5": bb.block3[5], "block4.2": bb.block4[2],
        "pe.encoder2": bb.prompt_encoder.encoder2, "dec_l1": net.hitnet.decoder_level1, "dec_l4": net.hitnet.decoder_level4, "SAM": net.hitnet.SAM,
        "T2_1": net.hitnet.Translayer2_1, "conv4": net.hitnet.conv4}
for k, m in mods.items(): m.register_forward_hook(hook(k))
def flat(o):
    if torch.is_tensor(o): return [o]
    r = []
    for e in o:
        if torch.is_tensor(e) or isinstance(e, (tuple, list)): r += flat(e)
    return r
with torch.no_grad():
    net.hitnet(x, d); net.hitnet(x, d)
for k, v in outs.items():
    n = len(v) // 2
    same = all(torch.equal(a, b) for a, b in zip(flat(v[0]), flat(v[n])))
    print(f"{k:16s} calls {len(v):3d} identical={same}")
