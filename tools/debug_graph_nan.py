"""Debug: which part of _stage (after a host sync) corrupts the next replay?"""
import os, sys
os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgtd

def run(mode, S=512, B=8, dpr=0.1, dtype=torch.bfloat16):
    torch.manual_seed(0)
    net = dgtd.nn.cod(drop_path_rate=dpr, compute_dtype=dtype).cuda().train()
    red = dgtd.dist.GradReducer(net, working_dtype=dtype)
    opt = dgtd.runner.FlatAdamW(red, graph_safe=True)
    data = dgtd.runner.SyntheticRGBD(S, B, device="cuda")
    bs = [data.batch_at(i) for i in range(2)]
    st = dgtd.runner.GraphedTrainStep(net, red, opt, warmup=2)
    st.capture(bs[0])
    pool = torch.cuda.MemPool() if mode == "own_pool" else None
    dummy = {k: v.clone() for k, v in st.static.items()}
    losses = []
    for i in range(5):
        b = bs[i % 2]
        s = st.static
        if mode == "copies_only":
            for k in ("input", "label", "depth"):
                s[k].copy_(torch.stack(list(b[k])))
        elif mode == "fft_only":
            s["x_hp"].copy_(net.high_pass(s["input"]))
        elif mode == "into_dummy":
            for k in ("input", "label", "depth"):
                dummy[k].copy_(torch.stack(list(b[k])))
            dummy["x_hp"].copy_(net.high_pass(dummy["input"]))
        elif mode == "own_pool":
            with torch.cuda.use_mem_pool(pool):
                st._stage(b)
        elif mode == "alloc_only":
            t = [torch.empty(64 << 20, dtype=torch.uint8, device="cuda").fill_(7) for _ in range(6)]
            del t
        st.opt.sync_lr(); st.graph_fb.replay()
        torch.cuda.synchronize()
        losses.append(round(st.loss.item(), 4))
    print(f"mode {mode}: {losses}", flush=True)
    del net, red, opt, st
    torch.cuda.empty_cache()

for mode in ("copies_only", "fft_only", "into_dummy", "alloc_only", "own_pool"):
    run(mode)
