#!/usr/bin/env python
"""Weight-gradient GEMMs of L identical layers: L separate launches (plain and split-K + sum, as the step does today) vs ONE
strided-batched launch over an arena [L, M, *] (deferred weight gradients)."""
import torch

dev = "cuda"


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, L, M, K, N in [("cnx s2 pw1", 27, 8192, 512, 2048), ("cnx s2 pw2", 27, 8192, 2048, 512), ("cnx s0 pw1", 3, 131072, 128, 512),
                         ("cnx s1 pw1", 3, 32768, 256, 1024), ("cnx s3 pw1", 3, 2048, 1024, 4096), ("pvt s3 fc1", 6, 8192, 320, 1280),
                         ("pvt s3 q", 6, 8192, 320, 320), ("pvt s1 fc1", 3, 131072, 64, 512), ("pvt s2 fc1", 4, 32768, 128, 1024)]:
    x = torch.randn(L, M, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(L, M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * L * M * N * K
    S = min(32, M // 1024)

    def separate():
        for i in range(L):
            dy[i].t() @ x[i]

    def split():
        for i in range(L):
            torch.bmm(dy[i].view(S, M // S, N).transpose(1, 2), x[i].view(S, M // S, K)).sum(0)

    t_sep, t_split = timed(separate), timed(split)
    t_b = timed(lambda: torch.bmm(dy.transpose(1, 2), x))
    print(f"{name:12s} L={L:2d} M={M:6d} K={K:4d} N={N:4d} | separate {t_sep:8.1f}us {fl / t_sep / 1e6:5.0f}TF | splitK{S} {t_split:8.1f}us {fl / t_split / 1e6:5.0f}TF"
          f" | batched {t_b:8.1f}us {fl / t_b / 1e6:5.0f}TF", flush=True)
