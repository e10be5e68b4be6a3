#!/usr/bin/env python
"""Library-GEMM shapes of the ConvNeXt/PVT linears at config 2 (bf16): forward, dgrad, wgrad, and split-K wgrad via bmm."""
import torch

dev = "cuda"


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, M, K, N in [("cnx s0 pw1", 131072, 128, 512), ("cnx s0 pw2", 131072, 512, 128), ("cnx s1 pw1", 32768, 256, 1024),
                      ("cnx s1 pw2", 32768, 1024, 256), ("cnx s2 pw1", 8192, 512, 2048), ("cnx s2 pw2", 8192, 2048, 512),
                      ("cnx s3 pw1", 2048, 1024, 4096), ("cnx s3 pw2", 2048, 4096, 1024), ("pvt s1 fc1", 131072, 64, 512),
                      ("pvt s1 fc2", 131072, 512, 64), ("pvt s2 fc1", 32768, 128, 1024), ("pvt s3 fc1", 8192, 320, 1280)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t_f = timed(lambda: torch.addmm(b, x, w.t()))
    t_d = timed(lambda: dy @ w)
    t_w = timed(lambda: dy.t() @ x)
    line = f"{name:12s} M={M:6d} K={K:4d} N={N:4d}  fwd {t_f:7.1f}us {fl / t_f / 1e6:6.0f}TF | dgrad {t_d:7.1f}us {fl / t_d / 1e6:6.0f}TF | wgrad {t_w:7.1f}us {fl / t_w / 1e6:6.0f}TF"
    for S in (4, 8, 16, 32):
        if M % S == 0 and M // S >= 512:
            t_s = timed(lambda: torch.bmm(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0))
            line += f" | splitK{S} {t_s:6.1f}us"
    print(line)
