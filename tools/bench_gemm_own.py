"""csrc/gemm.hip against the library GEMM (hipBLASLt through torch) at the model's Linear shapes (config 2: 512x512, batch 8):
correctness vs an fp32 reference and device time (HIP events over back-to-back launches, random operands, interleaved rounds).
    python tools/bench_gemm_own.py [--rounds 5] [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

L = dgtd._lib

# (name, M, N, K): forward x[M,K] W[N,K]^T
SHAPES = [
    ("cnx2.pwconv1", 8192, 2048, 512), ("cnx2.pwconv2", 8192, 512, 2048),
    ("cnx0.pwconv1", 131072, 512, 128), ("cnx0.pwconv2", 131072, 128, 512),
    ("cnx1.pwconv1", 32768, 1024, 256), ("cnx1.pwconv2", 32768, 256, 1024),
    ("cnx3.pwconv1", 2048, 4096, 1024), ("cnx3.pwconv2", 2048, 1024, 4096),
    ("pvt1.fc1", 131072, 512, 64), ("pvt1.fc2", 131072, 64, 512), ("pvt1.q", 131072, 64, 64),
    ("pvt2.fc1", 32768, 1024, 128), ("pvt2.fc2", 32768, 128, 1024), ("pvt2.q", 32768, 128, 128),
    ("pvt3.fc1", 8192, 1280, 320), ("pvt3.fc2", 8192, 320, 1280), ("pvt3.q", 8192, 320, 320), ("pvt3.kv", 2048, 640, 320),
    ("sq4096", 4096, 4096, 4096), ("sq8192k1024", 8192, 8192, 1024),
    ("pvt4.fc1", 2048, 2048, 512), ("pvt4.fc2", 2048, 512, 2048), ("pvt4.q", 2048, 512, 512), ("pvt4.kv", 2048, 1024, 512),
]


def timeit(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only-shapes", action="store_true")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    code = L.dtype_code(torch.empty(1, dtype=dt))
    st = L.stream_ptr()
    dev = "cuda"
    print(f"{'shape':14s} {'M':>7s} {'N':>5s} {'K':>5s} | {'own us':>8s} {'TF/s':>6s} | {'lib us':>8s} {'TF/s':>6s} | ratio | max err (own, lib) vs fp32")
    tot_own = tot_lib = 0.0
    for name, M, N, K in SHAPES:
        torch.manual_seed(0)
        x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
        w = (torch.randn(N, K, device=dev) * (K ** -0.5)).to(dt)
        b = (torch.randn(N, device=dev) * 0.1).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        assert L.load().dgtd_gemm_supported(M, N, K, code), (name, M, N, K)
        own = lambda: L.call("dgtd_gemm_bias", x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, code, st)
        lib_out = torch.empty_like(out)
        lib = lambda: torch.addmm(b, x, w.t(), out=lib_out)
        own(); lib()
        ref = (x[:2048].float() @ w.float().t() + b.float())
        e_own = float((out[:2048].float() - ref).abs().max())
        e_lib = float((lib_out[:2048].float() - ref).abs().max())
        t_own, t_lib = [], []
        for _ in range(args.rounds):
            t_own.append(timeit(own, args.iters))
            t_lib.append(timeit(lib, args.iters))
        to, tl = sorted(t_own)[len(t_own) // 2], sorted(t_lib)[len(t_lib) // 2]
        fl = 2.0 * M * N * K
        tot_own += to
        tot_lib += tl
        print(f"{name:14s} {M:7d} {N:5d} {K:5d} | {to:8.1f} {fl / to / 1e6:6.0f} | {tl:8.1f} {fl / tl / 1e6:6.0f} | {tl / to:5.2f} | {e_own:.3e} {e_lib:.3e}")
    print(f"sum: own {tot_own:.0f} us, library {tot_lib:.0f} us")
    if args.only_shapes:
        return

    # ---- fused epilogues at the dominant ConvNeXt stage-2 shape
    M, C4, Cc = 8192, 2048, 512
    torch.manual_seed(1)
    x = (torch.randn(M, Cc, device=dev) * 0.5).to(dt)
    w1 = (torch.randn(C4, Cc, device=dev) * Cc ** -0.5).to(dt)
    b1 = (torch.randn(C4, device=dev) * 0.1).to(dt)
    w2 = (torch.randn(Cc, C4, device=dev) * C4 ** -0.5).to(dt)
    b2 = (torch.randn(Cc, device=dev) * 0.1).to(dt)
    res = (torch.randn(M, Cc, device=dev)).to(dt)
    gamma = torch.rand(Cc, device=dev) + 0.5
    s = (torch.rand(8, device=dev) < 0.8).float() / 0.8
    pre, h = torch.empty(M, C4, device=dev, dtype=dt), torch.empty(M, C4, device=dev, dtype=dt)
    y, o = torch.empty(M, Cc, device=dev, dtype=dt), torch.empty(M, Cc, device=dev, dtype=dt)
    f_gelu = lambda: L.call("dgtd_gemm_bias_gelu", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), pre.data_ptr(), h.data_ptr(), M, C4, Cc, code, st)
    f_res = lambda: L.call("dgtd_gemm_bias_residual", h.data_ptr(), w2.data_ptr(), b2.data_ptr(), res.data_ptr(), s.data_ptr(), gamma.data_ptr(),
                           y.data_ptr(), o.data_ptr(), M, Cc, C4, M // 8, code, st)
    f_gelu(); f_res()
    pre_ref = x.float() @ w1.float().t() + b1.float()
    print("gelu: pre err %.3e, h err %.3e" % (float((pre.float() - pre_ref).abs().max()), float((h.float() - F.gelu(pre.float())).abs().max())))
    y_ref = h.float() @ w2.float().t() + b2.float()
    o_ref = res.float() + s.repeat_interleave(M // 8)[:, None] * gamma[None, :] * y.float()
    print("residual: y err %.3e, out err %.3e" % (float((y.float() - y_ref).abs().max()), float((o.float() - o_ref).abs().max())))
    # backward through the GELU: dpre = (dy W2) * gelu'(pre), colsum partials
    dy = (torch.randn(M, Cc, device=dev) * 0.1).to(dt)
    w2t = w2.t().contiguous()                     # [C4, Cc]: K-contiguous in the reduction dim
    dpre = torch.empty(M, C4, device=dev, dtype=dt)
    ws = torch.empty(L.load().dgtd_gemm_gelu_bwd_workspace(M, C4) // 4, device=dev, dtype=torch.float32)
    nb = C.c_int(0)
    f_bwd = lambda: L.call("dgtd_gemm_gelu_bwd", dy.data_ptr(), w2t.data_ptr(), pre.data_ptr(), dpre.data_ptr(), ws.data_ptr(), C.byref(nb), M, C4, Cc, code, st)
    f_bwd()
    p32 = pre.float().requires_grad_()
    F.gelu(p32).backward(dy.float() @ w2.float())
    print("gelu_bwd: dpre err %.3e (max |dpre| %.3e), colsum err %.3e" % (
        float((dpre.float() - p32.grad).abs().max()), float(p32.grad.abs().max()),
        float((ws.view(nb.value, C4).sum(0) - dpre.float().sum(0)).abs().max())))
    # transpose
    wt2 = torch.empty(C4, Cc, device=dev, dtype=dt)
    P, I = C.c_void_p * 1, C.c_int * 1
    L.call("dgtd_transpose_batched", P(w2.data_ptr()), P(wt2.data_ptr()), I(Cc), I(C4), 1, code, st)
    print("transpose exact:", bool(torch.equal(wt2, w2t)))
    lib_chain = {
        "pwconv1+GELU (lib: addmm + gelu)": (f_gelu, lambda: F.gelu(torch.addmm(b1, x, w1.t()))),
        "pwconv2+residual (lib: addmm + scale-residual torch ops)": (f_res, lambda: res + (torch.addmm(b2, h, w2.t()).float() * gamma).to(dt)),
        "dX2 through GELU (lib: mm + gelu_backward + colsum)": (f_bwd, lambda: (torch.ops.aten.gelu_backward(torch.mm(dy, w2), pre)).sum(0)),
    }
    for k, (a, b_) in lib_chain.items():
        ta = sorted(timeit(a, args.iters) for _ in range(args.rounds))[args.rounds // 2]
        tb = sorted(timeit(b_, args.iters) for _ in range(args.rounds))[args.rounds // 2]
        print(f"{k}: own {ta:.1f} us, library chain {tb:.1f} us")


if __name__ == "__main__":
    main()
