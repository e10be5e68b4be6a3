#!/usr/bin/env python
"""Collect the per-op PMC passes made with tools/pmc_ops.py into one JSON: HBM bytes per launch of every dgtd kernel
(FETCH_SIZE doubled: gfx950 under-reports wide coalesced reads by exactly 2x, MI355X_MICROARCH.md; units KiB)."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
out = {}
for op in sorted(os.listdir(root)):
    per = collections.OrderedDict()
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(root, op, kind, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*|void ", "", r["Kernel_Name"])
            if not any(k in name for k in ("dwconv", "ln_", "sra_", "attn_delta", "colsum", "conv3x3", "multi_reduce", "scale_residual", "gemm_tn")):
                continue
            e = per.setdefault(name[:72] + " grid=" + r["Grid_Size"], {"fetch": [0.0, 0], "write": [0.0, 0]})
            e[kind][0] += float(r["Counter_Value"]) * 1024 * (2 if kind == "fetch" else 1)
            e[kind][1] += 1
    out[op] = {k: {"fetch_bytes": round(v["fetch"][0] / max(v["fetch"][1], 1)), "write_bytes": round(v["write"][0] / max(v["write"][1], 1)),
                   "launches": v["fetch"][1]} for k, v in per.items()}
print(json.dumps(out, indent=1))

# ---- second output (stderr -> file): HBM bytes per C-ABI call keyed like bench.py's kernel keys
BENCH_KEYS = {
    # the package's own GEMM at the ConvNeXt stage-3 shape (template arguments: <T, BN, EPI, NSTAGE>; EPI 1 = bias+GELU, 2 = bias+residual,
    # 3 = through-GELU input gradient, 0 = plain: here the input gradient of pwconv1)
    "dgtd_gemm_bias_gelu[M=8192,N=2048,K=512]": ("mlp_residual_8192x512", ["Li128ELi1E"]),
    "dgtd_gemm_bias_residual[M=8192,N=512,K=2048]": ("mlp_residual_8192x512", ["Li128ELi2E"]),
    "dgtd_gemm_gelu_bwd[M=8192,N=2048,K=512]": ("mlp_residual_8192x512", ["Li128ELi3E"]),
    "dgtd_gemm_bias[M=8192,N=512,K=2048]": ("mlp_residual_8192x512", ["Li128ELi0E"]),
    "dgtd_gelu_bias_bwd[rows=8192,C=2048]": ("linear_gelu_8192x512x2048", ["colsum2_kernel"]),
    "dgtd_scale_residual_bias_bwd[rows=8192,C=512]": ("linear_residual_8192x2048x512", ["colsum2_kernel"]),
    "dgtd_scale_residual_fwd[rows=8192,C=512]": ("linear_residual_8192x2048x512", ["scale_residual_fwd_kernel"]),
    "dgtd_dwconv_bwd_weight_batched[n27,k7,32x32x512]": ("dwconv_batched_k7_n27_32x32x512", ["dwconv_bww_sw_kernel", "dwconv_bww_batched_kernel"]),
    "dgtd_dwconv_bwd_weight[k7,32x32x512]": ("dwconv_k7_32x32x512", ["dwconv_bwd_weight_kernel", "dwconv_bww_reduce"]),
    "dgtd_dwconv_fwd[k7,mode0,32x32x512]": ("dwconv_k7_32x32x512", ["dwconv_tiled_fwd_kernel"]),
    "dgtd_dwconv_bwd_weight[k7,128x128x128]": ("dwconv_k7_128x128x128", ["dwconv_bwd_weight_kernel", "dwconv_bww_reduce"]),
    "dgtd_dwconv_fwd[k7,mode0,128x128x128]": ("dwconv_k7_128x128x128", ["dwconv_tiled_fwd_kernel"]),
    "dgtd_dwconv_bwd_weight[k3,128x128x512]": ("dwconv_k3_128x128x512", ["dwconv_bwd_weight_kernel", "dwconv_bww_reduce"]),
    "dgtd_dwconv_fwd[k3,mode0,128x128x512]": ("dwconv_k3_128x128x512", ["ELi3ELi4ELi0E"]),
    "dgtd_dwconv_fwd[k3,mode1,128x128x512]": ("dwconv_k3_128x128x512", ["ELi3ELi4ELi1E"]),
    "dgtd_dwconv_fwd[k3,mode2,128x128x512]": ("dwconv_k3_128x128x512", ["ELi3ELi4ELi2E"]),
    "dgtd_layernorm_fwd[rows=8192,C=512]": ("layernorm_8192x512", ["ln_fwd_kernel"]),
    "dgtd_layernorm_bwd[rows=8192,C=512]": ("layernorm_8192x512", ["ln_bwd_kernel", "ln_bwd_reduce"]),
    "dgtd_colsum[rows=8192,C=2048]": ("colsum_8192x2048", ["colsum_kernel", "colsum_reduce"]),
    "dgtd_conv3x3_wgrad[Z=1,64x64,96->96]": ("conv3x3_96_64x64", ["conv3x3_wgrad_kernel", "conv3x3_wgrad_reduce"]),
    "dgtd_conv3x3_fwd[Z=1,64x64,96->96]": ("conv3x3_96_64x64", ["conv3x3_fwd_kernel"]),
    "dgtd_conv3x3_wgrad[Z=16,128x128,24->24]": ("conv3x3_Z16_24_128x128", ["conv3x3_wgrad_kernel", "conv3x3_wgrad_reduce"]),
    "dgtd_conv3x3_fwd[Z=16,128x128,24->24]": ("conv3x3_Z16_24_128x128", ["conv3x3_fwd_kernel"]),
    "dgtd_sra_attn_fwd[B=8,N=16384,Nkv=256,h=1]": ("attention_stage1", ["sra_fwd_bf16"]),
    "dgtd_sra_attn_bwd[B=8,N=16384,Nkv=256,h=1]": ("attention_stage1", ["attn_delta_kernel", "sra_bwd_dkdv", "sra_bwd_dq"]),
}
by_key = {}
for key, (op, subs) in BENCH_KEYS.items():
    tot = 0
    for k, v in out.get(op, {}).items():
        if any(s_ in k for s_ in subs):
            tot += v["fetch_bytes"] + v["write_bytes"]
    if tot:
        by_key[key] = tot
if len(sys.argv) > 2:
    json.dump(by_key, open(sys.argv[2], "w"), indent=1)
