#!/usr/bin/env python3
"""the CURRENT kernel table of the headline step, generated from a round's rocprofv3 kernel-stats
CSV and its PMC traffic summary (tools/collect_profiles.sh):
    python tools/kernel_table.py profiles/r06_v1_bench_kernel_stats.csv profiles/r06_v1_pmc_traffic.json [steps_in_trace=29]
One row per kernel above 0.15 % of the step: launches per step, average duration, measured HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), the rate that gives, the time per step."""
import csv
import json
import re
import sys

WHAT = {
    "sub_fwd_v5_kernel<3, true, 7": ("forward sub-block 2 / 3: BN + ReLU + dropout on load, depthwise K=3, pointwise MFMA GEMM, BN sums; keeps the depthwise output", "3 t", "VALU / LDS issue (producer waves) "),
    "sub_fwd_v5_kernel<3, true, 0": ("forward sub-block 1 (input stored activated)", "3 t", "same"),
    "sub_fwd_v5_kernel<3, true, 3": ("forward sub-block 1 of block 0 (raw prolog output: BN + ReLU on load)", "3 t", "same"),
    "sub_fwd_v4_kernel<1, false, 0": ("skip 1x1 conv of a mega block", "2 t", "HBM (cold stream)"),
    "sub_fwd_v4_kernel<1, false, 3": ("skip conv of block 0", "2 t", "HBM"),
    "se_combine_fwd_v3_kernel": ("SE squeeze + gate + residual combine, utterance resident in registers", "3 t", "VALU (two dropout hashes per element) / HBM latency"),
    "dgrad_dw_v6_kernel<7, false, true": ("last sub-block of a block: rebuilds dYbn from the tail's dZ, pointwise data gradient (MFMA) + transposed depthwise stencil + activation backward + BN-backward sums + depthwise weight gradients; stores dS", "5 t", "2 waves / SIMD latency chain (22 % s_waitcnt)"),
    "dgrad_dw_v6_kernel<7, false, false": ("middle sub-block: same without the rebuild", "5 t", "same"),
    "dgrad_dw_v6_kernel<8,": ("first sub-block: + the skip path's addend, no activation in front", "6 t", "same"),
    "dgrad_dw_v6_kernel<11,": ("first sub-block of block 0 (BN + ReLU of the prolog in front)", "6 t", "same"),
    "combine_bwd1_v3_kernel": ("mega-block tail backward in one pass (SE backward per utterance, dZ for skip and last sub-block)", "4 t", "HBM latency (per-utterance workgroups)"),
    "dgrad_v2_kernel<64>": ("skip-connection data gradient (+ stored dS of the skip conv)", "4 t", "HBM (at the cold-stream ceiling)"),
    "pgemm_tn_batched_kernel": ("67 pointwise weight gradients dW += dS^T Q as one pipelined TN contraction over stored operands", "67 x 2 t", "HBM (long launch)"),
    "wgrad_batched_v2_kernel": ("block 0's skip weight gradient, epilog slabs, pooling weights (slab units, split-K)", "-", "HBM"),
    "asp_v2_kernel<1>": ("attentive pooling backward without stored energies (recomputes the K = 128 product)", "-", "s_waitcnt (62 %): load / store coupling of the in-order vmcnt"),
    "asp_v2_kernel<0>": ("attentive pooling forward without stored energies", "-", "MFMA + exp VALU"),
    "gemm_nt_kernel<unsigned short, 2, 4, ProdDy": ("epilog conv data gradient (BN backward on load, K = 1536) on the generic tile GEMM", "-", "LDS-staged generic GEMM"),
    "wide_out_v2_kernel<128, 2>": ("attention data gradient + direct term through the epilog ReLU", "-", "HBM"),
    "wide_out_v2_kernel<256, 0>": ("epilog 1x1 conv 256 -> 1536 + BN sums", "-", "HBM write (store burst)"),
    "wide_in_v2_kernel<1>": ("attention hidden layer backward (K = 1536 -> 128)", "-", "HBM"),
    "wide_in_v2_kernel<0>": ("attention hidden layer forward", "-", "HBM"),
    "wgrad_kernel<unsigned short, ProdDy, ProdTaps>": ("prolog conv weight gradient", "-", "split-K TN GEMM"),
    "gemm_nt_kernel<unsigned short, 2, 4, ProdTaps": ("prolog k=3 conv on the packed rows x 80 operand", "-", "generic GEMM"),
    "adam_kernel": ("fused Adam over the flat buffer (6.2 M floats)", "-", "HBM, latency"),
}


def main():
    stats, pmc = sys.argv[1], json.load(open(sys.argv[2]))
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 29
    rows = []
    for r in csv.DictReader(open(stats)):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Name"]))
        if name.startswith(("at::", "__amd")):
            continue
        per, avg = int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3
        tr = pmc.get(name)
        mb = (2 * tr["FETCH_SIZE"] + tr["WRITE_SIZE"]) * 1024 / 1e6 if tr else None
        rows.append((avg * per, name, per, avg, mb))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"| kernel | what it does | passes (t = 39.3 MB) | launches / step | us / launch | HBM MB / launch (PMC) | TB/s | us / step | bound by |")
    print("|---|---|---|---|---|---|---|---|---|")
    rest = 0.0
    for t, name, per, avg, mb in rows:
        if t < 0.0015 * total:
            rest += t
            continue
        what = next((v for k, v in WHAT.items() if name.startswith(k)), ("", "-", ""))
        if mb:
            print(f"| `{name}` | {what[0]} | {what[1]} | {per:.0f} | {avg:.1f} | {mb:.0f} | {mb / avg:.2f} | {t:.0f} | {what[2]} |")      # MB / us = TB/s
        else:
            print(f"| `{name}` | {what[0]} | {what[1]} | {per:.0f} | {avg:.1f} | - | - | {t:.0f} | {what[2]} |")
    print(f"| (kernels below 0.15 % of the step) | | | | | | | {rest:.0f} | launch latency |")
    print(f"| **sum** | | | | | | | **{total:.0f}** | |")


if __name__ == "__main__":
    main()
