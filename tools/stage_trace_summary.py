#!/usr/bin/env python
"""Per-step, per-family kernel table from a rocprofv3 --kernel-trace CSV of tools/prof_stage.py (steps delimited by the `sde_euler_kernel` marker).
Steps are grouped by kind: in the DMD stage a step with more than 1.5 x the median kernel count is a VAE turn, the rest are student-only steps.
    python tools/stage_trace_summary.py <kernel_trace.csv> [top]"""
import collections
import csv
import re
import sys

FAMILIES = [
    ("GEMM fwd/dgrad (gemm_pp + split-K sum)", r"gemm_pp_kernel|splitk_sum"),
    ("GEMM wgrad (wgrad_pp + reduce)", r"wgrad_pp_kernel|wgrad_pp_grouped|wgrad_grouped_bias|rows_wgrad_mfma|linear_rows_wgrad|wgrad_reduce|wgrad_kernel|wgrad_small|wgrad_thin|colsum_kernel|colsum_partial|colsum_final"),
    ("rows Linear (adaLN / embedders)", r"linear_rows"),
    ("conv fwd/dgrad (conv_pp etc.)", r"conv_pp_kernel|conv_fwd|conv_thin|conv_in3|conv_to_image|convout"),
    ("attention fwd", r"attention_kernel"),
    ("attention bwd", r"attention_bwd"),
    ("norm + modulate / LayerNorm (fwd)", r"rmsnorm_modulate_kernel|rmsnorm_modulate8_kernel|gated_residual_kernel|gated_norm|layernorm_kernel|scale_residual_kernel|qknorm_rope_kernel|qknorm_rope16_kernel|rmsnorm_rowstat"),
    ("norm + modulate / LayerNorm (bwd)", r"rmsnorm_modulate_bwd|gated_residual_bwd|layernorm_bwd|layerscale_bwd|qknorm_rope_bwd|qknorm_rope16_bwd|colsum2|colsum_parts|dit_.*bwd|rms_gate_bwd|rowstat_kernel|dit_bwd_"),
    ("GroupNorm (decoder)", r"groupnorm|gn_|apply_kernel|bwd_partial|bwd_final|stats_from|short_|coop_"),
    ("elementwise (swiglu / gelu / silu / casts of ours)", r"swiglu|gelu|silu|transpose_kernel|linear_wt_kmajor|bn_running|pack_|nchw_to_nhwc|nhwc_to_nchw|im2col|col2im|maxpool|sumpool|relu_bwd|subpixel|diffaug"),
    ("losses (l1 / lpips / dmd / kl)", r"l1_mse|lpips|dmd_|kl_|mmd_|scalar_sum"),
    ("optimiser (sumsq + adamw)", r"sumsq_partial|norm_final|adamw_ema"),
    ("ATen / other", r".*"),
]


def fam(name):
    for f, pat in FAMILIES:
        if re.search(pat, name):
            return f
    return "ATen / other"


def main(path, top=25):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "sde_euler_kernel" in r["Kernel_Name"]]
    steps = [rows[a + 1:b] for a, b in zip(marks, marks[1:]) if b > a + 1]
    counts = sorted(len(s) for s in steps)
    med = counts[len(counts) // 2]
    kinds = collections.defaultdict(list)
    for s in steps:
        kinds["vae_turn" if len(s) > 1.5 * med else "step"].append(s)
    if "vae_turn" not in kinds:
        kinds = {"step": steps}
    for kind, ss in kinds.items():
        n = len(ss)
        wall = sum(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in ss) / n / 1e6
        fa, ka = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
        for s in ss:
            for r in s:
                d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                for agg, key in ((fa, fam(r["Kernel_Name"])), (ka, r["Kernel_Name"])):
                    agg[key][0] += d
                    agg[key][1] += 1
        tot = sum(a[0] for a in fa.values()) / n / 1e6
        print(f"## {kind}: {n} step(s); wall {wall:.2f} ms/step, sum of kernel durations {tot:.2f} ms/step, {sum(len(s) for s in ss) / n:.0f} kernels/step")
        print("#  ms/step   share  calls/step  family")
        for k, (d, c) in sorted(fa.items(), key=lambda kv: -kv[1][0]):
            print(f"{d / n / 1e6:9.3f}  {d / n / 1e6 / tot:6.1%} {c / n:10.1f}  {k}")
        print("#  ms/step  calls/step     avg_us  kernel")
        for k, (d, c) in sorted(ka.items(), key=lambda kv: -kv[1][0])[:top]:
            print(f"{d / n / 1e6:9.3f} {c / n:10.1f} {d / c / 1e3:11.1f}  [{fam(k)[:14]}] {k[:150]}")
        print()


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:3]))
