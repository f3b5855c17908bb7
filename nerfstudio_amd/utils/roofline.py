"""Roofline bookkeeping of the benchmark lines (bench.py): algorithmic work per launch, the live per-kernel
table (HIP events on the launch stream, `_native.PROFILE`), and the HBM-traffic figures of the committed PMC passes.

`achieved` is ALGORITHMIC work / measured launch time — never executed work: the forward recomputation inside
`nsamd_field_mlp_bwd` is reported separately (`executed_per_launch` / `executed_frac`)."""
from __future__ import annotations

import glob
import hashlib
import json
import os
import re

import torch

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(PKG)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-in MFMA = f32 vector peak
ROOFLINE_WARMUP_ITERS = 10  # untimed eager iterations in front of the profiled ones (measure_roofline); every rank runs them
FIELD_MACS = 11392  # MACs / sample of the main field: 32*64 + 64*16 + 63*64 + 64*64 + 64*3  


def algorithmic_model(key, main_points):
    """-> (bound, work per launch) of a profiled entry point; (None, None): not modelled."""
    m = re.search(r"L=(\d+),M=(\d+)", key)
    if key.startswith("nsamd_hashgrid_encode_fwd") and m:
        return "hbm", int(m.group(2)) * int(m.group(1)) * 8 * 8  # 8 corner gathers x 8 B (F=2 fp32) per level and sample
    if key.startswith("nsamd_hashgrid_encode_bwd") and m:
        return "hbm", int(m.group(2)) * int(m.group(1)) * 8 * 16  # read-modify-write of 8 corners x 8 B
    if key in ("nsamd_field_mlp_fwd", "nsamd_field_density_fwd"):
        return "mfma", main_points * 2 * FIELD_MACS
    if key == "nsamd_field_mlp_bwd_scatter_phase[apply]":
        # the table scatter whose ROUTE pass runs inside the field backward: the op's algorithmic figure is the
        # read-modify-write of 8 corners x 16 B per (sample, level); its two passes touch every corner update once each, so this
        # launch is credited with HALF of it — the other half is the `fused_route_pass` share of the gradient kernel's line.
        # (The 16-B records the passes hand over are implementation traffic and are counted nowhere.)
        return "hbm", main_points * 16 * 8 * 8
    if key in ("nsamd_field_mlp_bwd", "nsamd_field_mlp_bwd_scatter_phase[gradients+records]"):
        # training = 3x the forward FLOPs, the forward launch takes 1x, so the backward's ALGORITHMIC share is
        # 2x (data gradient + weight gradient); the recompute of the forward inside the kernel is executed, not algorithmic
        return "mfma", main_points * 2 * FIELD_MACS * 2
    m2 = re.search(r"\[M=(\d+)\]", key)
    if key.startswith("nsamd_density_mlp_fwd") and m2:
        return "hbm", int(m2.group(1)) * (10 * 4 + 4 + 8)  # enc row + selector in, density + pre out
    if key.startswith("nsamd_density_mlp_bwd") and m2:
        return "hbm", int(m2.group(1)) * (10 * 4 * 2 + 4 * 3)
    m3 = re.search(r"nsamd_adam_step\[n=(\d+)\]", key)
    if m3:
        return "hbm", int(m3.group(1)) * 28  # p, g, m, v read + p, m, v written
    return None, None


def step_algorithmic_bytes(rays, counts=(256, 96, 48), main_levels=16, prop_levels=5, params=0, updated_fraction=0.0):
    """SURVEY.md §8(d) whole-step HBM model of one nerfacto training iteration: hash gathers of every level (forward), the
    main table's scatter (read-modify-write), Adam over all parameters (28 B each), + the proposal tables' scatter on the
    fraction of iterations that update them."""
    fwd = sum(rays * s * prop_levels * 64 for s in counts[:-1]) + rays * counts[-1] * main_levels * 64
    bwd = rays * counts[-1] * main_levels * 128
    prop_bwd = sum(rays * s * prop_levels * 128 for s in counts[:-1])
    return fwd + bwd + params * 28 + updated_fraction * prop_bwd


# flops / bytes a launch actually executes where that differs from the algorithmic figure (reported next to it)
# Ray terms (include/nsamd.h, nsamd_field_mlp.ray_terms): head layer 0's 48 per-ray inputs are multiplied once per RAY, so a sample
# costs 32*64 + 64*16 + 15*64 + 64*64 + 64*3 = 8 320 MACs (+ 48*64 per ray = 64 per sample at 48 samples per ray) forward and in
# the data gradient, and the weight gradient's per-ray columns one 64 x 48 outer product (+ a 64 x 32 product for the appearance
# row) per 16-sample tile. The ALGORITHMIC figure stays SURVEY §8(d)'s dense 11 392.
FIELD_MACS_RAY_TERMS_FWD = 8320 + 64
FIELD_MACS_RAY_TERMS_BWD = (8320 + 64) + 8320 + (8320 + (64 * 48 + 64 * 32) // 16)  # recompute + data gradient + weight gradient


def executed_per_launch(key, main_points, ray_terms=False):
    if key in ("nsamd_field_mlp_bwd", "nsamd_field_mlp_bwd_scatter_phase[gradients+records]"):
        if ray_terms:
            return main_points * 2 * FIELD_MACS_RAY_TERMS_BWD
        return main_points * 2 * FIELD_MACS * 3  # + the forward recompute
    if key == "nsamd_field_mlp_fwd" and ray_terms:
        return main_points * 2 * FIELD_MACS_RAY_TERMS_FWD
    return None


# entry point -> the kernel name rocprofv3 --kernel-trace --stats lists for it (profiles/*_kernel_stats.csv)
ROCPROF_KERNEL = {
    "nsamd_field_mlp_bwd": "nsamd::field_mlp_bwd_kernel<false, RAYC>",
    "nsamd_field_mlp_bwd_scatter_phase[gradients+records]": "nsamd::field_mlp_bwd_kernel<true, true> (ray terms; <true, false> without)",
    "nsamd_field_mlp_bwd_scatter_phase[apply]": "nsamd::scatter_apply_kernel<true> (replayed graphs: the weight-gradient reduce rides it) "
                                                "+ nsamd::scatter_finish_kernel",
    "nsamd_field_mlp_fwd": "nsamd::field_mlp_fwd_kernel",
    "nsamd_hashgrid_encode_fwd": "nsamd::hash_encode_fwd_kernel",
    "nsamd_hashgrid_encode_bwd_set": "nsamd::scatter_route_fine_kernel + nsamd::scatter_apply_kernel + nsamd::scatter_finish_kernel",
    "nsamd_hashgrid_encode_bwd": "nsamd::scatter_route_* + nsamd::scatter_apply_kernel + nsamd::scatter_finish_kernel",
    "nsamd_hashgrid_encode_bwd_gated": "nsamd::scatter_route_* + nsamd::scatter_apply_kernel + nsamd::scatter_finish_kernel",
    "nsamd_adam_step": "nsamd::adam_kernel",
}


def kernel_sources_hash():
    """sha256 over the kernel sources: stamps profiles/pmc_traffic.json (scripts/collect_pmc.sh) so that a traffic
    figure measured on other kernels is never reported."""
    h = hashlib.sha256()
    base = os.path.join(PKG, "csrc")
    for path in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of `kernel_key` from the committed rocprofv3 PMC passes (scripts/collect_pmc.sh ->
    profiles/pmc_traffic.json: FETCH_SIZE, doubled for 16-B-per-lane streaming reads as MI355X_MICROARCH.md prescribes for
    gfx950, + WRITE_SIZE; separate --pmc passes). Counters cannot be read from inside this process, so the value is the
    one measured for this kernel by the PMC passes — and only if they ran on THESE kernel sources (the file carries
    their hash): None (JSON null) when the sources changed since, or the file has no entry for the kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        data = json.load(open(path))
        if data.get("_kernel_sources_sha256_16") != kernel_sources_hash():
            return None
        entry = data.get(kernel_key)
        return int(entry["hbm_bytes"]) if entry else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def profile_table(run_steps, steps):
    """Run `run_steps()` (which launches `steps` iterations eagerly on ONE stream) under the binding's HIP-event profiler.
    -> {entry point: (calls, total ms, mean ms)}."""
    from .. import _native as N

    N.PROFILE = {}
    run_steps()
    torch.cuda.synchronize()
    prof = N.profile_summary(N.PROFILE)
    N.PROFILE = None
    return prof


def roofline_entry(kernel, mean_ms, bound, work, main_points, ray_terms=False):
    sec = mean_ms * 1e-3
    if bound == "hbm":
        ach, peak, unit = work / sec / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        ach, peak, unit = work / sec / 1e12, F32_MFMA_PEAK_TFLOPS, "TFLOP/s"
    base = kernel.split("[")[0]
    rk = ROCPROF_KERNEL.get(kernel, ROCPROF_KERNEL.get(base))
    roof = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "traffic": pmc_traffic(kernel), "kernel": kernel, "avg_launch_ms": round(mean_ms, 4),
            "algorithmic_per_launch": int(work), "rocprof_kernel": rk}
    if kernel == "nsamd_field_mlp_bwd_scatter_phase[gradients+records]":
        # this launch also carries the table scatter's ROUTE pass: it derives and stores the x-pair records
        # — HBM work that `achieved` (flops only) does not credit; `roofline_min_ms` adds its streaming time at the HBM peak
        rec = main_points * 16 * 8 * 8  # its half of the scatter's algorithmic read-modify-write bytes (see the apply phase)
        roof["fused_route_pass"] = {"algorithmic_bytes": int(rec),
                                    "roofline_min_ms": round((work / (F32_MFMA_PEAK_TFLOPS * 1e12) + rec / (HBM_PEAK_GBS * 1e9)) * 1e3, 4),
                                    "frac_of_roofline_min": round((work / (F32_MFMA_PEAK_TFLOPS * 1e12) + rec / (HBM_PEAK_GBS * 1e9)) / sec, 4)}
    ex = executed_per_launch(kernel if "[" in kernel and kernel.startswith("nsamd_field_mlp_bwd_scatter_phase") else base, main_points,
                             ray_terms)
    if ex is not None:  # the utilisation view (work the launch executes, incl. recomputation)
        roof["executed_per_launch"] = ex
        roof["executed_frac"] = round(ex / sec / (1e9 if bound == "hbm" else 1e12) / peak, 4)
        if ray_terms:
            roof["ray_terms"] = ("head layer 0's 48 per-ray inputs (SH of the view direction, appearance row) are multiplied once per "
                                 "ray instead of once per sample: executed MACs per sample 8 384 of the dense layer stack's 11 392 "
                                 "that `achieved` is counted on")
    return roof


def measure_roofline(trainer, arena, steps, main_points):
    """Per-kernel table of `steps` untimed iterations of a trainer.HipTrainer (eager launches, one stream: a kernel's
    events must not include a concurrent branch's work) -> (roofline of the dominant kernel with the runner-up riding
    along, table rows sorted by time per step)."""
    graphs, trainer.graphs = trainer.graphs, None  # per-kernel events need eager launches
    runner = getattr(trainer, "runner", None)
    side = getattr(runner, "side_stream", None)
    if runner is not None:
        runner.side_stream = None
    trainer.opt_parallel = False

    def run():
        for _ in range(steps):
            trainer.train_iteration()
        trainer.finish()

    # untimed eager iterations first: the first eager launches after a stretch of graph replays and the state restore in front
    # of this call (a host-synchronous pause: the clocks drop) have been measured at 1.6 - 2.5 x the compute-bound kernel's time
    # (profiles/r06_s37_*: 0.41 ms, then 0.160 0.160 0.158 0.157; with ONE warm-up iteration still 0.28 0.25 0.160 0.157 0.157,
    # r06_s48_*; the memory- and latency-bound launches do not move) — a property of the hand-over, not of the kernel, that a
    # five-launch average must not carry
    for _ in range(ROOFLINE_WARMUP_ITERS):
        trainer.train_iteration()
    prof = profile_table(run, steps)
    trainer.graphs = graphs
    trainer.opt_parallel = True
    if runner is not None:
        runner.side_stream = side
    table = []
    for key, (calls, total_ms, mean_ms) in prof.items():
        bound, work = algorithmic_model(key, main_points)
        table.append({"kernel": key, "calls_per_step": calls / steps, "ms_per_step": total_ms / steps, "mean_ms": mean_ms,
                      "bound": bound, "work": work})
    table.sort(key=lambda r: -r["ms_per_step"])
    ranked = [r for r in table if r["bound"] is not None and r["work"]]
    rt = bool(getattr(runner, "ray_terms_on", False))
    roof = roofline_entry(ranked[0]["kernel"], ranked[0]["mean_ms"], ranked[0]["bound"], ranked[0]["work"], main_points, rt) if ranked else None
    # The main-field MLP backward and the main-table scatter are within a few percent of each other per step: which one is
    # "the dominant kernel" flips between runs. The runner-up rides along so that both are in every line.
    if roof is not None and len(ranked) > 1:
        r1 = ranked[1]
        roof["runner_up"] = roofline_entry(r1["kernel"], r1["mean_ms"], r1["bound"], r1["work"], main_points, rt)
    return roof, table


def algorithmic_model_ngp(key, kept_samples):
    """Algorithmic work of the field kernels on the packed samples (M = kept samples of the step, SURVEY.md §8d per-sample
    figures)."""
    M = float(kept_samples)
    base = key.split("[")[0]
    if base == "nsamd_hashgrid_encode_fwd" and "L=16" in key:
        return "hbm", float(re.search(r"M=(\d+)", key).group(1)) * 16 * 8 * 8
    if base in ("nsamd_hashgrid_encode_bwd", "nsamd_hashgrid_encode_bwd_set") and "L=16" in key:
        return "hbm", M * 16 * 8 * 16
    if base == "nsamd_field_mlp_fwd":
        return "mfma", M * 2 * FIELD_MACS
    if base == "nsamd_field_mlp_bwd":
        return "mfma", M * 2 * FIELD_MACS * 2
    return None
