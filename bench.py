#!/usr/bin/env python3
"""G+D train-step throughput of the MI355X-native SP-GAN hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 launches its own N ranks (one process per GPU, re-executing this file under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`); started under torchrun
(RANK/WORLD_SIZE in the environment) it is a rank of that job.  Rank 0 prints ONE JSON line.

A "step" is one iteration of the reference loop body (Generation/model.py:239-279): D-step + G-step
with both Adam updates (and, for N>1, both flat gradient all-reduces), on BASELINE config 2:
Chair-shaped synthetic clouds, 2048 points, per-GPU batch 32, WGAN loss + gradient penalty (lambda 10),
fp32.  Weak scaling: the per-GPU batch is fixed, global batch = 32*N.  Inputs are resident in HBM.

The JSON line also carries
  roofline      : the dominant kernel (the fp32-MFMA gemm_nt), its algorithmic FLOPs per launch / average
                  launch time measured with HIP events on the launch stream, against the 157.3 TFLOP/s fp32 matrix peak;
  mfma          : FLOPs the step actually issues on the matrix cores (our formulation, padded tiles), the summed duration of
                  those kernels (HIP events around every launch, GPU kept ahead of the host) -> time-weighted MFMA utilisation;
  drop_in_caller: the same step the way an unmodified reference loop would drive it (latent tiled to [B,N,nz], the unused
                  D(real) forward of the G step evaluated, EdgeConv1 on every copy of the tiled sphere);
  cpu_baseline  : the CPU oracle (a PyTorch restatement of the reference, kind "port") timed on this box's host
                  cores on a bounded sample of the same workload (batch 32 = the metric's own batch, and batch 4).
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch   # noqa: E402

# SPGAN_BENCH_SELFTEST=1: plumbing self-test on the CPU (tests/test_bench_cpu.py): tiny shapes, gloo, spgan.ops replaced by the
# test doubles of tests/kernel_model.py.  Exercises argument handling, the self-launch, the rendezvous, the data-parallel step and
# the JSON line; its numbers are meaningless and the line says so.
SELFTEST = os.environ.get("SPGAN_BENCH_SELFTEST", "0") == "1"
# --config {c2,c4,c5} (read before argparse: module-level code sizes things with it).  c2 = BASELINE configs[1] (N=2048, per-GPU batch 32,
# fp32: the headline, the default and what the driver runs); c4 = the per-GPU shape of configs[3] (N=4096, per-GPU batch 16: its
# 8-GPU global batch 128 is reached with --gpus 8); c5 = the per-GPU shape of configs[4] (N=2048, per-GPU batch 32, fp16 MFMA operands
# = `--mfma f16`).
CONFIG = next((sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--config"), "c2")
CONFIG = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--config=")), CONFIG)
if CONFIG not in ("c2", "c4", "c5"):
    raise SystemExit("--config must be one of c2, c4, c5")
N_POINTS = 128 if SELFTEST else (4096 if CONFIG == "c4" else 2048)
# 32 = BASELINE configs[1].  `--experiment-batch B` (experiments only; the line is then marked "experiment") is read before argparse because
# module-level code sizes things with it; the old SPGAN_BENCH_BATCH environment override is refused.
if os.environ.get("SPGAN_BENCH_BATCH"):
    raise SystemExit("SPGAN_BENCH_BATCH is no longer honoured: use `--experiment-batch B` (the JSON line is then marked as an experiment)")
EXPERIMENT_BATCH = next((int(sys.argv[i + 1]) for i, a in enumerate(sys.argv[:-1]) if a == "--experiment-batch"), None)
CONFIG_BATCH = 16 if CONFIG == "c4" else 32
PER_GPU_BATCH = 2 if SELFTEST else (EXPERIMENT_BATCH or CONFIG_BATCH)
NZ = 128
K_NN = 10
FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
FP16_MATRIX_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (never the 2:1-sparsity figure)
# "bf16x3": one fp32-equivalent product = six bf16 MFMA cross products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) -> the ceiling of
# that route in fp32-equivalent FLOPs is the dense bf16 peak / 6
SPLIT_BF16_PEAK_TFLOPS = FP16_MATRIX_PEAK_TFLOPS / 6.0
SPLIT_BF16_PEAK_NOTE = "2500 TFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md) / 6 bf16 MFMAs per fp32-equivalent product = 416.7 TFLOP/s fp32-equivalent"
PEAK_OF_MODE = {"f32": FP32_MATRIX_PEAK_TFLOPS, "f16": FP16_MATRIX_PEAK_TFLOPS, "bf16x3": SPLIT_BF16_PEAK_TFLOPS}
DTYPE_OF_MODE = {"f32": "f32", "f16": "f16 MFMA operands, f32 accumulate/epilogues/weight-gradients",
                 "bf16x3": "f32 storage and accumulation; the large gemm_nt products and the weight gradients with 256/128-tileable outputs with every f32 "
                           "operand split exactly into 3 bf16 terms (6 bf16 MFMA cross products: f32-equivalent, dropped terms <= 2^-26 relative); "
                           "the fused layer-backward launches (gemm_dual: D's 128- and 64-channel layers, EdgeConv2) and the remaining small products on the exact f32 MFMA"}
# algorithmic FLOPs per shape per step, reference formulation: WGAN-GP at N=2048
GF_PER_SHAPE_STEP = 67.3 if CONFIG == "c4" else 32.6     # 2 F_Gf + 4N*779,520 + 15 F_Df at N = 4096 / 2048


def _profiles_newest_first(suffix):
    """profiles/rNN[a]_<suffix>, newest round first (so a new round's collection is picked up without editing this file)."""
    import glob
    names = [os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_" + suffix))]
    names = [n for n in names if re.fullmatch(r"r\d\d[a-z]?_" + re.escape(suffix), n)]
    return tuple(sorted(names, key=lambda n: (n[1:3], n[3] == "_"), reverse=True))     # r06_ before r06a_ before r05_ ...


PMC_FILES = _profiles_newest_first("pmc_gemm_nt.json")
# whole-step PMC passes per bench mode: (config, mfma) -> file suffix (tools/pmc_step.py --config / --mfma, tools/collect_profiles.sh)
PMC_STEP_SUFFIX = {("c2", "f32"): "pmc_step.json", ("c4", "f32"): "pmc_step_c4.json", ("c2", "f16"): "pmc_step_f16.json",
                   ("c2", "bf16x3"): "pmc_step_bf16x3.json"}
ALGORITHMIC_HBM_GB_PER_STEP = 3.5      # ~110 MB per shape per step x 32 shapes


class Opts:
    np = N_POINTS; nk = 2 * K_NN; nz = NZ; softmax = True; off = False; attn = False
    use_head = False; eql = False; z_norm = False; small_d = False


def build_models(dev, variant=()):
    import spgan
    torch.manual_seed(123)                      # Generation/model.py:38-41
    O = type("O", (Opts,), {f: True for f in variant})
    G, D = spgan.Generator(O), spgan.Discriminator(O)
    if "attn" in variant:
        with torch.no_grad():
            G.attn.gamma.fill_(0.1)             # the gate is 0 at initialisation (modules.py:546); timing is value-independent
    return G.to(dev), D.to(dev)


def make_inputs(dev, rank, b, tiled_z=False):
    from spgan import fixture_rng as fr
    n_t = N_POINTS if N_POINTS in (512, 1024, 2048, 4096) else 256
    x = fr.sphere_template(n_t)[:N_POINTS][None].repeat(b, 1, 1).to(dev)
    real = fr.synthetic_real(b, N_POINTS, seed=1234 + rank).to(dev)
    # one latent per shape (the reference's default noise_generator, model.py:128-131); passed un-tiled [b,1,nz] -- spgan.Generator
    # evaluates the latent half of head.0 per shape instead of tiling it over the 2048 points first.  tiled_z: [b,N,nz] as the
    # reference's loop hands it over.
    zs = [fr.latent(b, N_POINTS, NZ, seed=4321 + rank + i) for i in range(2 if tiled_z else 4)]
    zs = [(z if tiled_z else z[:, :1, :]).contiguous().to(dev) for z in zs]
    alpha = fr.uniform("bench.alpha.%d" % rank, (b, 1, 1), 0.0, 1.0).to(dev)
    return x, real, zs, alpha


def _pmc_step_traffic(config="c2", mfma="f32"):
    """Whole-step HBM bytes from the committed rocprofv3 PMC passes of tools/pmc_step.py (FETCH_SIZE doubled + WRITE_SIZE) for
    this bench mode; None when no pass of this mode is committed."""
    suffix = PMC_STEP_SUFFIX.get((config, mfma))
    for name in (_profiles_newest_first(suffix) if suffix else ()):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            gb = d.get("hbm_bytes_per_step", {}).get("total_GB")
            if gb:
                return {"hbm_gb_per_step": round(float(gb), 2), "algorithmic_gb_per_step": ALGORITHMIC_HBM_GB_PER_STEP,
                        "ratio": round(float(gb) / ALGORITHMIC_HBM_GB_PER_STEP, 2), "source": "profiles/" + name,
                        "note": "committed rocprofv3 --pmc passes over a replayed step of this mode (not sampled in this run)"}
        except Exception:
            continue
    return None


ROCPROF_STATS_FILES = _profiles_newest_first("bench_kernel_stats.txt")


def _rocprof_reference(M):
    """The dominant kernel's per-grid average durations from the COMMITTED rocprofv3 --kernel-trace summary of `python bench.py`
    (profiles/r0N_bench_kernel_stats.txt, tools/collect_profiles.sh; grid = 1024 / 3072 workgroups x 512 threads for the one-pass /
    three-pass launch at C2): sum FLOPs / sum time over the two launches of a step -> the fraction this run's `frac` must reproduce."""
    import re
    want = {(M // 256) * (1024 // 256) * 512: M, (3 * M // 256) * (1024 // 256) * 512: 3 * M}
    for name in ROCPROF_STATS_FILES:
        try:
            txt = open(os.path.join(ROOT, "profiles", name)).read()
        except OSError:
            continue
        sec = txt.split("# per-grid breakdown")[-1]
        blk = re.search(r"gemm_nt_wide_kernel<1, 0, 0>[^\n]*\n((?:\s+grid_x[^\n]*\n)+)", sec)
        if not blk:
            continue
        rows = {int(g): float(a) for g, a in re.findall(r"grid_x\s+(\d+)\s+calls\s+\d+\s+total_us\s+[\d.]+\s+avg_us\s+([\d.]+)", blk.group(1))}
        got = {m: rows[g] for g, m in want.items() if g in rows}
        if len(got) != 2:
            continue
        fl = sum(2.0 * m * DOMINANT["N"] * DOMINANT["K"] for m in got)
        us = sum(got.values())
        return {"source": "profiles/" + name, "avg_launch_us": {("M=%d" % m): v for m, v in sorted(got.items())},
                "frac": round(fl / (us * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4)}
    return None


def _pmc_traffic():
    """HBM bytes per launch of the same kernel/shape from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 correction, + WRITE_SIZE); PMC counters cannot be sampled from inside this process."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                k = json.load(f)["kernels"][DOMINANT["pmc_key"]]
            hbm, algo = int(k["hbm_bytes_per_launch_corrected"]), int(k["algorithmic_bytes_per_launch"])
            if not 0.5 * algo <= hbm <= 4 * algo:      # a mis-parsed summary (a duration column read as a counter) must not reach the line
                continue
            return hbm, name
        except Exception:
            continue
    return None, None


# Dominant kernel of the step (profiles/r0*_bench_*_kernel_stats.txt): gemm_nt_wide_kernel<affine prologue, linear epilogue + column
# statistics + max-pool partials>, 256x256 tiles (csrc/gemm_wide.hip; round 1 and the first half of round 2 ran it as gemm_nt_kernel<1,0,1,0,1,0>
# with 128x64 tiles), at the Discriminator's 256->1024 layer (Discriminator.py:74-81,104): N = 1024, K = 256, M = B*N points per pass.
# Per step: ONE launch over the rows of three passes (D(real), D(fake), D(interpolate) of the D step, batched with per-pass BatchNorm:
# M = 3*B*N, 103 GF) and ONE launch for D(fake) of the G step (M = B*N, 34.4 GF).  The roofline entry covers both: FLOPs and time
# averaged per launch (achieved = sum FLOPs / sum time), each shape also listed on its own (`per_shape`).
# (The G step's unused D(real) only advances running statistics, TrainStep._seg_g.)
DOMINANT = {"N": 1024, "K": 256, "a_mode": 1,
            "pmc_key": "gemm_nt D.fc2.0 M=65536 N=1024 K=256 (affine prologue + statistics + pooling partials, output not stored)"}


def _cdiv(a, b):
    return (a + b - 1) // b


class MfmaAccounting:
    """HIP events (on the launch stream) around EVERY launch that runs on the matrix cores during a few eagerly issued steps:
    the dominant kernel's launches give `roofline`, all of them give the step's issued MFMA FLOPs and the summed duration of
    the kernels that issue them -> a time-weighted MFMA utilisation.  Which launches use MFMA, and their tile padding, follow the
    dispatch rules of csrc/gemm.hip (launch_nt / launch_tn) and csrc/graph.hip (knn_mfma for feature-space kNN)."""

    def __init__(self, M, peak, f16):
        self.M, self.peak, self.f16 = M, peak, f16
        self.rec = []          # (kind, useful flops, issued flops, e0, e1, dominant?)

    def _classify(self, kind, a):
        if kind == "gemm_nt":
            batch = max(int(a.batch), 1)
            small = a.M <= 64 and a.a_mode != 2 and a.epi_mode != 3 and not a.sp_val and batch == 1 and not a.pool_val
            if small:
                return None
            if self.f16 == "bf16x3" and a.N > 32 and a.mfma_f16:
                bn = 128 if (a.N % 128 == 0 and a.M % 256 == 0 and a.K % 32 == 0) else 64      # 256-row tiles (gemm_wide3.hip) / the 128-row split kernel
            elif self.f16 == "f16" and a.N > 32 and a.mfma_f16:
                bn = 128 if (a.N > 64 and a.K >= 512) else 64
            else:
                bn = 128 if (a.N > 64 and a.K >= 512) else (64 if a.N > 32 else 32)
            useful = 2.0 * a.M * a.N * a.K * batch
            issued = 2.0 * _cdiv(a.M, 128) * 128 * _cdiv(a.N, bn) * bn * _cdiv(a.K, 32) * 32 * batch
            # every launch of the dominant kernel at this layer: one pass (M = B*N, the G step) or three passes in one launch (D step)
            dom = (a.M % self.M == 0 and a.N == DOMINANT["N"] and a.K == DOMINANT["K"] and a.a_mode == DOMINANT["a_mode"] and bool(a.stats))
            return useful, issued, dom
        if kind == "gemm_tn":
            skinny = (a.Na <= 4 or a.Nb <= 4) and a.Na <= 2048 and a.Nb <= 2048
            if skinny and a.b_mode == 0 and not a.a_scale:
                return None
            tb = 128 if a.Nb > 64 else (64 if a.Nb > 32 else 32)
            useful = 2.0 * a.M * a.Na * a.Nb
            return useful, 2.0 * _cdiv(a.M, 32) * 32 * _cdiv(a.Na, 128) * 128 * _cdiv(a.Nb, tb) * tb, False
        if kind == "gemm_dual":                              # weight-gradient + input-gradient product of one layer in one launch
            f = 4.0 * a.M * a.Na * a.Nb
            return f, f, False
        if kind == "knn":
            if a.mode != 0 or a.C < 16:
                return None                                   # coordinate-space kNN: fp64 VALU kernel
            f = 2.0 * a.B * a.N * a.N * a.C
            return f, f, False
        return None

    def __call__(self, kind, a):
        c = self._classify(kind, a)
        if c is None:
            return None
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        self.rec.append((kind, c[0], c[1], e0, e1, c[2], int(getattr(a, "M", 0))))
        return lambda: e1.record(stream)

    def roofline(self, in_graph=None):
        dom = [(r[3], r[4], r[6]) for r in self.rec if r[5]]
        if not dom:
            return None
        r = self._roofline_from_events(dom)
        if in_graph is not None:
            # the figure of record: the launches INSIDE the replayed graph over the timed steps (GraphStamps); the eager-event figure stays
            # beside it -- it times the same kernel at the higher clocks of a GPU that idles between eagerly issued launches
            r["frac_eager_events"] = r["frac"]; r["achieved_eager_events"] = r["achieved"]
            r["per_shape_eager_events"] = r["per_shape"]
            ms, flops = in_graph["avg_launch_ms"], in_graph["flops_per_launch"]
            achieved = flops / (ms * 1e-3) / 1e12
            r.update(achieved=round(achieved, 2), frac=round(achieved / self.peak, 4), flops_per_launch=flops, avg_launch_ms=round(ms, 4),
                     launches_timed=in_graph["launches_timed"], per_shape=in_graph["per_shape"],
                     timing="device wall-clock stamps captured in front of and behind every launch of this kernel INSIDE the replayed hipGraph, "
                            "accumulated over exactly the timed steps (bench.py::GraphStamps); an empty stamp pair in the same graph (%.2f us) is "
                            "subtracted; wall clock %d kHz" % (in_graph["empty_pair_us"], in_graph["wall_clock_khz"]))
        else:
            r["timing"] = "HIP events around eagerly issued launches after the timed region (no replayed graph in this run)"
        ref = _rocprof_reference(self.M) if self.f16 == "f32" else None      # the committed kernel-trace summary is of the fp32-operand run
        if ref is not None:
            r["frac_rocprof_ref"] = ref["frac"]; r["rocprof_ref"] = ref
        return r

    def _roofline_from_events(self, dom):
        # All launches of the dominant kernel at this layer.  SURVEY 8(d): 2*256*1024 FLOPs per point x the points of a launch; the
        # launches differ in size (one pass or three), so FLOPs and time are averaged per launch: achieved = sum FLOPs / sum time.
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in dom) / len(dom)
        flops = sum(2.0 * m * DOMINANT["N"] * DOMINANT["K"] for _, _, m in dom) / len(dom)
        achieved = flops / (ms * 1e-3) / 1e12
        by_m = {}
        for e0, e1, m in dom:
            d = by_m.setdefault(m, [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1)
        per_shape = {("M=%d" % m): {"launches_timed": c, "avg_launch_ms": round(t / c, 4),
                                    "frac": round(2.0 * m * DOMINANT["N"] * DOMINANT["K"] / (t / c * 1e-3) / 1e12 / self.peak, 4)} for m, (c, t) in sorted(by_m.items())}
        rows_avg = sum(m for _, _, m in dom) / len(dom)
        t1, t1_file = _pmc_traffic() if self.f16 == "f32" else (None, None)       # the committed PMC passes measured the fp32-operand kernel
        ksym = {"f32": "gemm_nt_wide_kernel<1,0,0>", "f16": "gemm_nt_wide_kernel<1,0,1> (fp16 operands, v_mfma_f32_32x32x16_f16)",
                "bf16x3": "gemm_nt_wide3_kernel<1,0,2,4,0> (split-bf16 operands, 256 x 256 tiles, v_mfma_f32_32x32x16_bf16 x 6 per k-step)"}.get(self.f16, "gemm_nt")
        return {"bound": "mfma", "kernel": "%s at D.fc2.0 (N=%d K=%d, BN+LeakyReLU prologue, column-statistics + max-pool epilogue, output not stored): "
                                           "per step one launch over the three D-step passes (M=%d) and one for the G step (M=%d)"
                                           % (ksym, DOMINANT["N"], DOMINANT["K"], 3 * self.M, self.M),
                "achieved": round(achieved, 2), "peak": round(self.peak, 1), "unit": "TFLOP/s", "frac": round(achieved / self.peak, 4),
                **({"peak_derivation": SPLIT_BF16_PEAK_NOTE, "achieved_note": "fp32-equivalent FLOPs (2*M*N*K) per second; the bf16 pipe issues 6x as many"} if self.f16 == "bf16x3" else {}),
                "flops_per_launch": flops, "avg_launch_ms": round(ms, 4), "launches_timed": len(dom), "per_shape": per_shape,
                "traffic": None if t1 is None else int(t1 * rows_avg / self.M),
                "traffic_note": ("HBM bytes per launch: measured for the one-pass launch (M=%d) in separate rocprofv3 --pmc passes (profiles/%s: %s B "
                                 "= %.2fx its algorithmic 80.7 MB: A 67.1 MB + W 1.0 MB read once + own statistics/pooling records 12.6 MB written), scaled by the "
                                 "average rows per launch (operand and records grow with the rows)" % (self.M, t1_file, t1, t1 / 80740352.0)) if t1 is not None else
                                "not measured for this operand mode (the committed PMC passes are of the fp32-operand kernel; operands and records are the same bytes)"}

    def summary(self, steps, step_ms):
        if not self.rec:
            return None
        t_ms = sum(r[3].elapsed_time(r[4]) for r in self.rec) / steps
        useful = sum(r[1] for r in self.rec) / steps
        issued = sum(r[2] for r in self.rec) / steps
        by = {}
        for r in self.rec:
            d = by.setdefault(r[0], [0, 0.0, 0.0])
            d[0] += 1; d[1] += r[1]; d[2] += r[3].elapsed_time(r[4])
        return {"mfma_flops_issued_per_step": issued, "mfma_flops_useful_per_step": useful, "mfma_kernel_ms_per_step": round(t_ms, 3),
                "mfma_launches_per_step": len(self.rec) // steps,
                "mfma_util_issued": round(issued / (t_ms * 1e-3) / 1e12 / self.peak, 4),
                "mfma_util_useful": round(useful / (t_ms * 1e-3) / 1e12 / self.peak, 4),
                "mfma_time_share_of_step": round(t_ms / step_ms, 4),
                "by_kind": {k: {"launches_per_step": v[0] // steps, "tflops_useful": round(v[1] / (v[2] * 1e-3) / 1e12, 2), "ms_per_step": round(v[2] / steps, 3)}
                            for k, v in by.items()},
                "note": "HIP events around every matrix-core launch over %d eagerly issued steps (GPU kept busy ahead of the host so the events do not see host "
                        "issue gaps); issued = 2*M*N*K with M, N, K padded to the kernel's tiles; util = FLOPs / (summed duration of these kernels x %.1f TFLOP/s); "
                        "gemm_tn durations include the split-K reduction the same entry point launches when it is not deferred" % (steps, self.peak)}


class GraphStamps:
    """The dominant kernel's duration measured INSIDE the replayed hipGraph, over exactly the timed steps: a pair of one-thread
    device-timestamp launches (spgan_stamp_begin / spgan_stamp_end, csrc/pointwise.hip) is captured directly in front of and behind
    every launch of the dominant kernel; the end launch adds the elapsed wall-clock ticks to a per-shape accumulator that is zeroed
    before the timed region and read after it.  An EMPTY pair captured in the same graph measures what the two launch boundaries add
    (subtracted).  HIP events cannot be queried inside a replayed graph, and events around EAGERLY issued launches (the accounting
    steps below) time the kernel on a GPU whose clocks sit higher than in the continuously busy replayed step: round 4's line read
    0.73 that way where rocprofv3 over the replayed steps read 0.69 (review item 3).  Costs 6 launches of ~2 us per step (0.1 %)."""

    def __init__(self, acct, dev):
        self.acct, self.dev = acct, dev
        self.slots = torch.zeros(8, dtype=torch.int64, device=dev)
        self.acc = {}                 # M -> int64[2] (ticks, launches)
        self.empty = torch.zeros(2, dtype=torch.int64, device=dev)
        self._pool = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(4)]     # allocated outside any capture
        self.seen_in_step = 0
        from spgan import _lib
        self.lib = _lib.load()
        self.khz = int(self.lib.spgan_wall_clock_khz())

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def __call__(self, kind, a):
        c = self.acct._classify(kind, a) if kind == "gemm_nt" else None
        if c is None or not c[2] or self.khz <= 0:
            return None
        m = int(a.M)
        if m not in self.acc:
            if not self._pool:
                return None
            self.acc[m] = self._pool.pop()
        i = list(self.acc.keys()).index(m)
        if i == 0:        # once per step: an empty pair (two launch boundaries, nothing between)
            self.lib.spgan_stamp_begin(self.slots[7:].data_ptr(), self._stream())
            self.lib.spgan_stamp_end(self.slots[7:].data_ptr(), self.empty.data_ptr(), self._stream())
        slot = self.slots[i:].data_ptr()
        acc = self.acc[m].data_ptr()
        self.lib.spgan_stamp_begin(slot, self._stream())
        return lambda: self.lib.spgan_stamp_end(slot, acc, self._stream())

    def reset(self):
        for t in list(self.acc.values()) + [self.empty]:
            t.zero_()

    def result(self, peak):
        """-> dict for the roofline entry, or None when nothing was stamped (eager runs, no wall-clock rate)."""
        if not self.acc or self.khz <= 0:
            return None
        e = self.empty.tolist()
        empty_ms = (e[0] / e[1] / self.khz) if e[1] else 0.0
        per, tot_ms, tot_fl, n = {}, 0.0, 0.0, 0
        for m, t in sorted(self.acc.items()):
            ticks, cnt = t.tolist()
            if not cnt:
                continue
            ms = ticks / cnt / self.khz - empty_ms
            fl = 2.0 * m * DOMINANT["N"] * DOMINANT["K"]
            per["M=%d" % m] = {"launches_timed": cnt, "avg_launch_ms": round(ms, 4), "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4)}
            tot_ms += ms * cnt; tot_fl += fl * cnt; n += cnt
        if not n:
            return None
        return {"avg_launch_ms": tot_ms / n, "flops_per_launch": tot_fl / n, "launches_timed": n, "per_shape": per,
                "empty_pair_us": round(empty_ms * 1e3, 3), "wall_clock_khz": self.khz}


class _KeepBusy:
    """Puts ~ms of heavy work in front of an eagerly issued step, so that the host gets ahead with issuing and the HIP events bracket
    kernels, not issue gaps -- and the chip is at the clocks of a continuously busy GPU (a graph-replayed step), not at the boost
    clocks it reaches between sparse eager launches or the idle clocks a sleeping spin kernel would leave it at (that read 0.38 ms
    for the dominant kernel instead of 0.33).  The filler is the vendor SGEMM (torch.mm) on scratch operands of the dominant shape:
    measurement plumbing, and under rocprofv3 it shows up under its own (Cijk_...) name instead of polluting this library's symbols."""

    def __init__(self, dev, ms=25.0):
        self.A = torch.randn(PER_GPU_BATCH * N_POINTS, DOMINANT["K"], device=dev)
        self.Wt = torch.randn(DOMINANT["K"], DOMINANT["N"], device=dev) * 0.05
        self.out = torch.empty(PER_GPU_BATCH * N_POINTS, DOMINANT["N"], device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self._run(3)
        torch.cuda.synchronize()
        e0.record(); self._run(10); e1.record()
        torch.cuda.synchronize()
        self.n = max(int(ms / (e0.elapsed_time(e1) / 10.0)), 1)

    def _run(self, n):
        for _ in range(n):
            torch.mm(self.A, self.Wt, out=self.out)

    def __call__(self):
        self._run(self.n)


def cpu_baseline(budget_s=45.0):
    """The oracle's train step (same loss composition) on the host cores, bounded sample: the metric's own shape (C2: N=2048,
    batch 32; >= 1 warm-up + up to 3 timed steps inside the budget) and, for continuity with round 1, batch 4."""
    from oracle import spgan_oracle as orc
    from spgan import fixture_rng as fr
    ncpu = os.cpu_count() or 1

    def setup(b, n_pts):
        gp_ = {k: v.requires_grad_(True) for k, v in fr.init_params(orc.generator_shapes(), salt=1, perturb_bn=False).items()}
        dp_ = {k: v.requires_grad_(True) for k, v in fr.init_params(orc.discriminator_shapes(), salt=1, perturb_bn=False).items()}
        st = dict(gp_=gp_, dp_=dp_, gbuf=orc.bn_buffers(orc.generator_shapes()), dbuf=orc.bn_buffers(orc.discriminator_shapes()),
                  optG=orc.AdamState(gp_), optD=orc.AdamState(dp_), x=fr.sphere_template(n_pts)[None].repeat(b, 1, 1),
                  real=fr.synthetic_real(b, n_pts, seed=1234), z1=fr.latent(b, n_pts, NZ, seed=1), z2=fr.latent(b, n_pts, NZ, seed=2),
                  alpha=fr.uniform("bench.alpha.cpu", (b, 1, 1), 0.0, 1.0))
        return st

    def run(st):
        t = time.time()
        orc.train_step(st["gp_"], st["gbuf"], st["dp_"], st["dbuf"], st["optG"], st["optD"], st["x"], st["real"], st["z1"], st["z2"],
                       gan="wgan", use_gp=True, alpha=st["alpha"])
        return time.time() - t

    # PyTorch CPU oversubscribes badly on many-core hosts (256 threads: 143 s/step measured vs 1.5 s/step on 8):
    # pick the fastest thread count on a small probe first, then time the bounded samples with it.
    probe = setup(4, 1024)
    best, best_t = 1, None
    for th in [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        run(probe)
        dt = min(run(probe), run(probe))
        if best_t is None or dt < best_t:
            best, best_t = th, dt
        if dt > 5.0:
            break
    torch.set_num_threads(best)

    def sample(b, budget, max_steps):
        st = setup(b, N_POINTS)
        run(st)                                                          # warm-up
        t0 = time.time(); n = 0
        while True:
            run(st)
            n += 1
            if time.time() - t0 > budget or n >= max_steps:
                break
        return (time.time() - t0) / n, n

    dt4, n4 = sample(4, 6.0, 4)
    dt32, n32 = sample(PER_GPU_BATCH, budget_s - 15.0, 3)
    return {"value": round(PER_GPU_BATCH / dt32, 3), "unit": "shapes/s", "cores": best, "kind": "port", "host_cpus": ncpu,
            "value_batch4": round(4 / dt4, 3),
            "sample": "oracle train steps (WGAN-GP, N=%d; PyTorch CPU fp32, %d threads = fastest of a probe over 8..64 on a %d-CPU host): "
                      "batch %d (the metric's batch): 1 warm-up + %d timed, %.2f s/step; batch 4: 1 warm-up + %d timed, %.2f s/step"
                      % (N_POINTS, best, ncpu, PER_GPU_BATCH, n32, dt32, n4, dt4)}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def self_launch(n):
    """`python bench.py --gpus N` without torchrun: re-execute under torch.distributed.run with one rank per GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL / device-memory sharing across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class _Watchdog:
    """First contact with a multi-GPU node must yield a line, not a hang (round-4 review item 7).  A phase that can block forever (the first
    collective, the first steps of a schedule RCCL has never carried) is armed with a deadline; when it passes, rank 0 prints the line
    `fallback()` returns (the last completed measurement plus a "dp_fallback" key naming the phase) and every rank leaves with os._exit --
    the main thread may be parked inside a HIP / RCCL call that never returns."""

    def __init__(self, rank):
        self.rank, self.timer, self.fallback = rank, None, None

    def arm(self, phase, seconds):
        import threading
        self.disarm()
        self.phase = phase

        def fire():
            line = self.fallback(phase) if self.fallback is not None else None
            if self.rank == 0:
                if line is None:
                    line = {"metric": "G+D train-step shapes/sec", "value": None, "unit": "shapes/s", "higher_is_better": True}
                    line["dp_fallback"] = "no measurement: '%s' did not complete within %.0f s" % (phase, seconds)
                print(json.dumps(line), flush=True)
            os._exit(0 if line is not None and line.get("value") else 3)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def _param_checksum(nets):
    """Order-independent exact checksum of the parameters' BITS (int32 views summed in int64): equal on two ranks / two schedules iff --
    up to a 2^-64 accident -- the parameters are bit-identical."""
    tot = torch.zeros((), dtype=torch.int64, device=next(nets[0].parameters()).device)
    for net in nets:
        for p_ in net.parameters():
            tot = tot + p_.detach().contiguous().view(torch.int32).to(torch.int64).sum()
    return tot


def first_contact(dev, rank, world, variant, wd, seconds):
    """Before anything is timed on N > 1 ranks: (1) one all-reduce of ones under the watchdog -> the ranks RCCL really connects;
    (2) ONE eager step under the strictly sequential schedule and ONE under the overlapped default (the generator's forward of the G step
    issued under D's gradient all-reduce, train.py) from identical state on identical inputs: parameters must be BIT-EQUAL across the two
    schedules and across all ranks (checksums all-gathered).  Returns the report that goes into the JSON line; "overlap_ok" False makes
    main() run the sequential schedule (the SPGAN_DP_OVERLAP=0 behaviour) and say so under "dp_fallback"."""
    import spgan
    rep = {"overlap_ok": True}
    wd.arm("first collective (all-reduce of one float per rank)", seconds)
    ones = torch.ones(1, device=dev)
    torch.distributed.all_reduce(ones)
    rep["rccl_ranks_seen"] = int(round(ones.item()))
    wd.arm("equivalence probe: one sequential and one overlapped eager step", seconds)
    sums = []
    try:
        for overlap in (False, True):
            G, D = build_models(dev, variant)
            tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, distributed=True, graph=False)
            tr.overlap_g_forward = overlap
            x, real, zs, alpha = make_inputs(dev, rank, PER_GPU_BATCH)
            tr.step(x, real, zs[0], zs[1], alpha=alpha)
            if overlap and os.environ.get("SPGAN_BENCH_TEST_BREAK", "") == "overlap":      # test hook: a schedule that computes something else
                with torch.no_grad():
                    next(G.parameters()).add_(1e-3)
            sums.append(_param_checksum((G, D)))
            if rep.get("native_comm_world") is None and getattr(tr.dpD, "_comm", None) is not None:
                from spgan import _lib
                rep["native_comm_world"] = int(_lib.load().spgan_comm_world(tr.dpD._comm))
            del tr, G, D
        both = torch.stack(sums)
        gathered = [torch.zeros_like(both) for _ in range(world)]
        torch.distributed.all_gather(gathered, both)
        allsums = torch.stack(gathered).cpu()                     # [world, 2]
        rep["schedules_bit_equal"] = bool((allsums[:, 0] == allsums[:, 1]).all())
        rep["ranks_bit_equal"] = bool((allsums == allsums[0:1]).all())
        if not (rep["schedules_bit_equal"] and rep["ranks_bit_equal"]):
            rep["overlap_ok"] = False
            rep["dp_fallback"] = "sequential schedule: the overlapped step's parameters are not bit-equal to the sequential step's%s" % (
                "" if rep["ranks_bit_equal"] else " / not equal on all ranks")
    except Exception as e:                                   # noqa: BLE001
        rep["overlap_ok"] = False
        rep["dp_fallback"] = "sequential schedule: the equivalence probe raised %s: %s" % (type(e).__name__, str(e)[:200])
    wd.disarm()
    return rep


def time_steps(tr, step_fn, steps, dist_on, dev):
    if dist_on:
        torch.distributed.barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    t_issue = time.perf_counter() - t0        # host time to enqueue the K steps (the GPU may still be running)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    return dt, t_issue


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # >= 50 timed steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=("c2", "c4", "c5"), default="c2",
                    help="BASELINE config whose per-GPU shape is timed: c2 = configs[1] (N=2048, batch 32, fp32; default, the headline), "
                         "c4 = configs[3] per GPU (N=4096, batch 16), c5 = configs[4] per GPU (N=2048, batch 32, fp16 MFMA operands)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--experiment-batch", type=int, default=None, help="per-GPU batch other than BASELINE's 32: an experiment, marked as such in the line")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the drop-in-caller and MFMA-accounting legs (they run after the timed region)")
    ap.add_argument("--no-graph", action="store_true", help="issue every step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--mfma", choices=("f32", "f16", "bf16x3"), default="f32",
                    help="operand precision of the MFMA contractions: f32 (BASELINE configs[1], the headline) or f16 operands with fp32 "
                         "accumulation (the per-GPU shape of BASELINE configs[4], 'fp16 MFMA MLPs')")
    ap.add_argument("--reference-schedule", action="store_true",
                    help="also evaluate the two provably redundant pieces of the reference loop (EdgeConv1 on every copy of the tiled "
                         "sphere, the G step's unused D(real) forward) in the HEADLINE timing -- for comparison; see DESIGN.md")
    ap.add_argument("--variant", default="", help="comma-separated non-default generator flags (attn, eql, use_head, off, z_norm) or "
                    "small_d: times that variant instead of the headline configuration (SURVEY 8(f) N4); the JSON line says so")
    args = ap.parse_args()
    if args.config == "c5":
        if args.mfma not in ("f32", "f16"):
            raise SystemExit("--config c5 is the fp16-operand mode (--mfma f16)")
        args.mfma = "f16"
    variant = tuple(v for v in args.variant.split(",") if v)

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))                     # the ranks print the JSON line; this process only waits for them
    if args.gpus != world_env:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start `python bench.py --gpus N` directly or under torch.distributed.run with "
                         "--nproc-per-node N" % (args.gpus, world_env))

    import spgan
    if SELFTEST:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import install_kernel_models
        install_kernel_models()
        torch.set_num_threads(2)
    world = world_env
    # SPGAN_DIST_BACKEND=gloo: rehearse the N > 1 schedule (graphs, overlapped all-reduce, rank-synchronous Adam) on a box with ONE GPU -- all
    # ranks share device 0 and the all-reduce goes through gloo's CUDA path.  A functional check only: such a line carries "rehearsal".
    backend = "gloo" if SELFTEST else os.environ.get("SPGAN_DIST_BACKEND", "nccl")
    rank = spgan.init_process_group_from_env(backend)        # no-op for a plain single-process launch
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if SELFTEST:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
        if backend != "nccl":
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    world_seen = torch.distributed.get_world_size() if dist_on else 1
    if world_seen != args.gpus:
        raise SystemExit("process group has %d ranks, --gpus asked for %d" % (world_seen, args.gpus))

    spgan.ops.set_mfma_operands(args.mfma)
    # The step is captured once into a hipGraph and replayed (TrainStep(graph=True)): issuing its launches from Python takes
    # as long as the GPU needs to run them.  Data-parallel runs capture three graphs (D step | Adam(D) + G step | Adam(G)) and issue
    # the two RCCL all-reduces eagerly between them; SPGAN_GRAPH=0 / --no-graph fall back to eager issue.
    use_graph = not args.no_graph and os.environ.get("SPGAN_GRAPH", "1") != "0" and not SELFTEST
    graph_warmup = 3
    x, real, zs, alpha = make_inputs(dev, rank, PER_GPU_BATCH)

    def measure(overlap=None):
        """Fresh models (same initialisation), prime + capture, W warm-up steps, K timed steps.  overlap: None = TrainStep's default."""
        G_, D_ = build_models(dev, variant)
        tr_ = spgan.TrainStep(G_, D_, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, distributed=dist_on, graph=use_graph,
                              graph_warmup=graph_warmup, reference_schedule=args.reference_schedule)
        if overlap is not None:
            tr_.overlap_g_forward = bool(overlap)
        step_ = lambda i: tr_.step(x, real, zs[(2 * i) % 4], zs[(2 * i + 1) % 4], alpha=alpha)
        if use_graph:
            if stamps is not None:
                spgan.ops.launch_timer = stamps   # the dominant kernel's launches are captured between device-timestamp launches (GraphStamps)
            for i in range(graph_warmup + 1):     # eager priming steps + the capture, before the W warm-up steps
                step_(i)
            spgan.ops.launch_timer = None
        for i in range(args.warmup):
            step_(i)
        if stamps is not None:
            stamps.reset()                        # accumulate over exactly the K timed steps
        if overlap and os.environ.get("SPGAN_BENCH_TEST_HANG", "") == "overlap":      # test hook: the overlapped schedule never returns
            time.sleep(1e6)
        dt_, t_issue_ = time_steps(tr_, step_, args.steps, dist_on, dev)
        return G_, D_, tr_, dt_, t_issue_

    peak = PEAK_OF_MODE[args.mfma]
    acct = MfmaAccounting(PER_GPU_BATCH * N_POINTS, peak, args.mfma) if (rank == 0 and not SELFTEST) else None
    stamps = GraphStamps(acct, dev) if (acct is not None and use_graph and os.environ.get("SPGAN_BENCH_STAMPS", "1") != "0") else None
    contact, dp_seq, wd = None, None, None

    def basic_line(m, schedule, why):
        """The measured line a watchdog prints when a LATER phase hangs on a multi-GPU node (the complete line is assembled at the end)."""
        return {"metric": "G+D train-step shapes/sec @%d pts, bs=%d per GPU (WGAN-GP)" % (N_POINTS, PER_GPU_BATCH), "value": m["shapes_per_s"],
                "unit": "shapes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
                "config": {"workload": "BASELINE %s per-GPU shape, data parallel" % args.config, "global_batch": PER_GPU_BATCH * world,
                           "n_points": N_POINTS, "parallelism": "dp%d" % world},
                "world_size_observed": world_seen, "hipgraph_replay": bool(use_graph),
                "first_contact": {k: v for k, v in (contact or {}).items() if k != "overlap_ok"}, "dp_schedule": schedule, "dp_fallback": why}

    if dist_on and world > 1:
        # First contact with a multi-GPU node (no such node exists in this build's loop): never hang, always leave a line.  Order = risk:
        # the first collective, the strictly SEQUENTIAL schedule measured in full, the equivalence probe, and only then the OVERLAPPED
        # default -- each under a deadline after which rank 0 prints what has been measured so far with a "dp_fallback" key.
        seconds = float(os.environ.get("SPGAN_BENCH_WATCHDOG_S", "300"))
        wd = _Watchdog(rank)
        contact = first_contact(dev, rank, world, variant, wd, seconds)
        wd.arm("sequential schedule: capture, warm-up and the timed steps", seconds)
        G, D, tr, dt, t_issue = measure(overlap=False)
        wd.disarm()
        dp_seq = {"ms_per_step": round(dt / args.steps * 1e3, 3), "shapes_per_s": round(PER_GPU_BATCH * world * args.steps / dt, 2)}
        if contact["overlap_ok"] and os.environ.get("SPGAN_DP_OVERLAP", "1") != "0":
            wd.fallback = lambda phase: basic_line(dp_seq, "sequential", "sequential schedule measured in full; '%s' did not complete within %.0f s" % (phase, seconds))
            wd.arm("overlapped schedule (generator forward under D's all-reduce): capture, warm-up and the timed steps", seconds)
            G2_, D2_, tr2_, dt2_, ti2_ = measure(overlap=True)
            wd.disarm()
            wd.fallback = None
            del G, D, tr
            G, D, tr, dt, t_issue = G2_, D2_, tr2_, dt2_, ti2_
            dp_schedule = "overlapped"
        else:
            dp_schedule = "sequential"
            if "dp_fallback" not in contact and os.environ.get("SPGAN_DP_OVERLAP", "1") == "0":
                contact["dp_fallback"] = "sequential schedule: SPGAN_DP_OVERLAP=0"
    else:
        G, D, tr, dt, t_issue = measure()

    in_graph = stamps.result(peak) if stamps is not None else None
    if wd is not None:
        # the phases after the timed region (eager accounting steps with their own collectives) run under the deadline too: what is measured is kept
        final_m = {"ms_per_step": round(dt / args.steps * 1e3, 3), "shapes_per_s": round(PER_GPU_BATCH * world * args.steps / dt, 2)}
        wd.fallback = lambda phase: basic_line(final_m, dp_schedule, "timed region complete; '%s' did not complete within %.0f s" % (phase, seconds))
        wd.arm("accounting steps after the timed region", seconds)
    ACCT_STEPS = 4
    if not SELFTEST:
        # a replayed graph offers no per-launch hook: the matrix-core launches are bracketed with HIP events over a few eager steps
        # of the same TrainStep right after the timed region (same process, same tensors; not part of `value`).  Data-parallel:
        # every rank runs them (the all-reduces inside must match up); only rank 0 records.
        busy = _KeepBusy(dev)
        spgan.ops.launch_timer = acct
        for i in range(ACCT_STEPS):
            busy()
            tr._eager_step(x, real, zs[(2 * i) % 4], zs[(2 * i + 1) % 4], alpha=alpha)
        torch.cuda.synchronize()
        spgan.ops.launch_timer = None

    drop_in = None
    if wd is not None:
        wd.disarm()
    if not SELFTEST and not args.no_extra_legs and not variant and world == 1:      # comparison legs: single-GPU runs only (nothing after the timed region may cost a multi-GPU line)
        # The same step as an unmodified reference loop would drive it (model.py:246-248,272-273): z tiled to [B,N,nz], the G step's
        # unused D(real) forward evaluated, EdgeConv1 on every copy of the tiled sphere.  Fresh models (same initialisation), its own
        # captured graph; every rank takes part.
        G2, D2 = build_models(dev, variant)
        tr2 = spgan.TrainStep(G2, D2, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, distributed=dist_on, graph=use_graph,
                              graph_warmup=graph_warmup, reference_schedule=True)
        x2, real2, zt, alpha2 = make_inputs(dev, rank, PER_GPU_BATCH, tiled_z=True)
        step2 = lambda i: tr2.step(x2, real2, zt[0], zt[1], alpha=alpha2)
        for i in range((graph_warmup + 1 if use_graph else 0) + 2):
            step2(i)
        dt2, _ = time_steps(tr2, step2, 10, dist_on, dev)
        drop_in = {"ms_per_step_reference_schedule_tiled_z": round(dt2 / 10 * 1e3, 3),
                   "shapes_per_s": round(PER_GPU_BATCH * world * 10 / dt2, 2),
                   "note": "z handed over tiled [B,N,128] (33.5 MB), reference_schedule=True: EdgeConv1 on all B copies of the sphere, full D(real) "
                           "forward in the G step; 10 steps after priming"}
        del tr2, G2, D2, zt

    literal = None
    if not SELFTEST and not args.no_extra_legs and not variant and world == 1:
        # The reference's loop body EXECUTED LITERALLY (examples/reference_loop.py: model.py:239-279 statement for statement -- separate G() /
        # D() calls, requires_grad toggles, dis_loss / gen_loss, .backward(), torch.optim.Adam) on the HIP modules, latent tiled
        # [B,N,128]; no TrainStep.  (a) issued eagerly from Python, (b) the same function under spgan.CapturedBody (the caller's own
        # loop body replayed as a hipGraph).
        from reference_loop import LoopState, reference_loop_body
        G3, D3 = build_models(dev, variant)
        G3.train(); D3.train()
        oG = torch.optim.Adam(filter(lambda p: p.requires_grad, G3.parameters()), lr=1e-4, betas=(0.5, 0.99), capturable=True)
        oD = torch.optim.Adam(filter(lambda p: p.requires_grad, D3.parameters()), lr=1e-4, betas=(0.5, 0.99), capturable=True)
        gp3 = spgan.GradientPenalty(10.0, gamma=1)
        x3, real3, zt3, alpha3 = make_inputs(dev, rank, PER_GPU_BATCH, tiled_z=True)
        st3 = LoopState(G3, D3, oG, oD, gan="wgan", gp=lambda netD, r, f: gp3(netD, r, f, alpha=alpha3))
        fn3 = lambda x_, d_, a_, b_: reference_loop_body(st3, x_, d_, a_, b_)[:2]
        body = spgan.CapturedBody(fn3, modules=(G3, D3), warmup=2)             # captured leg first: the body is wrapped from its first call
        for i in range(4):
            body(x3, real3, zt3[i % 2], zt3[(i + 1) % 2])
        dt4, iss4 = time_steps(None, lambda i: body(x3, real3, zt3[i % 2], zt3[(i + 1) % 2]), 10, False, dev)
        for _ in range(3):
            fn3(x3, real3, zt3[0], zt3[1])
        dt3, iss3 = time_steps(None, lambda i: fn3(x3, real3, zt3[i % 2], zt3[(i + 1) % 2]), 10, False, dev)
        literal = {"eager_ms_per_step": round(dt3 / 10 * 1e3, 3), "eager_host_issue_ms_per_step": round(iss3 / 10 * 1e3, 3),
                   "eager_shapes_per_s": round(PER_GPU_BATCH * 10 / dt3, 2),
                   "captured_ms_per_step": round(dt4 / 10 * 1e3, 3), "captured_shapes_per_s": round(PER_GPU_BATCH * 10 / dt4, 2),
                   "captured": bool(body._graph is not None and not body.eager),
                   "note": "model.py:239-279 statement for statement (examples/reference_loop.py::reference_loop_body) with torch.optim.Adam(capturable=True), "
                           "tiled latent [B,N,128], WGAN-GP composition; eager = issued from Python, captured = the same function under "
                           "spgan.CapturedBody (hipGraph replay of the caller's own loop body); 10 steps each after warm-up"}
        del G3, D3, body, st3

    split = None
    if not SELFTEST and not args.no_extra_legs and not variant and world == 1 and args.mfma == "f32" and rank == 0:
        # The SAME step with the split-bf16 products (`--mfma bf16x3`): fresh models (same initialisation), its own captured graph, 20 timed
        # steps after priming, then the matrix-core accounting of that mode over 4 eager steps.  Reported beside the fp32 headline, not as it.
        spgan.ops.set_mfma_operands("bf16x3")
        try:
            G4, D4 = build_models(dev, variant)
            tr4 = spgan.TrainStep(G4, D4, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, distributed=False, graph=use_graph,
                                  graph_warmup=graph_warmup, reference_schedule=args.reference_schedule)
            step4 = lambda i: tr4.step(x, real, zs[(2 * i) % 4], zs[(2 * i + 1) % 4], alpha=alpha)
            for i in range((graph_warmup + 1 if use_graph else 0) + 3):
                step4(i)
            n4 = 20
            dt4s, _ = time_steps(tr4, step4, n4, False, dev)
            acct4 = MfmaAccounting(PER_GPU_BATCH * N_POINTS, SPLIT_BF16_PEAK_TFLOPS, "bf16x3")
            busy4 = _KeepBusy(dev)
            spgan.ops.launch_timer = acct4
            for i in range(ACCT_STEPS):
                busy4()
                tr4._eager_step(x, real, zs[(2 * i) % 4], zs[(2 * i + 1) % 4], alpha=alpha)
            torch.cuda.synchronize()
            spgan.ops.launch_timer = None
            ms4 = dt4s / n4 * 1e3
            split = {"mfma": "bf16x3", "dtype": DTYPE_OF_MODE["bf16x3"], "ms_per_step": round(ms4, 3), "shapes_per_s": round(PER_GPU_BATCH * n4 / dt4s, 2),
                     "steps": n4, "hipgraph_replay": bool(use_graph), "roofline": acct4.roofline(None), "mfma_accounting": acct4.summary(ACCT_STEPS, ms4),
                     "note": "`python bench.py --mfma bf16x3` in this process: same models, inputs and schedule; every large gemm_nt product on the bf16 "
                             "matrix pipe with exactly split operands (csrc/gemm_wide3.hip); not the headline (`value` above is the fp32-operand step)"}
            del tr4, G4, D4
        finally:
            spgan.ops.launch_timer = None
            spgan.ops.set_mfma_operands(args.mfma)

    if rank == 0:
        ms = dt / args.steps * 1e3
        shapes_s = PER_GPU_BATCH * world * args.steps / dt
        line = {
            "metric": "G+D train-step shapes/sec @%d pts, bs=%d per GPU (WGAN-GP)" % (N_POINTS, PER_GPU_BATCH), "value": round(shapes_s, 2), "unit": "shapes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_OF_MODE[args.mfma],
            "data": "synthetic",
            "config": {"workload": "%s: Chair-shaped synthetic clouds, %d pts, per-GPU batch %d, WGAN + gradient penalty (lambda 10), "
                                   "1 D-step + 1 G-step, Adam(1e-4, (0.5,0.99)), k=10; one latent per shape (default noise_generator) handed over un-tiled [b,1,128]"
                                   % ("EXPERIMENT (not a BASELINE config)" if EXPERIMENT_BATCH is not None else
                                      {"c2": "BASELINE configs[1]", "c4": "BASELINE configs[3] per-GPU shape (4096 pts, 128 shapes over 8 GPUs = 16 per GPU)",
                                       "c5": "BASELINE configs[4] per-GPU shape (fp16 MFMA operands)"}[args.config], N_POINTS, PER_GPU_BATCH),
                       "global_batch": PER_GPU_BATCH * world, "n_points": N_POINTS,
                       "parallelism": "dp%d" % world},
            "world_size_observed": world_seen, "collective_backend": (torch.distributed.get_backend() if dist_on else None),
            "bench_config": args.config, "gf_per_shape_step_reference": GF_PER_SHAPE_STEP,
            "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 3), "hipgraph_replay": bool(use_graph), "reference_schedule": bool(args.reference_schedule),
            "step_tflops_algorithmic": round(shapes_s * GF_PER_SHAPE_STEP / 1e3, 2),
            "reference_formulation_tflops_over_fp32_matrix_peak_NOT_a_utilisation": round(shapes_s * GF_PER_SHAPE_STEP / 1e3 / (FP32_MATRIX_PEAK_TFLOPS * world), 4),
        }
        if contact is not None:
            line["first_contact"] = {k: v for k, v in contact.items() if k not in ("overlap_ok", "dp_fallback")}
            line["rccl_ranks_seen"] = contact.get("rccl_ranks_seen")
            line["dp_schedule"] = dp_schedule
            line["dp_sequential_schedule"] = dp_seq
            if "dp_fallback" in contact:
                line["dp_fallback"] = contact["dp_fallback"]
        if EXPERIMENT_BATCH is not None and not SELFTEST:
            line["experiment"] = "per-GPU batch %d instead of BASELINE's 32" % PER_GPU_BATCH
        if dist_on and not SELFTEST and backend != "nccl":
            line["rehearsal"] = "all %d ranks on one GPU, all-reduce through %s: a functional run of the N > 1 schedule, not a measurement" % (world, backend)
        if SELFTEST:
            line["selftest"] = True
            line["data"] = "selftest: CPU test doubles, tiny shapes -- NOT a measurement"
            line["config"]["workload"] = "SELFTEST (N=%d, per-rank batch %d, gloo, kernel-model doubles): plumbing only" % (N_POINTS, PER_GPU_BATCH)
        if variant:
            line["variant"] = list(variant)
            line["config"]["workload"] += "; NON-HEADLINE variant flags: " + ",".join(variant)
        if acct is not None:
            line["roofline"] = acct.roofline(in_graph)
            line["mfma"] = acct.summary(ACCT_STEPS, ms)
            if line["mfma"] is not None:
                # FLOPs the build really issues on the matrix cores / whole step time / peak (the reference-formulation fraction above
                # divides FLOPs the build does not execute)
                line["step_mfma_frac_issued"] = round(line["mfma"]["mfma_flops_issued_per_step"] / (ms * 1e-3) / 1e12 / (peak * 1.0), 4)
            if EXPERIMENT_BATCH is None and not variant:
                line["hbm_traffic"] = _pmc_step_traffic("c4" if args.config == "c4" else "c2", args.mfma)
        if drop_in is not None:
            line["drop_in_caller"] = drop_in
        if literal is not None:
            line["literal_loop"] = literal
        if split is not None:
            line["split_bf16"] = split
        if world == 1 and not args.no_cpu_baseline and not SELFTEST:
            line["cpu_baseline"] = cpu_baseline()
            line["speedup_vs_cpu_baseline"] = round(shapes_s / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
