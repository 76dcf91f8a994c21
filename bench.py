#!/usr/bin/env python3
"""G+D train-step throughput of the MI355X-native SP-GAN hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one iteration of the reference loop body (Generation/model.py:239-279): D-step + G-step
with both Adam updates (and, for N>1, both flat gradient all-reduces), on BASELINE config 2:
Chair-shaped synthetic clouds, 2048 points, per-GPU batch 32, WGAN loss + gradient penalty (lambda 10),
fp32.  Weak scaling: the per-GPU batch is fixed, global batch = 32*N.  Inputs are resident in HBM.

The JSON line also carries
  roofline     : the dominant kernel (the fp32-MFMA gemm_nt), its algorithmic FLOPs per launch / average
                 launch time measured with HIP events on the launch stream, against the 157.3 TFLOP/s fp32 matrix peak;
  cpu_baseline : the CPU oracle (a PyTorch restatement of the reference, kind "port") timed on this box's host
                 cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sp-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch   # noqa: E402

N_POINTS = 2048
PER_GPU_BATCH = int(os.environ.get("SPGAN_BENCH_BATCH", "32"))   # 32 = BASELINE configs[1]; the override is for experiments only
NZ = 128
K_NN = 10
FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
FP16_MATRIX_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (never the 2:1-sparsity figure)
# algorithmic FLOPs per shape per step, reference formulation: WGAN-GP at N=2048
GF_PER_SHAPE_STEP = 32.6


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


def make_inputs(dev, rank, b):
    from spgan import fixture_rng as fr
    x = fr.sphere_template(N_POINTS)[None].repeat(b, 1, 1).to(dev)
    real = fr.synthetic_real(b, N_POINTS, seed=1234 + rank).to(dev)
    # one latent per shape (the reference's default noise_generator, model.py:128-131); passed un-tiled [b,1,nz] -- spgan.Generator
    # evaluates the latent half of head.0 per shape instead of tiling it over the 2048 points first
    zs = [fr.latent(b, N_POINTS, NZ, seed=4321 + rank + i)[:, :1, :].contiguous().to(dev) for i in range(4)]
    alpha = fr.uniform("bench.alpha.%d" % rank, (b, 1, 1), 0.0, 1.0).to(dev)
    return x, real, zs, alpha


def _pmc_traffic():
    """HBM bytes per launch of the same kernel/shape from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 correction, + WRITE_SIZE); PMC counters cannot be sampled from inside this process."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_gemm_nt.json")) as f:
            return json.load(f)["kernels"][DOMINANT["pmc_key"]]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


# Dominant kernel of the step (profiles/r01_bench_*_kernel_stats.txt): gemm_nt_kernel<affine prologue, linear epilogue + column
# statistics + max-pool partials, 128x64 tile> at the Discriminator's 256->1024 layer (Discriminator.py:74-81,104): M = B*N points,
# N = 1024, K = 256; 4 launches per step (D(real), D(fake), D(interpolate) of the D step and D(fake) of the G step; the G step's
# unused D(real) only advances running statistics, TrainStep._seg_g).
DOMINANT = {"N": 1024, "K": 256, "a_mode": 1,
            "pmc_key": "gemm_nt D.fc2.0 M=65536 N=1024 K=256 (affine prologue + statistics + pooling partials, output not stored)"}


class DominantKernelTimer:
    """Brackets every launch of the dominant kernel inside the timed region with HIP events recorded on the launch stream
    (spgan.ops.launch_timer hook) -- the live per-launch duration behind `roofline.achieved`."""

    def __init__(self, M, peak=FP32_MATRIX_PEAK_TFLOPS):
        self.M, self.events, self.peak = M, [], peak

    def __call__(self, kind, a):
        if kind != "gemm_nt" or a.M != self.M or a.N != DOMINANT["N"] or a.K != DOMINANT["K"] or a.a_mode != DOMINANT["a_mode"] or not a.stats:
            return None
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        self.events.append((e0, e1))
        return lambda: e1.record(stream)

    def roofline(self):
        if not self.events:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1 in self.events) / len(self.events)
        flops = 2.0 * self.M * DOMINANT["N"] * DOMINANT["K"]          # 2*N*256*1024 per shape x the shapes of one launch
        achieved = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "gemm_nt_kernel<1,0,1,0,1> at D.fc2.0 (M=%d N=%d K=%d, BN+LeakyReLU prologue, column-statistics + max-pool epilogue, output not stored)"
                                           % (self.M, DOMINANT["N"], DOMINANT["K"]),
                "achieved": round(achieved, 2), "peak": self.peak, "unit": "TFLOP/s", "frac": round(achieved / self.peak, 4),
                "flops_per_launch": flops, "avg_launch_ms": round(ms, 4), "launches_timed": len(self.events), "traffic": _pmc_traffic()}


def cpu_baseline(budget_s=25.0):
    """The oracle's train step (same loss composition) on the host cores, bounded sample: C2 shape at a reduced
    batch (the [B,N,N] sort of the reference formulation needs ~64 MB per shape per EdgeConv)."""
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
    # pick the fastest thread count on a tiny probe first, then time the bounded sample with it.
    probe = setup(2, 512)
    best, best_t = 1, None
    for th in [t for t in (4, 8, 16, 32, 64) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        run(probe)
        dt = min(run(probe), run(probe))
        if best_t is None or dt < best_t:
            best, best_t = th, dt
        if dt > 5.0:
            break
    torch.set_num_threads(best)
    b = 4
    st = setup(b, N_POINTS)
    run(st)                                                          # warm-up
    t0 = time.time(); n = 0
    while True:
        run(st)
        n += 1
        if time.time() - t0 > budget_s or n >= 8:
            break
    dt = (time.time() - t0) / n
    return {"value": round(b / dt, 3), "unit": "shapes/s", "cores": best, "kind": "port", "host_cpus": ncpu,
            "sample": "%d oracle train steps (WGAN-GP, N=%d, batch %d; PyTorch CPU fp32, %d threads = fastest of a probe over 4..64 "
                      "on a %d-CPU host), %.2f s/step" % (n, N_POINTS, b, best, ncpu, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue every step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--mfma", choices=("f32", "f16"), default="f32",
                    help="operand precision of the MFMA contractions: f32 (BASELINE configs[1], the headline) or f16 operands with fp32 "
                         "accumulation (the per-GPU shape of BASELINE configs[4], 'fp16 MFMA MLPs')")
    ap.add_argument("--reference-schedule", action="store_true",
                    help="also evaluate the two provably redundant pieces of the reference loop (EdgeConv1 on every copy of the tiled "
                         "sphere, the G step's unused D(real) forward) -- for comparison; see DESIGN.md")
    ap.add_argument("--variant", default="", help="comma-separated non-default generator flags (attn, eql, use_head, off, z_norm) or "
                    "small_d: times that variant instead of the headline configuration (SURVEY 8(f) N4); the JSON line says so")
    args = ap.parse_args()
    variant = tuple(v for v in args.variant.split(",") if v)

    import spgan
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = spgan.init_process_group_from_env("nccl")        # no-op for a plain single-process launch
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))

    spgan.ops.set_mfma_operands(args.mfma)
    G, D = build_models(dev, variant)
    # The step is captured once into a hipGraph and replayed (TrainStep(graph=True)): issuing its ~570 launches from Python takes
    # as long as the GPU needs to run them.  Data-parallel runs capture three graphs (D step | Adam(D) + G step | Adam(G)) and issue
    # the two RCCL all-reduces eagerly between them; SPGAN_GRAPH=0 / --no-graph fall back to eager issue.
    use_graph = not args.no_graph and os.environ.get("SPGAN_GRAPH", "1") != "0"
    graph_warmup = 3
    tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, distributed=dist_on, graph=use_graph,
                         graph_warmup=graph_warmup, reference_schedule=args.reference_schedule)
    x, real, zs, alpha = make_inputs(dev, rank, PER_GPU_BATCH)

    def one_step(i):
        tr.step(x, real, zs[(2 * i) % 4], zs[(2 * i + 1) % 4], alpha=alpha)

    if use_graph:
        for i in range(graph_warmup + 1):     # eager priming steps + the capture, before the W warm-up steps
            one_step(i)
    for i in range(args.warmup):
        one_step(i)
    timer = DominantKernelTimer(PER_GPU_BATCH * N_POINTS, FP32_MATRIX_PEAK_TFLOPS if args.mfma == "f32" else FP16_MATRIX_PEAK_TFLOPS) if rank == 0 else None
    if not use_graph:
        spgan.ops.launch_timer = timer
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    t_issue = time.perf_counter() - t0        # host time to enqueue the K steps (the GPU may still be running)
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    spgan.ops.launch_timer = None
    if use_graph and rank == 0:
        # a replayed graph offers no per-launch hook: the dominant kernel's launches are bracketed with HIP events over a few
        # eager steps of the same TrainStep right after the timed region (same process, same tensors; not part of `value`)
        spgan.ops.launch_timer = timer
        for i in range(4):
            tr._eager_step(x, real, zs[(2 * i) % 4], zs[(2 * i + 1) % 4], alpha=alpha)
        torch.cuda.synchronize()
        spgan.ops.launch_timer = None

    if rank == 0:
        ms = dt / args.steps * 1e3
        shapes_s = PER_GPU_BATCH * world * args.steps / dt
        line = {
            "metric": "G+D train-step shapes/sec @2048 pts, bs=32 per GPU (WGAN-GP)", "value": round(shapes_s, 2), "unit": "shapes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.mfma == "f32" else "f16 MFMA operands, f32 accumulate/epilogues/weight-gradients",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Chair-shaped synthetic clouds, 2048 pts, per-GPU batch 32, WGAN + gradient penalty (lambda 10), "
                                   "1 D-step + 1 G-step, Adam(1e-4, (0.5,0.99)), k=10; one latent per shape (default noise_generator) handed over un-tiled [b,1,128]", "global_batch": PER_GPU_BATCH * world, "n_points": N_POINTS,
                       "parallelism": "dp%d" % world},
            "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 3), "hipgraph_replay": bool(use_graph), "reference_schedule": bool(args.reference_schedule),
            "step_tflops_algorithmic": round(shapes_s * GF_PER_SHAPE_STEP / 1e3, 2),
            "step_frac_of_fp32_matrix_peak": round(shapes_s * GF_PER_SHAPE_STEP / 1e3 / (FP32_MATRIX_PEAK_TFLOPS * world), 4),
        }
        if variant:
            line["variant"] = list(variant)
            line["config"]["workload"] += "; NON-HEADLINE variant flags: " + ",".join(variant)
        line["roofline"] = timer.roofline()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["speedup_vs_cpu_baseline"] = round(shapes_s / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
