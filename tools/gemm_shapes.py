#!/usr/bin/env python3
"""Per-shape timing of every GEMM launch of one train step (development aid).
Wraps the two C-ABI GEMM entry points with HIP events, runs a few bench steps and prints, per
(entry, M, N, K, prologue, epilogue) signature: calls/step, average us, TFLOP/s, ms/step."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
import bench
import spgan
from spgan import _lib


class Proxy:
    def __init__(self, lib):
        self._lib = lib
        self.rec = []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in ("spgan_gemm_nt", "spgan_gemm_tn"):
            return fn

        def wrapped(ref, stream):
            a = ref._obj
            if name == "spgan_gemm_nt":
                sig = ("nt", a.M, a.N, a.K, a.a_mode, a.epi_mode, int(bool(a.stats)), int(bool(a.sp_val)))
            else:
                sig = ("tn", a.M, a.Na, a.Nb, a.b_mode if hasattr(a, "b_mode") else -1, int(bool(a.a_scale)), 0, int(bool(a.a_sp_val)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(ref, stream)
            e1.record()
            self.rec.append((sig, e0, e1))
            return r
        return wrapped


def main():
    steps = 3
    dev = torch.device("cuda", 0)
    G, D = bench.build_models(dev)
    tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, distributed=False)
    x, real, zs, alpha = bench.make_inputs(dev, 0, bench.PER_GPU_BATCH)
    for i in range(3):
        tr.step(x, real, zs[0], zs[1], alpha=alpha)
    torch.cuda.synchronize()
    px = Proxy(_lib.load())
    _lib._lib = px
    for i in range(steps):
        tr.step(x, real, zs[0], zs[1], alpha=alpha)
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for sig, e0, e1 in px.rec:
        t = e0.elapsed_time(e1) * 1e3
        c = agg.setdefault(sig, [0, 0.0])
        c[0] += 1; c[1] += t
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values()) / steps / 1e3
    print("# GEMM launches by shape; %d steps; total %.3f ms/step (event-timed, includes launch gaps)" % (steps, tot))
    print("%-4s %8s %6s %6s %4s %4s %3s %3s %7s %9s %8s %8s" % ("kind", "M", "N/Na", "K/Nb", "pro", "epi", "st", "sp", "n/step", "avg_us", "TF", "ms/step"))
    for sig, (n, t) in rows:
        kind, M, N, K = sig[:4]
        avg = t / n
        print("%-4s %8d %6d %6d %4d %4d %3d %3d %7.1f %9.1f %8.1f %8.3f" % (kind, M, N, K, sig[4], sig[5], sig[6], sig[7], n / steps, avg,
                                                                   2.0 * M * N * K / avg / 1e6, t / steps / 1e3))


if __name__ == "__main__":
    main()
