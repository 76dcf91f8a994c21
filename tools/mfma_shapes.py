#!/usr/bin/env python3
"""Per-shape table of the matrix-core launches of one train step (bench.py's MfmaAccounting with the shape kept): which GEMM
shapes the step spends its MFMA time on and at what rate.  `python tools/mfma_shapes.py [--mfma f16|bf16x3] > gpurun_out/mfma_shapes.txt`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sp-gan_amd")]
import torch   # noqa: E402

import bench   # noqa: E402
import spgan   # noqa: E402


class Shapes(bench.MfmaAccounting):
    def __call__(self, kind, a):
        done = super().__call__(kind, a)
        if done is not None:
            if kind == "gemm_nt":
                key = ("nt", a.M, a.N, a.K, "a%d e%d%s%s%s" % (a.a_mode, a.epi_mode, " stats" if a.stats else "", " pool" if a.pool_val else "", " z%d" % a.batch if a.batch > 1 else ""))
            elif kind == "gemm_tn":
                key = ("tn", a.M, a.Na, a.Nb, "b%d%s%s" % (a.b_mode, (" lazyA" if a.A2 else " affA") if a.a_scale else "", " defer" if a.defer_reduce else ""))
            elif kind == "gemm_dual":
                key = ("dual", a.M, a.Na, a.Nb, "dgrad+wgrad%s%s" % ({0: "", 1: " lazyA", 2: " actA"}.get(int(a.a_mode), ""), " edge" if a.e_idx else ""))
            else:
                key = ("knn", a.B * a.N, a.N, a.C, "k%d" % a.k)
            self.rec[-1] = self.rec[-1] + (key,)
        return done


def main():
    mode = sys.argv[sys.argv.index("--mfma") + 1] if "--mfma" in sys.argv else "f32"      # f32 | f16 | bf16x3
    f16 = mode == "f16"
    dev = torch.device("cuda", 0)
    spgan.ops.set_mfma_operands(mode)
    G, D = bench.build_models(dev)
    tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, graph=False)
    x, real, zs, alpha = bench.make_inputs(dev, 0, bench.PER_GPU_BATCH)
    for i in range(3):
        tr.step(x, real, zs[0], zs[1], alpha=alpha)
    peak = {"f32": bench.FP32_MATRIX_PEAK_TFLOPS, "f16": bench.FP16_MATRIX_PEAK_TFLOPS, "bf16x3": bench.FP16_MATRIX_PEAK_TFLOPS / 6.0}[mode]   # bf16x3: six bf16 MFMAs per product
    acct = Shapes(bench.PER_GPU_BATCH * bench.N_POINTS, peak, f16)
    busy = bench._KeepBusy(dev)
    steps = 4
    spgan.ops.launch_timer = acct
    for i in range(steps):
        busy()
        tr._eager_step(x, real, zs[0], zs[1], alpha=alpha)
    torch.cuda.synchronize()
    spgan.ops.launch_timer = None
    agg = {}
    for r in acct.rec:
        d = agg.setdefault(r[-1], [0, 0.0, 0.0, 0.0])
        d[0] += 1; d[1] += r[3].elapsed_time(r[4]); d[2] += r[1]; d[3] += r[2]
    tot = sum(v[1] for v in agg.values()) / steps
    print("# matrix-core launches of one WGAN-GP train step (B=32, N=2048, %s operands; util = of %.1f TFLOP/s): %.3f ms/step in %d launches" % (mode, peak, tot, len(acct.rec) // steps))
    print("%-4s %8s %6s %6s  %-22s %5s %9s %9s %8s %7s" % ("kind", "M", "N/Na", "K/Nb", "flavour", "n/st", "avg_us", "us/step", "TF(use)", "util"))
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        n = v[0] / steps
        avg = v[1] / v[0] * 1e3
        tf = v[2] / (v[1] * 1e-3) / 1e12
        print("%-4s %8d %6d %6d  %-22s %5.1f %9.1f %9.1f %8.1f %6.1f%%" % (key[0], key[1], key[2], key[3], key[4], n, avg, v[1] / steps * 1e3, tf, 100 * tf / acct.peak))


if __name__ == "__main__":
    main()
