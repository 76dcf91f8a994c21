#!/usr/bin/env python3
"""gemm_nt with fp16 operands on the step's large shapes: the row-pipelined 256-row-tile kernel (csrc/gemm_wide16.hip) against gemm_wide.hip's fp16
instantiation / the 128-row kernels (SPGAN_NT_WIDE16=0): us per launch in a hot loop, TF, error against the product of fp16-rounded operands.
usage: nt16_bench.py   (re-runs itself with SPGAN_NT_WIDE16=0 / 1 and the tile widths)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
SHAPES = [(65536, 1024, 256, "pool"), (196608, 1024, 256, "pool"), (65536, 128, 1280, "plain"), (65536, 1280, 128, "plain"), (65536, 256, 256, "plain"),
          (131072, 256, 128, "plain"), (196608, 256, 128, "stats"), (131072, 128, 128, "plain"), (65536, 256, 256, "bnbwd"), (65536, 128, 256, "bnbwd"), (65536, 320, 64, "plain")]


def main():
    import torch
    from spgan import ops
    ops.set_mfma_operands("f16")
    def timeit(f, reps=16):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    torch.manual_seed(0)
    out = []
    for (M, N, K, fl) in SHAPES:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
        sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.3
        gamma, beta = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
        ref_y = torch.randn(M, N, device="cuda") if fl == "bnbwd" else None
        def call():
            if fl == "pool": return ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 2048, 0.2, pro=(sc, sh, 0.2))
            if fl == "stats": return ops.gemm_nt(A, W, b, pro=(sc, sh, 0.2), stats=True)
            if fl == "bnbwd": return ops.gemm_nt_bnbwd(A, W, ref_y, gamma, beta, beta, gamma, 0.2, pro=(sc, sh, 0.2))
            return ops.gemm_nt(A, W, b)
        err = float("nan")
        if fl in ("plain", "stats"):
            rows = torch.randint(0, M, (512,), device="cuda")
            Ad = A[rows] if fl == "plain" else torch.nn.functional.leaky_relu(A[rows] * sc + sh, 0.2)
            ref = Ad.half().double() @ W.half().double().t() + b.double()
            o = call(); Y = o if fl == "plain" else o[0]
            err = ((Y[rows].double() - ref).abs().max() / ref.abs().max()).item()
        t = timeit(call)
        out.append("%dx%dx%d %-5s %6.1f us %6.1f TF err %.0e" % (M, N, K, fl, t, 2.0 * M * N * K / 1e6 / t, err))
    print("\n".join(out), flush=True)


if __name__ == "__main__":
    if not os.environ.get("NT16_CHILD"):
        for e in ("SPGAN_NT_WIDE16=0", "SPGAN_NT_WIDE16=1", "SPGAN_NT_WIDE16=1,SPGAN_NT16_TILE_N=128"):
            env = dict(os.environ, NT16_CHILD="1")
            for kv in e.split(","):
                k, v = kv.split("="); env[k] = v
            print("## " + e, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
    else:
        main()
