#!/usr/bin/env python3
"""gemm_nt on the step's large shapes, split-bf16 operands on 256-row tiles (csrc/gemm_wide3.hip) against the other routes:
us per launch (hot loop and behind HBM-bound copies), TF fp32-equivalent, max error against an fp64 product.
usage: nt3_bench.py [--env K=V ...]   (re-runs itself once per --env setting: the kernel's experiment switches are read once per process)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]

SHAPES = [  # M, N, K, flavour
    (65536, 1024, 256, "pool"), (196608, 1024, 256, "pool"), (65536, 128, 1280, "plain"), (65536, 1280, 128, "plain"),
    (65536, 256, 256, "plain"), (131072, 256, 128, "plain"), (196608, 256, 128, "stats"), (131072, 128, 128, "plain"), (65536, 256, 256, "bnbwd"),
]


def main():
    import torch
    from spgan import ops
    big = torch.empty(256 * 1024 * 1024 // 4, device="cuda"); big2 = torch.empty_like(big)

    def timeit(f, mix, reps=12):
        for _ in range(3):
            f()
        if mix:
            ev = []
            for _ in range(reps):
                big2.copy_(big); big.copy_(big2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); ev.append((e0, e1))
            torch.cuda.synchronize()
            return sum(a.elapsed_time(b) for a, b in ev) / reps * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    torch.manual_seed(0)
    for (M, N, K, fl) in SHAPES:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
        sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.3
        gamma, beta = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
        ref_y = torch.randn(M, N, device="cuda") if fl == "bnbwd" else None

        def call():
            if fl == "pool":
                return ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 2048, 0.2, pro=(sc, sh, 0.2))
            if fl == "stats":
                return ops.gemm_nt(A, W, b, pro=(sc, sh, 0.2), stats=True)
            if fl == "bnbwd":
                return ops.gemm_nt_bnbwd(A, W, ref_y, gamma, beta, beta, gamma, 0.2, pro=(sc, sh, 0.2))
            return ops.gemm_nt(A, W, b)
        # fp64 reference on a row sample
        rows = torch.randint(0, M, (512,), device="cuda")
        Ad = A[rows].double()
        if fl != "plain":
            Ad = torch.nn.functional.leaky_relu(A[rows] * sc + sh, 0.2).double()      # the prologue itself in fp32, as the kernel evaluates it
        ref = Ad @ W.double().t() + (b.double() if fl != "bnbwd" else 0)
        row = []
        images = {}

        def provider(Wt):
            key = (Wt.data_ptr(), tuple(Wt.shape))
            if key not in images:
                images[key] = ops.split_image(Wt)
            return images[key]
        for name, mode, hint in (("f32", "f32", 0), ("x3-128", "bf16x3", 1), ("x3-wide", "bf16x3", 2), ("x3-wide+img", "bf16x3", 2)):
            ops.set_mfma_operands(mode)
            ops.w_image_provider = provider if name.endswith("+img") else None
            with ops.nt_tile_hint(hint):
                try:
                    out = call()
                except Exception as e:       # noqa: BLE001
                    row.append("%s: %s" % (name, type(e).__name__)); continue
                torch.cuda.synchronize()
                err = float("nan")
                if fl in ("plain", "stats"):
                    Y = out if fl == "plain" else out[0]
                    err = ((Y[rows].double() - ref).abs().max() / ref.abs().max()).item()
                elif fl == "pool":
                    pooled = out[2]
                    err = 0.0 if torch.isfinite(pooled).all() else float("inf")
                t0, t1 = timeit(call, 0), timeit(call, 1)
            tf = 2.0 * M * N * K / 1e6
            row.append("%s %6.1f / %6.1f us (%5.1f TF) err %.1e" % (name, t0, t1, tf / t0, err))
        ops.set_mfma_operands("f32"); ops.w_image_provider = None
        print("M=%6d N=%4d K=%4d %-5s | " % (M, N, K, fl) + " | ".join(row), flush=True)


if __name__ == "__main__":
    envs = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--env"]
    if envs and not os.environ.get("NT3_CHILD"):
        for e in envs:
            env = dict(os.environ); env["NT3_CHILD"] = "1"
            for kv in e.split(","):
                if kv:
                    k, v = kv.split("="); env[k] = v
            print("## " + (e or "(default)"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
    else:
        main()
