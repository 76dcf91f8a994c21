#!/usr/bin/env python3
"""First hour on a multi-GPU node: which collective for the two flat gradient messages of a train step?

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 tools/collective_probe.py

Times, at the two message sizes of the step (G: 585,155 floats = 2.34 MB, D: 980,353 floats = 3.92 MB),
  * all_reduce           : one dist.all_reduce of the flat buffer (RCCL picks ring / tree),
  * rccl                 : spgan_allreduce_flat -- the library's own RCCL communicator, issued on the step's stream (GPU runs only),
  * one_hop              : all_to_all_single (reduce-scatter in one hop on the fully connected xGMI node) + spgan_reduce_chunks (local,
                           fixed order) + all_gather_into_tensor  (spgan.DataParallel(collective="one_hop")),
and the train step itself with SPGAN_DP_COLLECTIVE = all_reduce / one_hop and SPGAN_DP_OVERLAP = 1 / 0 (generator forward issued
under D's all-reduce or after it).  Rank 0 prints one JSON object; copy it to profiles/.  Without a GPU (or with --selftest) it
runs on gloo with tiny sizes: a plumbing check only (tests/test_dp_gloo.py::test_collective_probe_selftest)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sp-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    selftest = "--selftest" in sys.argv or not torch.cuda.is_available()
    import spgan
    if selftest:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import install_kernel_models
        install_kernel_models()
    rank = spgan.init_process_group_from_env("gloo" if selftest else "nccl")
    if not dist.is_initialized():
        raise SystemExit("start under torch.distributed.run with --nproc-per-node >= 2")
    world = dist.get_world_size()
    dev = torch.device("cpu") if selftest else torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    sizes = {"G": 585155, "D": 980353} if not selftest else {"G": 1031, "D": 2050}
    iters = 3 if selftest else 200

    class Holder(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(n, device=dev))

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        dist.barrier()

    out = {"world": world, "backend": dist.get_backend(), "iters": iters, "messages": {}}
    for name, n in sizes.items():
        res = {}
        for coll in ("all_reduce", "one_hop") + (() if selftest else ("rccl",)):
            m = Holder(n)
            dp = spgan.DataParallel(m, collective=coll)
            nf = dp.flat.grad.numel()                       # the flat buffer is padded to 16-byte multiples
            dp.flat.grad.copy_(torch.arange(nf, device=dev, dtype=torch.float32) * 1e-6 + rank)
            ref = dp.flat.grad.clone()
            for _ in range(3):
                dp.flat.grad.copy_(ref); dp.allreduce_grads()
            want = torch.arange(nf, device=dev, dtype=torch.float32) * 1e-6 * world + sum(range(world))
            assert torch.allclose(dp.flat.grad, want, rtol=1e-5, atol=1e-4), "%s: wrong sum" % coll
            sync()
            t0 = time.perf_counter()
            for _ in range(iters):
                dp.allreduce_grads()
            sync()
            res[coll + "_us"] = round((time.perf_counter() - t0) / iters * 1e6, 1)
        res["bytes"] = 4 * n
        out["messages"][name] = res
    # the train step under the four schedules
    B, N = (2, 128) if selftest else (32, 2048)

    class O:
        np = N; nk = 20; nz = 128; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False; small_d = False
    from spgan import fixture_rng as fr
    out["step_ms"] = {}
    for coll, overlap, single in [(c, o, "0") for c in ("all_reduce", "one_hop") for o in ("1", "0")] + ([] if selftest else [("rccl", "1", "0"), ("rccl", "1", "1")]):
        if True:
            os.environ["SPGAN_DP_COLLECTIVE"] = coll; os.environ["SPGAN_DP_OVERLAP"] = overlap; os.environ["SPGAN_DP_SINGLE_GRAPH"] = single
            torch.manual_seed(123)
            G, D = spgan.Generator(O).to(dev), spgan.Discriminator(O, num_point=N).to(dev)
            tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, distributed=True, graph=not selftest)
            x = fr.sphere_template(N if N in (512, 1024, 2048, 4096) else 256)[:N][None].repeat(B, 1, 1).to(dev)
            real = fr.synthetic_real(B, N, seed=1234 + rank).to(dev)
            z = [fr.latent(B, N, seed=1 + rank + i)[:, :1].contiguous().to(dev) for i in range(2)]
            alpha = fr.uniform("probe.alpha.%d" % rank, (B, 1, 1), 0.0, 1.0).to(dev)
            steps = 2 if selftest else 30
            for _ in range(2 if selftest else 8):
                tr.step(x, real, z[0], z[1], alpha=alpha)
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(x, real, z[0], z[1], alpha=alpha)
            sync()
            t = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out["step_ms"]["%s,overlap=%s%s" % (coll, overlap, ",single_graph" if single == "1" else "")] = round(t.item(), 3)
            del tr, G, D
    if rank == 0:
        if selftest:
            out["selftest"] = True
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
