"""CPU, world_size 2 and 4 over gloo: the data-parallel path (spgan.parallel + TrainStep(distributed=True)) against its
contract (Generation/model.py:79-84): an N-rank step equals the single-process step on the concatenated batch with
per-replica BatchNorm -- the shards evaluated one after the other from the same weights, their gradients averaged (what
nn.DataParallel's reduce-add of the replica gradients of the full-batch-mean loss gives), one Adam update.

The single-process side runs the SAME harness segments (TrainStep._seg_d / _seg_g / _seg_opt_g) on W model copies and sums
their flat gradient buffers by hand where the distributed run calls dist.all_reduce."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BG, N = 8, 128            # global batch, points


class O:
    np = N; nk = 20; nz = 128; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False; small_d = False


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _models(salt):
    import spgan
    from oracle import spgan_oracle as orc
    from spgan import fixture_rng as fr
    G, D = spgan.Generator(O), spgan.Discriminator(O)
    G.load_state_dict({**G.state_dict(), **fr.init_params(orc.generator_shapes(), salt=salt)})
    D.load_state_dict({**D.state_dict(), **fr.init_params(orc.discriminator_shapes(), salt=salt)})
    return G, D


def _inputs():
    from spgan import fixture_rng as fr
    x = fr.sphere_template(256)[:N][None].repeat(BG, 1, 1)
    return x, fr.synthetic_real(BG, N, seed=7), fr.latent(BG, N, seed=8), fr.latent(BG, N, seed=9), fr.uniform("dp.alpha", (BG, 1, 1), 0.0, 1.0)


def _flat(module):
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()])


def _worker(rank, world, port, out, gan, use_gp, collective="all_reduce"):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      SPGAN_DP_COLLECTIVE=collective)
    from helpers import install_kernel_models
    import spgan
    install_kernel_models()
    torch.set_num_threads(2)
    assert spgan.init_process_group_from_env("gloo") == rank
    # different initial weights per rank on purpose: sync_params() must make them rank 0's
    G, D = _models(100 + rank)
    tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, distributed=True)
    sh = lambda t: spgan.shard_batch(t, rank, world).contiguous()
    info = tr.step(*[sh(t) for t in _inputs()], keep_grads=True)
    gD = torch.cat([g.reshape(-1) for g in info["d_grads"].values()])
    gG = torch.cat([g.reshape(-1) for g in info["g_grads"].values()])
    torch.save(dict(flatD=_flat(D), flatG=_flat(G), gD=gD, gG=gG, loss_d=info["loss_d"], loss_g=info["loss_g"],
                    bufD={k: v.clone() for k, v in D.state_dict().items() if "running" in k}), os.path.join(out, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _single_process_reference(world, gan, use_gp):
    """The same step on the concatenated batch in ONE process: W replicas from rank 0's weights, shard r on replica r (per-replica
    BatchNorm statistics), gradient buffers summed by hand, every replica's Adam applies sum/W."""
    import spgan
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)                                    # same CPU reduction orders as the workers
    reps = []
    for r in range(world):
        G, D = _models(100)                                     # rank 0's weights (what sync_params broadcasts)
        reps.append((G, D, spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, distributed=False)))
    ins = _inputs()
    sh = lambda t, r: spgan.shard_batch(t, r, world).contiguous()
    infos = [dict() for _ in range(world)]
    real_ts = []
    for r, (G, D, tr) in enumerate(reps):
        x, real, z_d, z_g, alpha = [sh(t, r) for t in ins]
        real_ts.append(tr._seg_d(x, real, z_d, alpha, False, infos[r]))
    gD_sum = sum(tr.optD.fp.grad for _, _, tr in reps)
    for _, _, tr in reps:
        tr.optD.fp.grad.copy_(gD_sum)
    for r, (G, D, tr) in enumerate(reps):
        x, real, z_d, z_g, alpha = [sh(t, r) for t in ins]
        tr._seg_g(x, real_ts[r], z_g, 1.0 / world, True, infos[r])
    gG_sum = sum(tr.optG.fp.grad for _, _, tr in reps)
    for r, (G, D, tr) in enumerate(reps):
        tr.optG.fp.grad.copy_(gG_sum)
        tr._seg_opt_g(1.0 / world, True, infos[r])
    torch.set_num_threads(nthreads)
    G0, D0, _ = reps[0]
    gD = torch.cat([g.reshape(-1) for g in infos[0]["d_grads"].values()])
    gG = torch.cat([g.reshape(-1) for g in infos[0]["g_grads"].values()])
    return dict(flatD=_flat(D0), flatG=_flat(G0), gD=gD, gG=gG, loss_d=[i["loss_d"] for i in infos], loss_g=[i["loss_g"] for i in infos],
                bufD=[{k: v.clone() for k, v in D.state_dict().items() if "running" in k} for _, D, _ in reps])


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("world,gan,use_gp,collective", [(2, "ls", False, "all_reduce"), (2, "wgan", True, "all_reduce"), (4, "wgan", True, "all_reduce"),
                                                         (2, "wgan", True, "one_hop")])
def test_n_rank_step_equals_single_process_on_concatenated_batch(tmp_path, world, gan, use_gp, collective):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), gan, use_gp, collective), nprocs=world, join=True)
    ranks = [torch.load(tmp_path / ("r%d.pt" % r)) for r in range(world)]
    for r in ranks[1:]:
        assert torch.equal(ranks[0]["flatD"], r["flatD"]) and torch.equal(ranks[0]["flatG"], r["flatG"]), "ranks diverged after one step"
        assert torch.equal(ranks[0]["gD"], r["gD"]) and torch.equal(ranks[0]["gG"], r["gG"])
    _setup_paths()
    from helpers import kernel_models
    with kernel_models():                  # the CPU doubles for this block only: the patch must not outlive the test (order independence)
        ref = _single_process_reference(world, gan, use_gp)
    a = ranks[0]
    # the all-reduced gradient is the mean of the per-shard gradients (sum order of gloo's reduction may differ for W > 2)
    assert _rel(a["gD"], ref["gD"]) <= 1e-6, _rel(a["gD"], ref["gD"])
    # G's gradient is taken through the UPDATED D: for W > 2 the other summation order of D's all-reduce flips the +-lr first Adam step of D's
    # noise-level gradient elements (see below), which G's gradient then sees -- a few 1e-6 of its norm (1.5e-6 observed); W = 2 is bit-identical
    assert _rel(a["gG"], ref["gG"]) <= (1e-6 if world == 2 else 1e-5), _rel(a["gG"], ref["gG"])
    if world == 2:
        assert torch.equal(a["gD"], ref["gD"]) and torch.equal(a["gG"], ref["gG"])          # a + b in both runs: bit-identical
        assert torch.equal(a["flatD"], ref["flatD"]) and torch.equal(a["flatG"], ref["flatG"])
    # parameters after Adam: an element moves by <= lr = 1e-4; equal up to the rounding of the gradient sum
    # (W = 2: bit-identical, asserted above).  Adam divides by sqrt(v) = |g| on the first step, so where a gradient element is
    # itself rounding noise (|g| within a few ulp of the sum's rounding error) a different summation order can flip the sign of a
    # +-lr step: such elements are compared through the gradient (above), the rest element-wise.
    for fa, fr_, g in ((a["flatD"], ref["flatD"], ref["gD"]), (a["flatG"], ref["flatG"], ref["gG"])):
        diff = (fa - fr_).abs()
        solid = g.abs() > 1e-6 * g.abs().max()
        assert diff[solid].max().item() <= 1e-6, diff[solid].max().item()
        assert diff.max().item() <= 2.01e-4 and (diff > 1e-6).float().mean().item() < 1e-3
    # per-replica losses and BatchNorm running statistics (each rank keeps the statistics of its own shard)
    for r in range(world):
        assert torch.allclose(ranks[r]["loss_d"], ref["loss_d"][r], rtol=1e-6, atol=1e-7)
        assert torch.allclose(ranks[r]["loss_g"], ref["loss_g"][r], rtol=1e-5, atol=1e-7)
        for k, v in ranks[r]["bufD"].items():
            assert torch.allclose(v, ref["bufD"][r][k], rtol=1e-6, atol=1e-7), k
    # and the shards really differ: the mean is not any single rank's gradient
    assert (ranks[0]["bufD"]["mlps.1.running_mean"] - ranks[1]["bufD"]["mlps.1.running_mean"]).abs().max().item() > 1e-6


def test_shard_batch_and_flat_allreduce_single_process():
    import spgan
    t = torch.arange(24).view(8, 3)
    assert torch.equal(spgan.shard_batch(t, 1, 4), t[2:4])
    with pytest.raises(ValueError):
        spgan.shard_batch(t, 0, 3)
    lin = torch.nn.Linear(4, 3)
    dp = spgan.DataParallel(lin)
    assert dp.module is lin and dp.world_size == 1 and dp.allreduce_grads() == 1.0
    assert lin.weight.grad is not None and lin.weight.grad.data_ptr() == dp.flat.grad.data_ptr()


def _hop_worker(rank, world, port, out):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from helpers import install_kernel_models
    import spgan
    install_kernel_models()
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for n_out in (5, 64, 1001):                         # flat sizes that do and do not divide by the world size
            sums = {}
            for kind in ("all_reduce", "one_hop"):
                lin = torch.nn.Linear(7, n_out, bias=False)
                tail = torch.nn.Parameter(torch.zeros(3))
                mod = torch.nn.Module(); mod.lin = lin; mod.tail = tail
                dp = spgan.DataParallel(mod, collective=kind)
                grads = torch.randn(dp.flat.grad.numel(), generator=torch.Generator().manual_seed(100 + rank))   # the flat buffer pads its members
                dp.flat.grad.copy_(grads)
                scale = dp.allreduce_grads()
                assert scale == 1.0 / world
                sums[kind] = dp.flat.grad.clone()
            sums["input"] = grads
            res[n_out] = sums
        torch.save(res, os.path.join(out, "hop%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_one_hop_collective_equals_all_reduce(world, tmp_path):
    """DataParallel(collective="one_hop") -- all-to-all + local sum in rank order (ops.reduce_chunks) + all-gather -- gives every rank
    the sums of dist.all_reduce: bit-identical across ranks, and equal to the all-reduce result up to its summation order."""
    port = _free_port()
    mp.spawn(_hop_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "hop%d.pt" % r)) for r in range(world)]
    for n_out in (5, 64, 1001):
        want = sum(ranks[r][n_out]["input"].double() for r in range(world))
        for r in range(world):
            assert torch.equal(ranks[r][n_out]["one_hop"], ranks[0][n_out]["one_hop"])          # every rank holds the same bits
            assert torch.allclose(ranks[r][n_out]["one_hop"].double(), want, rtol=0, atol=1e-5)
            assert torch.allclose(ranks[r][n_out]["one_hop"], ranks[r][n_out]["all_reduce"], rtol=0, atol=1e-5)


def test_collective_probe_selftest():
    """tools/collective_probe.py (the script for the first hour on a multi-GPU node: all_reduce vs one_hop at the step's two message
    sizes, and the step under both collectives with / without the generator forward issued under D's all-reduce) runs end to end on
    two gloo ranks and prints its JSON object -- a plumbing check, the numbers are meaningless here."""
    import json
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SPGAN_DP_COLLECTIVE", "SPGAN_DP_OVERLAP"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "collective_probe.py"), "--selftest"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["world"] == 2 and out[0]["selftest"] is True
    assert set(out[0]["messages"]) == {"G", "D"} and all("one_hop_us" in v and "all_reduce_us" in v for v in out[0]["messages"].values())
    assert len(out[0]["step_ms"]) == 4
