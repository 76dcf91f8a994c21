"""CPU, world_size 2 over gloo: the data-parallel path (spgan.parallel + TrainStep(distributed=True)).
Two ranks each take half of the batch; after one step both ranks hold identical parameters, and the
all-reduced gradient equals the mean of the per-rank gradients (what nn.DataParallel's reduce-add of
replica gradients of the full-batch-mean loss gives, Generation/model.py:79-84; BN stays per replica)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import inspect
    import kernel_model as km
    import spgan
    import spgan.modules as modules
    import spgan.ops as ops
    for name, fn in inspect.getmembers(km, inspect.isfunction):
        if not name.startswith("_"):
            setattr(ops, name, fn)
    ops.SparseAffine = km.SparseAffine
    modules._require_gpu = lambda t, what: None
    from oracle import spgan_oracle as orc
    from spgan import fixture_rng as fr
    torch.set_num_threads(2)
    assert spgan.init_process_group_from_env("gloo") == rank

    class O:
        np = 128; nk = 20; nz = 128; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False; small_d = False
    G, D = spgan.Generator(O), spgan.Discriminator(O)
    # different initial weights per rank on purpose: sync_params() must make them rank 0's
    sd = G.state_dict(); G.load_state_dict({**sd, **fr.init_params(orc.generator_shapes(), salt=100 + rank)})
    sd = D.state_dict(); D.load_state_dict({**sd, **fr.init_params(orc.discriminator_shapes(), salt=100 + rank)})
    tr = spgan.TrainStep(G, D, gan="ls", distributed=True)
    Bg, N = 4, 128
    x = fr.sphere_template(256)[:N][None].repeat(Bg, 1, 1)
    real = fr.synthetic_real(Bg, N, seed=7)
    z_d, z_g = fr.latent(Bg, N, seed=8), fr.latent(Bg, N, seed=9)
    sh = lambda t: spgan.shard_batch(t, rank, world).contiguous()
    info = tr.step(sh(x), sh(real), sh(z_d), sh(z_g), keep_grads=True)
    flatD = torch.cat([p.detach().reshape(-1) for p in D.parameters()])
    flatG = torch.cat([p.detach().reshape(-1) for p in G.parameters()])
    gD = torch.cat([g.reshape(-1) for g in info["d_grads"].values()])
    torch.save(dict(flatD=flatD, flatG=flatG, gD=gD, local_gD=None), os.path.join(out, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["flatD"], b["flatD"]) and torch.equal(a["flatG"], b["flatG"]), "ranks diverged after one step"
    assert torch.equal(a["gD"], b["gD"])


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
