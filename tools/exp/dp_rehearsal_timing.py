"""Per-step wall time and graph bookkeeping of the data-parallel TrainStep with two gloo ranks on one GPU (development aid).
torchrun --nproc-per-node 2 tools/exp/dp_rehearsal_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch, torch.distributed as dist
import bench, spgan
rank = spgan.init_process_group_from_env("gloo")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
G, D = bench.build_models(dev)
tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, distributed=True, graph=True, graph_warmup=3)
x, real, zs, alpha = bench.make_inputs(dev, rank, bench.PER_GPU_BATCH)
def sync(): torch.cuda.synchronize(); dist.barrier()
for i in range(10):
    sync(); t0 = time.time()
    tr.step(x, real, zs[i % len(zs)], zs[(i + 1) % len(zs)], alpha=alpha)
    sync()
    if rank == 0:
        print("step %d: %.1f ms  graph=%s recaptures=%d eager_calls=%d use_graph=%s" % (i, (time.time() - t0) * 1e3, tr._graph is not None, tr._recaptures, tr._eager_calls, tr.use_graph), flush=True)
# time the pieces of a replayed step
if tr._graph is not None and len(tr._graph) == 4:
    g1, gf, g2, g3 = tr._graph
    def t(fn):
        sync(); t0 = time.time(); fn(); torch.cuda.synchronize(); return (time.time() - t0) * 1e3
    parts = [("g1.replay", lambda: g1.replay()), ("allreduce D", lambda: tr.dpD.allreduce_grads()), ("gf.replay", lambda: gf.replay() if gf is not None else None),
             ("g2.replay", lambda: g2.replay()), ("allreduce G", lambda: tr.dpG.allreduce_grads()), ("g3.replay", lambda: g3.replay())]
    for n, fn in parts:
        ms = t(fn)
        if rank == 0: print("%-12s %.1f ms" % (n, ms), flush=True)

# replayed steps phase by phase (synchronising after each phase), several times
if tr._graph is not None and len(tr._graph) == 4:
    g1, gf, g2, g3 = tr._graph
    for it in range(6):
        row = []
        for n, fn in parts:
            t0 = time.time(); fn(); torch.cuda.synchronize(); row.append("%s %.1f" % (n.split(".")[0].replace("allreduce ", "ar"), (time.time() - t0) * 1e3))
        dist.barrier()
        if rank == 0: print("iter %d: " % it + " | ".join(row), flush=True)
    # the same without synchronising between the phases
    for it in range(6):
        sync(); t0 = time.time()
        g1.replay(); tr.dpD.allreduce_grads_begin(); gf.replay(); tr.dpD.allreduce_grads_end(); g2.replay(); tr.dpG.allreduce_grads(); g3.replay()
        t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
        if rank == 0: print("unsynchronised iter %d: issue %.1f ms, total %.1f ms" % (it, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
