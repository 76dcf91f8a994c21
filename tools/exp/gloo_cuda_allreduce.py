"""gloo all-reduce of a 3.9 MB CUDA tensor between two ranks that share one GPU (development aid for the bench.py rehearsal mode)."""
import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
t = torch.ones(980353, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    w = dist.all_reduce(t, async_op=True); w.wait(); torch.cuda.synchronize()
    if dist.get_rank() == 0: print("all_reduce 3.9 MB cuda tensor via gloo: %.1f ms" % ((time.time() - t0) * 1e3), flush=True)
c = t.cpu()
for i in range(2):
    t0 = time.time(); dist.all_reduce(c)
    if dist.get_rank() == 0: print("cpu tensor: %.1f ms" % ((time.time() - t0) * 1e3), flush=True)
