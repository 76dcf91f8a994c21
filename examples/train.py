#!/usr/bin/env python3
"""The reference's training loop (Generation/model.py:199-330) over the MI355X path, end to end on one GPU:
device-resident dataset -> sphere prior + latent noise drawn on the device -> TrainStep (hipGraph replay) -> checkpoints in the
reference's layout ({'G_model','G_optimizer','G_epoch'} / {'D_model','D_optimizer','D_epoch'}, model.py:505-525) -> sample dump.

    python examples/train.py --data chair.npy --np 2048 --bs 32 --epochs 2 --out runs/chair
    python examples/train.py --synthetic 256 --np 512 --bs 8 --epochs 1 --out /tmp/run      # no dataset needed
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]

import torch                                                   # noqa: E402
import spgan                                                   # noqa: E402
from spgan import fixture_rng as fr                            # noqa: E402
from spgan.dataset import DeviceDataset                        # noqa: E402
from spgan.sampling import InputSampler, save_xyz              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None, help=".npy/.npz/.h5 with [S, P, 3] clouds")
    ap.add_argument("--synthetic", type=int, default=0, help="use this many synthetic clouds instead of --data")
    ap.add_argument("--np", type=int, default=2048); ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--nz", type=int, default=128); ap.add_argument("--nk", type=int, default=20)
    ap.add_argument("--nv", type=float, default=0.2); ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--gan", default="ls"); ap.add_argument("--gp", action="store_true")
    ap.add_argument("--lr_g", type=float, default=1e-4); ap.add_argument("--lr_d", type=float, default=1e-4)
    ap.add_argument("--augment", action="store_true"); ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--out", default="runs/spgan"); ap.add_argument("--choice", default="chair")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--mfma", choices=("f32", "f16", "bf16x3"), default="f32", help="operand mode of the matrix products (INTEGRATION.md section 4a)")
    ap.add_argument("--lr_decay", action="store_true", help="StepLR on both optimisers, stepped once per epoch (Generation/model.py:99-110,309-312)")
    ap.add_argument("--lr_decay_feq", type=int, default=40); ap.add_argument("--lr_decay_rate", type=float, default=0.7)
    a = ap.parse_args()
    spgan.ops.set_mfma_operands(a.mfma)

    class Opts:                                                # the fields of Generation/config.py the modules read
        np = a.np; nk = a.nk; nz = a.nz; nv = a.nv; softmax = True; off = False; attn = False; use_head = False
        eql = False; z_norm = False; small_d = False; n_rand = False; n_mix = False

    dev = torch.device("cuda", 0)
    torch.manual_seed(123)                                     # model.py:38-41
    src = a.data if a.data else torch.cat([fr.synthetic_real(64, a.np, seed=i) for i in range((a.synthetic + 63) // 64)])[:a.synthetic or 64]
    data = DeviceDataset(src, num_points=a.np, batch_size=a.bs, scale=a.scale, augment=a.augment, device=dev, seed=0)
    G, D = spgan.Generator(Opts).to(dev), spgan.Discriminator(Opts, num_point=a.np).to(dev)
    step = spgan.TrainStep(G, D, gan=a.gan, use_gp=a.gp, lr_g=a.lr_g, lr_d=a.lr_d, graph=not a.no_graph)
    smp = InputSampler(Opts, device=dev, seed=1)
    x = smp.sphere_generator(a.bs)                             # model.py:231
    os.makedirs(a.out, exist_ok=True)
    it = 0
    scheds = [spgan.optim.StepLR(o, a.lr_decay_feq, a.lr_decay_rate) for o in (step.optG, step.optD)] if a.lr_decay else []
    for epoch in range(a.epochs):
        t0 = time.time()
        for real in data:
            info = step.step(x, real, smp.noise_generator(a.bs, compact=True), smp.noise_generator(a.bs, compact=True))
            it += 1
        torch.cuda.synchronize()
        print("epoch %d: %d steps, %.1f shapes/s, lossD %.4f lossG %.4f real_acc %.2f fake_acc %.2f" % (
            epoch, data.num_batches, data.num_batches * a.bs / (time.time() - t0), info["loss_d"].item(), info["loss_g"].item(),
            info["real_acc"].item(), info["fake_acc"].item()))
        tag = os.path.join(a.out, "%d_%s" % (epoch, a.choice))
        for sch in scheds:
            sch.step()
        torch.save({"G_model": G.state_dict(), "G_optimizer": step.optG.state_dict(), "G_epoch": epoch}, tag + "_G.pth")
        torch.save({"D_model": D.state_dict(), "D_optimizer": step.optD.state_dict(), "D_epoch": epoch}, tag + "_D.pth")
    G.eval()
    with torch.no_grad():
        out = G(x[:4], smp.noise_generator(4))
    for i in range(4):
        save_xyz(os.path.join(a.out, "sample", "%d.xyz" % i), out[i])
    print("wrote %s" % a.out)


if __name__ == "__main__":
    main()
