"""EXPERIMENT (data for DESIGN.md 13.6 item 0): could the max-pool of the Discriminator's 256 -> 1024 layer be found from a one-plane fp16 product
plus an exact recomputation of a few candidates?  For the layer's real operands at the benchmarked size (a3 = lrelu(bn3(y3)) of a train-mode pass,
W = fc2.0.weight): per (128-row tile t, channel c) the approximate tile maximum T[t,c] of the fp16-rounded operands' product and the rigorous bound
B[t,c] = 2^-10 * max_{m in t} ||a_m|| * ||w_c|| on its distance from the exact product; a tile can hold the shape's maximum only if
T + B >= max_t' (T - B).  Printed: how many of a shape's N/128 tiles qualify per channel (mean / median / 90 % / max), i.e. the share of the exact
product that would have to be recomputed -- for the maximum and the minimum (the pool needs both: the sign of the BatchNorm scale decides)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
from spgan import nets, ops, fixture_rng as fr
from oracle import spgan_oracle as orc

TILE = 128
for tag, (B, N), trained in (("c2 random init, real clouds", (32, 2048), 0), ("c2 after 30 D steps' worth of drift (weights scaled, biases shifted)", (32, 2048), 1),
                            ("c4 random init", (16, 4096), 0)):
    P = {k: v.cuda() for k, v in fr.init_params(orc.discriminator_shapes(), salt=17).items()}
    if trained:      # not a trained net (no data set here): a second, differently scaled parameter set, to see the sensitivity of the counts
        g = torch.Generator(device="cuda").manual_seed(5)
        P = {k: (v * (1.0 + 0.5 * torch.randn(v.shape, device="cuda", generator=g)) if v.dim() > 1 else v + 0.1 * torch.randn(v.shape, device="cuda", generator=g))
             for k, v in P.items()}
    real = fr.synthetic_real(B, N, seed=171).transpose(2, 1).contiguous().cuda()
    pooled, ctx = nets.d_forward(P, None, real, training=True, update_running=False, head=False)
    y3, (sc3, sh3, _, _) = ctx["ys"][2], ctx["bns"][2]
    a3 = torch.nn.functional.leaky_relu(y3 * sc3 + sh3, 0.2)
    W = P["fc2.0.weight"].reshape(1024, -1)
    M = B * N
    tiles = N // TILE
    na, nw = a3.norm(dim=1), W.norm(dim=1)
    a16, w16 = a3.half().float(), W.half().float()
    qual = {"max": [], "min": []}
    exact_in = 0
    for b in range(B):                       # one shape at a time: [N,1024] products
        rows = slice(b * N, (b + 1) * N)
        ya = (a16[rows] @ w16.t()).view(tiles, TILE, -1)
        ye = (a3[rows].double() @ W.double().t()).view(tiles, TILE, -1)
        bound = (2.0 ** -10) * na[rows].view(tiles, TILE).max(1).values[:, None] * nw[None, :]          # [tiles, C]
        assert ((ya.double() - ye).abs().amax(1) <= bound.double()).all(), "bound violated"
        for kind, sgn in (("max", 1.0), ("min", -1.0)):
            T = (sgn * ya).amax(1)                                  # approximate tile extreme, [tiles, C]
            lower = (T - bound).amax(0, keepdim=True)
            q = (T + bound >= lower)
            qual[kind].append(q.sum(0).float())
            te = (sgn * ye).amax(1)                                 # the exact extreme's tile must be among them
            exact_in += int((~q.gather(0, te.argmax(0, keepdim=True))).sum())
    print("== %s: B=%d N=%d (%d tiles of %d rows per shape)" % (tag, B, N, tiles, TILE))
    for kind in ("max", "min"):
        c = torch.cat(qual[kind])
        s = c.sort().values
        print("   %s: qualifying tiles per (shape, channel): mean %.2f  median %d  90%% %d  max %d of %d  -> %.1f %% of the exact product"
              % (kind, c.mean().item(), int(s[len(s) // 2]), int(s[int(len(s) * 0.9)]), int(s[-1]), tiles, 100.0 * c.mean().item() / tiles))
    print("   exact extreme outside the candidate tiles: %d (must be 0)" % exact_in)
