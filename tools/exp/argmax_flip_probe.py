"""EXPERIMENT: the Discriminator's pool at the benchmarked size (golden G17 inputs): how many of the B x 1024 arg-max rows differ between the
exact-fp32 and the split-bf16 operand mode, and how close the competing values were (a near-tie resolved the other way moves the gradients
below the pool discretely -- tests/test_benchsize_golden_gpu.py bounds them by the REFERENCE's own float32-vs-float64 distance)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import spgan
from spgan import nets, ops, fixture_rng as fr
from oracle import spgan_oracle as orc
for tag, (B, N) in (("c2", (32, 2048)), ("c4", (16, 4096))):
    P = {k: v.cuda() for k, v in fr.init_params(orc.discriminator_shapes(), salt=17).items()}
    real = fr.synthetic_real(B, N, seed=171).transpose(2, 1).contiguous().cuda()
    out = {}
    for mode in ("f32", "bf16x3"):
        ops.set_mfma_operands(mode)
        pooled, ctx = nets.d_forward(P, None, real, training=True, update_running=False, head=False)
        out[mode] = (pooled.clone(), ctx["argmax"].clone(), ctx["yarg"].clone())
    ops.set_mfma_operands("f32")
    diff = (out["f32"][1] != out["bf16x3"][1])
    gap = (out["f32"][0] - out["bf16x3"][0]).abs()
    print("%s: %d of %d arg-max rows differ; pooled values differ by at most %.2e (rel %.2e); at the flipped entries the two pooled values differ by %s"
          % (tag, int(diff.sum()), diff.numel(), gap.max().item(), (gap / out["f32"][0].abs().clamp_min(1e-30)).max().item(),
             ["%.1e" % v for v in gap[diff].tolist()[:8]]))
