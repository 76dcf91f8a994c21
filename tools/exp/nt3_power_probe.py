"""EXPERIMENT: the split-bf16 gemm_nt launch (M = 65536, N = 1024, K = 1024, output stored) on operands of different DATA: N(0,1), zeros, ones, N(0,1)
rounded to bf16 (mid = lo = 0) -- the bf16 matrix pipe is power-limited: DESIGN.md section 13.2."""
import os, sys, torch
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "sp-gan_amd")]
from spgan import ops
ops.set_mfma_operands("bf16x3")
img = {}
def prov(W):
    k = (W.data_ptr(), tuple(W.shape))
    if k not in img: img[k] = ops.split_image(W)
    return img[k]
ops.w_image_provider = prov
def timeit(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M, N, K = 65536, 1024, 1024
Y = torch.empty(M, N, device="cuda")
for name, A, W in (("randn", torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.1),
                   ("zeros", torch.zeros(M, K, device="cuda"), torch.zeros(N, K, device="cuda")),
                   ("ones", torch.ones(M, K, device="cuda"), torch.ones(N, K, device="cuda")),
                   ("bf16-exact randn (mid = lo = 0)", torch.randn(M, K, device="cuda").bfloat16().float(), (torch.randn(N, K, device="cuda") * 0.1).bfloat16().float())):
    img.clear()
    with ops.nt_tile_hint(2):
        t = timeit(lambda: ops.gemm_nt(A, W, None, out=Y))
    print("%-40s %7.1f us  %6.1f TF" % (name, t, 2.0 * M * N * K / 1e6 / t), flush=True)
