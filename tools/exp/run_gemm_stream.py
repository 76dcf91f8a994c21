"""EXPERIMENT: tools/exp/gemm_stream.hip against the product's gemm_nt on the step's short-K shapes (us per launch, TF, in a hot loop and behind HBM-bound copies)."""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'sp-gan_amd')]
so = os.path.join(ROOT, 'tools/exp/libgemm_stream.so')
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + ROOT + "/include", "-I" + ROOT + "/sp-gan_amd/csrc",
                    os.path.join(ROOT, "tools/exp/gemm_stream.hip"), "-o", so], check=True)
import torch
from spgan import ops
lib = C.CDLL(so)
P, I = C.c_void_p, C.c_int
lib.exp_gemm_stream.argtypes = [P, I, P, I, P, I, I, I, I, P, I, I, P]
big = torch.empty(256 * 1024 * 1024 // 4, device='cuda'); big2 = torch.empty_like(big)
def timeit(f, mix, reps=20):
    for _ in range(3): f()
    if mix:
        ev = []
        for _ in range(reps):
            big2.copy_(big); big.copy_(big2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); f(); e1.record(); ev.append((e0, e1))
        torch.cuda.synchronize(); return sum(a.elapsed_time(b) for a, b in ev) / reps * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
s = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(65536, 256, 128), (65536, 128, 128), (65536, 384, 64), (65536, 128, 64), (196608, 256, 128), (655360, 128, 64)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') * 0.1; b = torch.randn(N, device='cuda'); Y = torch.empty(M, N, device='cuda')
    ref = ops.gemm_nt(A, W, b)
    fl = 2.0 * M * N * K / 1e6
    row = []
    for mix in (0, 1):
        t = timeit(lambda: ops.gemm_nt(A, W, b), mix); row.append("prod %6.1f us (%5.1f TF)" % (t, fl / t))
        for dbl in (0, 1):
            for slots in ((256, 512) if K == 64 else (256,)):
                rc = lib.exp_gemm_stream(A.data_ptr(), K, W.data_ptr(), K, Y.data_ptr(), N, M, N, K, b.data_ptr(), dbl, slots, s)
                torch.cuda.synchronize()
                err = (Y - ref).abs().max().item()
                t = timeit(lambda: lib.exp_gemm_stream(A.data_ptr(), K, W.data_ptr(), K, Y.data_ptr(), N, M, N, K, b.data_ptr(), dbl, slots, s), mix)
                row.append("stream d%d s%d %6.1f us (%5.1f TF, rc %d, err %.1e)" % (dbl, slots, t, fl / t, rc, err))
        row.append("|")
    print("M=%6d N=%3d K=%3d  " % (M, N, K) + "  ".join(row))
