"""EXPERIMENT: tools/exp/cosched.hip -- matrix-core-bound and HBM-bound workgroups in one launch against the two parts alone."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "tools/exp/libcosched.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tools/exp/cosched.hip"), "-o", so], check=True)
import torch
lib = C.CDLL(so)
P, I, SZ = C.c_void_p, C.c_int, C.c_size_t
lib.exp_cosched.argtypes = [P, I, I, P, P, SZ, I, I, P]
lib.exp_stream.argtypes = [P, P, SZ, I, P]
s = torch.cuda.current_stream().cuda_stream
n = 336 * 1024 * 1024 // 4
src = torch.randn(n, device="cuda"); dst = torch.empty_like(src); out = torch.empty(4096 * 256, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for lds in (66 * 1024, 33 * 1024):
    for a_blocks in (512, 256):
        iters = int(os.environ.get("ITERS", "6000")) * (1 if a_blocks == 512 else 2)      # the same MFMA work in total
        for b_blocks in (2048, 8192):
            ta = t(lambda: lib.exp_cosched(out.data_ptr(), iters, a_blocks, src.data_ptr(), dst.data_ptr(), n, 0, lds, s))
            tb = t(lambda: lib.exp_stream(src.data_ptr(), dst.data_ptr(), n, b_blocks, s))
            tb2 = t(lambda: lib.exp_cosched(out.data_ptr(), 0, 0, src.data_ptr(), dst.data_ptr(), n, b_blocks, lds, s))
            tab = t(lambda: lib.exp_cosched(out.data_ptr(), iters, a_blocks, src.data_ptr(), dst.data_ptr(), n, b_blocks, lds, s))
            fl = 2.0 * 32 * 32 * 2 * 8 * 4 * iters * a_blocks / 1e6
            print("lds %3d KB  A: %4d wgs %6.1f us (%5.1f TF)   B: %5d wgs alone %6.1f us (%.2f TB/s), in the co-kernel's footprint %6.1f us   A+B one launch %6.1f us  (sum %6.1f, max %6.1f)" %
                  (lds // 1024, a_blocks, ta, fl / ta, b_blocks, tb, 2 * n * 4 / tb / 1e6, tb2, tab, ta + tb, max(ta, tb)))
