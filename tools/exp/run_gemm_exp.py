import ctypes as C, os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[os.path.join(ROOT,'sp-gan_amd')]
import torch
from spgan._lib import GemmNTArgs
lib=C.CDLL(os.path.join(ROOT,'tools/exp/libexp.so')); lib.exp_gemm.argtypes=[C.POINTER(GemmNTArgs),C.c_int,C.c_void_p]
def run(M,N,K,abl,reps=20):
    A=torch.randn(M,K,device='cuda'); W=torch.randn(N,K,device='cuda')*0.1; Y=torch.empty(M,N,device='cuda'); b=torch.randn(N,device='cuda')
    a=GemmNTArgs(); a.A=A.data_ptr(); a.lda=K; a.W=W.data_ptr(); a.ldw=K; a.Y=Y.data_ptr(); a.ldy=N; a.M,a.N,a.K=M,N,K; a.bias=b.data_ptr()
    s=torch.cuda.current_stream().cuda_stream
    for _ in range(3): lib.exp_gemm(C.byref(a),abl,s)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): lib.exp_gemm(C.byref(a),abl,s)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for (M,N,K,tag) in [(65536,1024,256,'D.L4'),(65536,256,1024,'dgrad4'),(65536,128,1280,'conv_out'),(655360,128,64,'conv_w3')]:
    r=[run(M,N,K,abl) for abl in range(4)]
    fl=2.0*M*N*K/1e9
    print('%-9s full %.3f ms (%.1f TF) | no-gload %.3f | no-store %.3f | no-mfma %.3f' % (tag, r[0], fl/r[0], r[1], r[2], r[3]))
