import ctypes as C, os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[os.path.join(ROOT,'sp-gan_amd')]
import torch
from spgan import ops
lib=C.CDLL(os.path.join(ROOT,'tools/exp/libexp2.so'))
P=C.c_void_p; I=C.c_int
lib.exp_gemm_v2.argtypes=[P,I,P,I,P,I,I,I,I,P,I,P]
def run(M,N,K,var,reps=20,check=False):
    A=torch.randn(M,K,device='cuda'); W=torch.randn(N,K,device='cuda')*0.1; Y=torch.empty(M,N,device='cuda'); b=torch.randn(N,device='cuda')
    s=torch.cuda.current_stream().cuda_stream
    f=lambda: lib.exp_gemm_v2(A.data_ptr(),K,W.data_ptr(),K,Y.data_ptr(),N,M,N,K,b.data_ptr(),var,s)
    for _ in range(3): f()
    if check:
        ref=ops.gemm_nt(A,W,b); torch.cuda.synchronize(); print('  maxdiff', (Y-ref).abs().max().item())
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
def base(M,N,K,reps=20):
    A=torch.randn(M,K,device='cuda'); W=torch.randn(N,K,device='cuda')*0.1; b=torch.randn(N,device='cuda')
    for _ in range(3): ops.gemm_nt(A,W,b)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): ops.gemm_nt(A,W,b)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
run(1024,256,128,1,reps=1,check=True)
for (M,N,K,tag) in [(65536,1024,256,'D.L4'),(65536,256,1024,'dgrad4'),(65536,128,1280,'conv_out'),(655360,128,64,'conv_w3'),(65536,256,128,'D.L3')]:
    fl=2.0*M*N*K/1e9
    r=[run(M,N,K,v) for v in range(4)]
    b0=base(M,N,K)
    print('%-9s base %.3f (%.0f TF) | v2 sb,wps2 %.3f (%.0f) | v2 db,wps2 %.3f (%.0f) | sb,wps3 %.3f (%.0f) | db,wps1 %.3f (%.0f)' % (tag,b0,fl/b0,r[0],fl/r[0],r[1],fl/r[1],r[2],fl/r[2],r[3],fl/r[3]))
