import ctypes as C, os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[os.path.join(ROOT,'sp-gan_amd')]
import torch
from spgan import ops
lib=C.CDLL(os.path.join(ROOT,'tools/exp/libexp4.so'))
P=C.c_void_p; I=C.c_int
lib.exp_gemm_v3.argtypes=[P,I,P,I,P,I,I,I,I,P,I,P,I,P,P,P,P,P]
VARS=[int(v) for v in os.environ.get("VARS","0,3,4,7,9").split(",")]
MIX=int(os.environ.get("MIX","0"))
_big=torch.empty(256*1024*1024//4,device='cuda') if MIX else None
_big2=torch.empty_like(_big) if MIX else None
def timeit(f,reps=20):
    for _ in range(3): f()
    if MIX:   # every timed launch behind ~0.3 ms of HBM-bound copying: the clocks of a mixed workload instead of a pure MFMA loop
        ev=[]
        for _ in range(reps):
            _big2.copy_(_big); _big.copy_(_big2)
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record(); f(); e1.record(); ev.append((e0,e1))
        torch.cuda.synchronize(); return sum(a.elapsed_time(b) for a,b in ev)/reps
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
def mk(M,N,K):
    A=torch.randn(M,K,device='cuda'); W=torch.randn(N,K,device='cuda')*0.1; Y=torch.empty(M,N,device='cuda'); b=torch.randn(N,device='cuda')
    sc=torch.rand(K,device='cuda')+0.5; sh=torch.randn(K,device='cuda')*0.1
    st=torch.zeros(M//128,N,2,device='cuda'); pv=torch.zeros(M//128,N,2,device='cuda'); pa=torch.zeros(M//128,N,2,device='cuda',dtype=torch.int32)
    return A,W,Y,b,sc,sh,st,pv,pa
def run(M,N,K,var,mode,check=False):
    A,W,Y,b,sc,sh,st,pv,pa=mk(M,N,K)
    s=torch.cuda.current_stream().cuda_stream
    f=lambda: lib.exp_gemm_v3(A.data_ptr(),K,W.data_ptr(),K,Y.data_ptr(),N,M,N,K,b.data_ptr(),var,s,mode,sc.data_ptr(),sh.data_ptr(),st.data_ptr(),pv.data_ptr(),pa.data_ptr())
    rc=f()
    if rc!=0: return float('nan')
    if check:
        Ap=torch.nn.functional.leaky_relu(A*sc+sh,0.01) if mode&1 else A
        ref=Ap.double()@W.double().t()+b.double(); torch.cuda.synchronize()
        if mode&2:
            r=ref.view(M//128,128,N)
            print('  var',var,'mode',mode,'sum',(st[:,:,0].double()-r.sum(1)).abs().max().item(),'m2',((st[:,:,1].double()-r.var(1,unbiased=False)*128).abs().max()/ (r.var(1,unbiased=False)*128).abs().max()).item(),
                  'max',(pv[:,:,0].double()-r.max(1)[0]).abs().max().item(),'min',(pv[:,:,1].double()-r.min(1)[0]).abs().max().item(),
                  'arg', ((pa[:,:,0].long()-torch.arange(M//128,device='cuda')[:,None]*128)!=r.max(1)[1]).float().mean().item())
        else:
            print('  var',var,'mode',mode,'maxdiff', (Y-ref).abs().max().item())
    return timeit(f)
for v in VARS:
    for mode in range(4): run(1024,256,128,v,mode,check=True)
torch.backends.cuda.matmul.allow_tf32=False
SH=[(65536,1024,256,'D.L4'),(65536,256,256,'256x256'),(65536,256,128,'D.L3'),(65536,1280,128,'dT')]
for (M,N,K,tag) in SH:
    fl=2.0*M*N*K/1e9
    for mode in range(4):
        r=[run(M,N,K,v,mode) for v in VARS]
        print('%-9s mode %d (pro %d, %s) | ' % (tag,mode,mode&1,'stats+pool' if mode&2 else 'store') + ' '.join('v%d %6.1f(%5.1f)' % (v,x*1e3,fl/x) for v,x in zip(VARS,r)))
# production fused kernel (BN+LReLU prologue, statistics + pooling epilogue, nothing stored): the argument block of a real call is
# captured through the launch_timer hook and re-launched in the same hot loop as the experiments
import copy
from spgan import _lib
cap=[]
def grab(kind,a):
    if kind=="gemm_nt": cap.append(a)
    return None
for (M,N,K,tag) in SH:
    A,W,Y,b,sc,sh,st,pv,pa=mk(M,N,K)
    g=torch.ones(N,device='cuda'); be=torch.zeros(N,device='cuda')
    cap.clear(); ops.launch_timer=grab
    keep=ops.gemm_bn_pool(A,W,b,(g,be,None,None),2048,0.01,pro=(sc,sh,0.01))
    Yp=ops.gemm_nt(A,W,b)
    ops.launch_timer=None
    lib2=_lib.load(); s_=torch.cuda.current_stream().cuda_stream
    t1=timeit(lambda: lib2.spgan_gemm_nt(C.byref(cap[0]), s_))
    t2=timeit(lambda: lib2.spgan_gemm_nt(C.byref(cap[1]), s_))
    fl=2.0*M*N*K/1e9
    print('%-9s production (SPGAN_NT_FORCE=%s): fused pro+stats+pool %6.1f us (%5.1f TF) | plain store %6.1f us (%5.1f TF)' % (tag, os.environ.get("SPGAN_NT_FORCE","-"), t1*1e3, fl/t1, t2*1e3, fl/t2))
