import ctypes as C, os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib=C.CDLL(os.path.join(ROOT,'tools/exp/libexp5.so')); P=C.c_void_p; I=C.c_int
lib.exp_knn_pp.argtypes=[P,I,I,I,I,I,P,P]
B,N,Cn,k=32,2048,64,10
x=torch.randn(B*N,Cn,device='cuda'); idx=torch.empty(B*N,k,dtype=torch.int32,device='cuda'); s=torch.cuda.current_stream().cuda_stream
for abl in (0,1,2,4,5,6,7):
    for _ in range(2): lib.exp_knn_pp(x.data_ptr(),B,N,Cn,k,abl,idx.data_ptr(),s)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): lib.exp_knn_pp(x.data_ptr(),B,N,Cn,k,abl,idx.data_ptr(),s)
    e1.record(); torch.cuda.synchronize(); print('abl',abl,['full (anti-phase)','no selection','no mfma','staging+barriers only','group 1 idle','both groups in phase','anti-phase + setprio 1 on the MFMA phase','anti-phase + setprio 3'][abl],'%.1f us'%(e0.elapsed_time(e1)/10*1e3))
