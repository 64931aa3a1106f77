import sys, os, time
sys.path.insert(0, "bayes-bridge_amd")
import numpy as np, torch
from bayesbridge_amd import HipSparseDesignMatrix, simulate, _lib
from ctypes import c_void_p
n,p,f = 1000000, 50000, .002
indptr, indices = simulate.simulate_binary_csr_device(n,p,f,seed=111)
nnz = indices.numel()
data = torch.randn(nnz, dtype=torch.float64, device='cuda')
offset = torch.zeros(p, dtype=torch.float64, device='cuda')
for storage in ('tiled','csr'):
    d = HipSparseDesignMatrix.from_device_csr(n,p,nnz,indptr.data_ptr(),indices.data_ptr(),data.data_ptr(),offset.data_ptr(),add_intercept=True,device=0,storage=storage)
    lib=_lib.load()
    v=torch.randn(p+1,dtype=torch.float64,device='cuda'); w=torch.randn(n,dtype=torch.float64,device='cuda')
    on=torch.empty(n,dtype=torch.float64,device='cuda'); oP=torch.empty(p+1,dtype=torch.float64,device='cuda')
    torch.cuda.synchronize()
    for _ in range(3):
        lib.bbx_design_dot_dev(d.handle,c_void_p(v.data_ptr()),c_void_p(on.data_ptr())); lib.bbx_design_tdot_dev(d.handle,c_void_p(w.data_ptr()),c_void_p(oP.data_ptr()))
    d.synchronize(); d.set_timing(True); d.reset_timing()
    for _ in range(50):
        lib.bbx_design_dot_dev(d.handle,c_void_p(v.data_ptr()),c_void_p(on.data_ptr())); lib.bbx_design_tdot_dev(d.handle,c_void_p(w.data_ptr()),c_void_p(oP.data_ptr()))
    tm=d.get_timing(); db,tb=d.matvec_bytes
    for name,b in (("dot",db),("tdot",tb)):
        cnt,ms=tm[name]; print(storage,name,"%.4f ms"%(ms/cnt),"%.1f MB"%(b/1e6),"%.0f GB/s"%(b/(ms/cnt)/1e6))
    X=torch.sparse_csr_tensor(indptr.long(),indices.long(),data,size=(n,p))
    ref=v[0]+X@v[1:]
    print(" err", float((on-ref).abs().max()))
    del d
