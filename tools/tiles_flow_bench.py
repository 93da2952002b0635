import ctypes as C, sys
import numpy as np
sys.path.insert(0,'/root/repo')
from latok_amd import _lib
lib=_lib.ensure_init()
for model,(lo,hi) in ((0,(64,192)),(1,(128,384))):
    n=1_000_000
    row=np.zeros(n+1,np.int64); _lib.check(lib.latok_corpus_offsets(0x1A70C0DE+model,0,n,lo,hi,row.ctypes.data)); total=int(row[-1]); w=(total+63)//64
    d_row=lib.latok_dev_alloc(row.nbytes); d_cps=lib.latok_dev_alloc(total*4); a=lib.latok_dev_alloc(w*8); b=lib.latok_dev_alloc(w*8)
    _lib.check(lib.latok_memcpy_h2d(d_row,row.ctypes.data,row.nbytes)); _lib.check(lib.latok_corpus_fill_device(0x1A70C0DE+model,model,0,n,d_row,d_cps,None)); _lib.check(lib.latok_sync())
    ms=C.c_float(0); mt=C.c_float(0); nf=C.c_int64(0)
    for r in range(4):
        _lib.check(lib.latok_bench_tiles_flow(d_cps,d_row,n,total,a,b,200,C.byref(ms)))
        _lib.check(lib.latok_bench_split_mask(d_cps,d_row,n,total,a,0,200,None,C.byref(mt),C.byref(nf)))
        alg=4*total+8*(n+1)
        print(f"model {model}: tiles kernel in the flow {ms.value/200*1e3:.2f} us/launch = {alg/(ms.value/200/1e3)/1e9/8000:.3f} of 8 TB/s | isolated {mt.value/200*1e3:.2f} us = {alg/(mt.value/200/1e3)/1e9/8000:.3f}", flush=True)
