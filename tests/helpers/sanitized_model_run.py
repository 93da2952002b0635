"""Run inside a process with libasan/libubsan preloaded (see tests/test_sanitizers.py): drives the AddressSanitizer +
UBSan builds of the oracle (latok_oracle.c) and of the CPU model (fused_model.cpp, which includes the product's
lane_math.h) over adversarial batches.  argv: <model.so> <oracle.so>"""
import ctypes as C, os, sys, random, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import ALPHABETS, random_strings, pack
M=C.CDLL(sys.argv[1]); O=C.CDLL(sys.argv[2])
M.fused_split_batch.argtypes=[C.c_void_p,C.c_void_p,C.c_int64,C.c_void_p,C.c_void_p,C.c_void_p]
O.oracle_split_batch.argtypes=[C.c_void_p,C.c_void_p,C.c_int64,C.c_void_p,C.c_void_p]
M.fused_block_mask.argtypes=[C.c_void_p,C.c_void_p,C.c_int64,C.c_void_p]
rng=random.Random(3)
for it in range(60):
    kind=rng.choice(list(ALPHABETS)); 
    texts=random_strings(rng, rng.randint(1,80), 0, rng.choice([5,300,9000]), ALPHABETS[kind])
    cps,row=pack(texts); total=int(row[-1])
    v1=np.zeros(total,np.uint8); b1=np.zeros((total+63)//64,np.uint64); v2=np.zeros(total,np.uint8); b2=np.zeros((total+63)//64,np.uint64)
    M.fused_split_batch(cps.ctypes.data,row.ctypes.data,len(texts),v1.ctypes.data,b1.ctypes.data,None)
    O.oracle_split_batch(cps.ctypes.data,row.ctypes.data,len(texts),v2.ctypes.data,b2.ctypes.data)
    assert np.array_equal(v1,v2) and np.array_equal(b1,b2)
    pb=np.zeros_like(b1); M.fused_split_batch(cps.ctypes.data,row.ctypes.data,len(texts),None,pb.ctypes.data,None); assert np.array_equal(pb,b2)
for it in range(200):
    n=rng.choice([1,2,64,65,4096,4097,9000]); a1=np.array([rng.random()<.05 for _ in range(n)],np.int8); a2=np.array([rng.random()<.1 for _ in range(n)],np.int8); out=np.zeros(n,np.int8)
    M.fused_block_mask(a1.ctypes.data,a2.ctypes.data,n,out.ctypes.data)
print('sanitized run ok')
