import sys, time
sys.path.insert(0,'/root/repo/kv-match_amd'); sys.path.insert(0,'kv-match_amd')
import numpy as np, kvmatch, ctypes as C
from kvmatch import datagen
n=100_000_000
s=datagen.gen_series(n)
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(s)
    lib=ctx._lib
    for w in (25,50,100,200,400):
        for rep in range(2):
            f=C.c_void_p(); t=time.perf_counter()
            rc=lib.kvm_index_build(ctx._h, w, C.byref(f)); dt=time.perf_counter()-t
            info=[lib.kvm_index_file_info(f,i) for i in range(7)]
            lib.kvm_index_file_free(f)
        print("w=%d total %.3f s  step1 %.1f ms step2 %.1f ms encode %.1f ms  rows %d -> %d, intervals %d -> %d"%(w,dt,info[4]/1e3,info[5]/1e3,info[6]/1e3,info[0],info[1],info[2],info[3]),flush=True)
