"""kvm_index_build at n = 1e8 for the five window lengths of the reference's index set: total, step 1 (GPU + copy), step 2 (host
merge walk), encode; minimum and median of 7 calls (the host threads share the box with other jobs: single calls scatter by 25 %)."""
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
        tot=[]; st=[]
        for rep in range(8):
            f=C.c_void_p(); t=time.perf_counter()
            rc=lib.kvm_index_build(ctx._h, w, C.byref(f)); dt=time.perf_counter()-t
            info=[lib.kvm_index_file_info(f,i) for i in range(7)]
            lib.kvm_index_file_free(f)
            if rep: tot.append(dt); st.append(info[4:7])
        st=np.array(st)/1e3
        print("w=%d total min %.3f median %.3f s  step1 %.1f ms step2 min %.1f median %.1f ms encode %.1f ms  rows %d -> %d, intervals %d -> %d"%(w,min(tot),np.median(tot),np.median(st[:,0]),st[:,1].min(),np.median(st[:,1]),np.median(st[:,2]),info[0],info[1],info[2],info[3]),flush=True)
