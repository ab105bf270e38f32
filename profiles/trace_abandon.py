import sys; sys.path.insert(0,"kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n,m=100_000_000,8192
s=datagen.gen_series(n); qoff=n//3+17
planted=[n//7,n//2+1000,n-3*m]
q=datagen.plant_copies(s,qoff,m,planted,[1.0,1.0,1.0],[0.0,0.02,-0.03],0.01)
L,R=datagen.gen_intervals(n,m,1_000_000,64.0,must_cover=[qoff]+planted,seed=1)
c=kvmatch.Phase2Context(0); c.load_series(s)
for i in range(4):
    a=kvmatch.QueryEngine(c).verify((L,R),1,q,10.0); print("elapsed", a.elapsed_ms, len(a), file=sys.stderr)
