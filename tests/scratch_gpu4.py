import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orc
from orc import *
import gpu_api
o = orc.Api('orc'); g = gpu_api.GpuApi()
rng = np.random.default_rng(0x0ff1ce)
bad = 0
def chk(name, a, b):
    global bad
    if not np.array_equal(a, b):
        bad += 1
        if bad < 10: print("MISMATCH", name, len(a), len(b), a[:8], b[:8], flush=True)
n = 600_000
for vt in [INT64, DOUBLE, INT32, UINT8]:
  for kt, hi in [(INT32, 30000), (INT32, 200), (INT8, 100), (UINT16, 60000), (INT64, 1000)]:
    for knp in [0, .05]:
      for use_mask in [True, False]:
        v=random_arr(rng,vt,n,.01,2); k=random_arr(rng,kt,n,knp,1,lo=0,hi=hi); m=random_arr(rng,BOOL,n,0.1,3) if use_mask else None
        if m is None:
            mm = Arr(BOOL, np.ones(n, bool))
            a=o.filter_groupby_sum(v,k,mm)
        else:
            a=o.filter_groupby_sum(v,k,m)
        b=g.filter_groupby_sum(v,k,m)
        for i,(x,y) in enumerate(zip(a,b)):
          if x.dtype==np.float64:
            if not (x.shape==y.shape and np.allclose(x,y,rtol=1e-12)): bad+=1; print("fgs float mismatch",vt,kt)
          else: chk(('fgs',vt,kt,hi,knp,use_mask,i),x,y)
print("fast-path parity mismatches", bad, flush=True)
# overflow -> fallback: 600k distinct keys > 512*3584? no: 600k < 1.8M; force by many distinct keys in 3M rows
n2 = 3_000_000
v=random_arr(rng,INT64,n2,0,0); k=Arr(INT32, rng.permutation(n2).astype(np.int32)); m=Arr(BOOL, np.ones(n2,bool))
a=o.filter_groupby_sum(v,k,m); b=g.filter_groupby_sum(v,k,m)
for i,(x,y) in enumerate(zip(a,b)): chk(('overflow-fallback',i),x,y)
print("TOTAL MISMATCHES", bad, flush=True)
