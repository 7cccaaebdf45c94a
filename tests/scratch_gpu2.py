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
    ok = a.same(b) if isinstance(a, Result) else np.array_equal(a, b)
    if not ok:
        bad += 1
        if bad < 15: print("MISMATCH", name, a, b, flush=True)
# rerun cmp/arith quickly (stream fix)
for t in NUMERIC:
    a=random_arr(rng,t,3001,.2,3,lo=-5,hi=5); b=random_arr(rng,t,3001,.3,6,lo=-5,hi=5)
    for op in range(6): chk(('cmp',t,op),o.compare(op,a,b),g.compare(op,a,b))
    for op in range(3): chk(('ar',t,op),o.arith(op,a,b),g.arith(op,a,b))
print("cmp/arith", bad, flush=True)
for kt in [BOOL, INT8, UINT16, INT32, INT64, FLOAT, DOUBLE, UINT64]:
  for knp in [0,.1]:
    batches=[[random_arr(rng,kt,n,knp,off,lo=-20,hi=20)] for n,off in [(1000,0),(5000,3),(77,1)]]
    if kt in (FLOAT,DOUBLE):
        nb=[]
        for b in batches:
            d=b[0].data.copy(); d[:10]=[0.0,-0.0,np.nan,np.nan,1,1,-0.0,0.0,2,np.nan]
            nb.append([Arr(kt,d,b[0].valid,b[0].offset,b[0].length)])
        batches=nb
    io,ngo,uo=o.group_ids(batches); ig,ngg,ug=g.group_ids(batches)
    if ngo!=ngg: bad+=1; print("NG mismatch",kt,ngo,ngg)
    for a,b in zip(io,ig): chk(('ids',kt,knp),a,b)
    for (va,vb,nca),(vc,vd,ncb) in zip(uo,ug):
        chk(('uniq',kt), va.view(np.uint8) if va.dtype!=bool else va, vc.view(np.uint8) if vc.dtype!=bool else vc); chk(('uniqv',kt),vb,vd)
        if nca!=ncb: bad+=1; print("nc mismatch")
batches=[[random_arr(rng,INT32,2000,.1,1,lo=0,hi=5),random_arr(rng,INT8,2000,.1,2,lo=0,hi=3)]]
io,ngo,uo=o.group_ids(batches); ig,ngg,ug=g.group_ids(batches); assert ngo==ngg; chk('ids2',io[0],ig[0])
for (va,vb,nca),(vc,vd,ncb) in zip(uo,ug): chk('u2',va,vc); chk('u2v',vb,vd)
print("grouper", bad, flush=True)
for t in [BOOL]+NUMERIC:
  for vnp in [0,.3,1]:
    n=30000; ng=500
    v=random_arr(rng,t,n,vnp,5); ids=rng.integers(0,ng-5,n).astype(np.uint32)
    for fn in ['sum','count','mean']:
      for mode in ([0,1] if fn=='count' else [0]):
        a=o.hash_aggregate(fn,v,ids,ng,mode); b=g.hash_aggregate(fn,v,ids,ng,mode)
        if a.values.dtype==np.float64:
            ok = a.length==b.length and a.null_count==b.null_count and a.has_validity==b.has_validity and np.array_equal(a.validity_bits,b.validity_bits) and np.allclose(a.values,b.values,rtol=1e-12,atol=1e-9)
            if not ok: bad+=1; print("AGG F mismatch",fn,t,vnp,a,b)
        else: chk(('agg',fn,t,vnp,mode),a,b)
print("agg", bad, flush=True)
for vt in [INT64, DOUBLE, INT32, UINT8]:
  for kt in [INT32, INT64, INT8]:
    for knp in [0, .05]:
      v=random_arr(rng,vt,50000,.01,2); k=random_arr(rng,kt,50000,knp,1,lo=0,hi=3000); m=random_arr(rng,BOOL,50000,0.1,3)
      a=o.filter_groupby_sum(v,k,m); b=g.filter_groupby_sum(v,k,m)
      for i,(x,y) in enumerate(zip(a,b)):
        if x.dtype==np.float64:
          if not (x.shape==y.shape and np.allclose(x,y,rtol=1e-12)): bad+=1; print("fgs float mismatch",vt,kt)
        else: chk(('fgs',vt,kt,knp,i),x,y)
print("TOTAL MISMATCHES", bad)
# perf probe headline
import torch
from arrow1_amd import compute, device
n = 100_000_000
vals = device.DeviceArray(INT64, torch.randint(-100,100,(n,),dtype=torch.int64,device='cuda').view(torch.uint8), None, n, 0, 0)
keys = device.DeviceArray(INT32, torch.randint(0,1000000,(n,),dtype=torch.int32,device='cuda').view(torch.uint8), None, n, 0, 0)
mb = torch.randint(0,256,(n//8+64,),dtype=torch.uint8,device='cuda')
mask = device.DeviceArray(BOOL, mb, None, n, 0, 0)
bufs = compute.GroupByBuffers('cuda', 1<<20, INT32)
for it in range(3):
    torch.cuda.synchronize(); t0=time.time(); r = compute.filter_groupby(g.ctx, vals, keys, mask, buffers=bufs); torch.cuda.synchronize(); dt=time.time()-t0
    print("filter_groupby 100M -> %d groups, %.3f ms, %.1f Mrows/s, %.0f GB/s algorithmic" % (r.num_groups, dt*1e3, n/dt/1e6, 12.125*n/dt/1e9))
