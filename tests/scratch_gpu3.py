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
    ok = np.array_equal(a, b)
    if not ok:
        bad += 1
        if bad < 10:
            d = np.nonzero(a != b)[0]
            print("MISMATCH", name, len(a), "first diff at", d[:5], a[d[:5]], b[d[:5]], flush=True)
for t in NUMERIC:
  for npb in [0,.1,1]:
    for order in [ASC, DESC]:
      for (lo,hi,n) in [(None,None,20000),(0,5,30000),(-3,3,500),(None,None,1),(None,None,0)]:
        a = random_arr(rng, t, n, npb, 3, lo=lo, hi=hi)
        if t in (FLOAT, DOUBLE) and n > 10:
            d = a.data.copy()
            d[rng.random(len(d))<0.05] = np.nan; d[rng.random(len(d))<0.05] = -0.0; d[rng.random(len(d))<0.05] = 0.0
            d[rng.random(len(d))<0.02] = np.inf; d[rng.random(len(d))<0.02] = -np.inf
            a = Arr(t, d, a.valid, 3, n)
        chk(('sort',t,npb,order,lo,hi,n), o.sort_indices(a,order), g.sort_indices(a,order))
      chunks=[random_arr(rng,t,m,npb,k,lo=-50,hi=50) for k,m in enumerate([100,0,3333,1,64,5000,0])]
      chk(('sortc',t,npb,order), o.sort_indices(chunks,order), g.sort_indices(chunks,order))
# full-range ints
for t in [INT32, UINT32, INT64, UINT64, INT16]:
    info=np.iinfo(NP_OF[t]); d=rng.integers(info.min, info.max, 100000, endpoint=True, dtype=NP_OF[t])
    a=Arr(t,d)
    for order in [ASC,DESC]: chk(('full',t,order), o.sort_indices(a,order), g.sort_indices(a,order))
print("TOTAL MISMATCHES", bad, flush=True)
import torch
from arrow1_amd import compute, device
n = 100_000_000
for name, t, tensor in [("uint64 full range", UINT64, torch.randint(-2**63, 2**63-1, (n,), dtype=torch.int64, device='cuda')),
                        ("float64 [0,1)", DOUBLE, torch.rand(n, dtype=torch.float64, device='cuda')),
                        ("int64 [-100,100]", INT64, torch.randint(-100,101,(n,),dtype=torch.int64,device='cuda'))]:
    arr = device.DeviceArray(t, tensor.view(torch.uint8), None, n, 0, 0)
    for it in range(2):
        torch.cuda.synchronize(); t0=time.time(); r = compute.sort_indices(g.ctx, arr); torch.cuda.synchronize(); dt=time.time()-t0
    idx = r.values[:n*8].view(torch.int64)
    sv = tensor.view(torch.int64 if t != DOUBLE else torch.float64)[idx]
    if t == UINT64:
        ok = bool(((sv[1:] ^ (-2**63)) >= (sv[:-1] ^ (-2**63))).all())
    else:
        ok = bool((sv[1:] >= sv[:-1]).all())
    print("sort_indices %s 100M: %.2f ms, %.1f Mrows/s sorted=%s" % (name, dt*1e3, n/dt/1e6, ok), flush=True)
