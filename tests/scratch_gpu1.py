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
t0=time.time()
for t in [BOOL]+NUMERIC:
  for vnp in [0, .1, 1]:
    for fnp in [0, .1, .999]:
      for tp in [0, .1, .5, 1]:
        for (vo, fo) in [(0,0),(3,4)]:
          for ns in [DROP, EMIT_NULL]:
            n = 5000
            v = random_arr(rng, t, n, vnp, vo); f = random_arr(rng, BOOL, n, fnp, fo, hi=tp)
            chk(('filter',t,vnp,fnp,tp,vo,fo,ns), o.filter(v,f,ns), g.filter(v,f,ns))
            if t == BOOL:
              assert o.filter_output_size(f,ns)==g.filter_output_size(f,ns)
              chk(('take_indices',fnp,tp,fo,ns), o.take_indices(f,ns), g.take_indices(f,ns))
print("filter done", bad, time.time()-t0, flush=True)
f = random_arr(rng, BOOL, 70000, .1, 3)
for ns in [DROP, EMIT_NULL]: chk('ti32', o.take_indices(f,ns), g.take_indices(f,ns))
for t in [BOOL, INT8, INT16, INT32, INT64, DOUBLE]:
  for it in INTS:
    for vnp in [0,.2]:
      for inp in [0,.2,1]:
        for (vo, io) in [(0,0),(5,3)]:
          n = 100 if it in (UINT8, INT8) else 3000
          v = random_arr(rng, t, n, vnp, vo); i = random_arr(rng, it, 1777, inp, io, lo=0, hi=n-1)
          chk(('take',t,it,vnp,inp), o.take(v,i), g.take(v,i))
v = random_arr(rng, INT32, 10)
for it, val in [(INT8,-1),(INT64,10),(UINT64,10),(INT16,11)]:
    i = Arr(it, [0,1,val,2])
    for api in (o, g):
        try: api.take(v,i); print("no error!?", api.prefix); bad+=1
        except StatusError as e: print(api.prefix, e)
print("take done", bad, flush=True)
for t in NUMERIC+[BOOL]:
  for (np1,np2) in [(0,0),(.2,0),(.2,.3),(1,0)]:
    a=random_arr(rng,t,3001,np1,3,lo=-5,hi=5); b=random_arr(rng,t,3001,np2,6,lo=-5,hi=5)
    if t in (FLOAT,DOUBLE): a.data[::17]=np.nan; a=Arr(t,a.data,a.valid,3,3001)
    for op in range(6):
      chk(('cmp',t,op),o.compare(op,a,b),g.compare(op,a,b))
      chk(('cmps',t,op),o.compare(op,a,1),g.compare(op,a,1))
      chk(('scmp',t,op),o.compare(op,1,b),g.compare(op,1,b))
    if t == BOOL: continue
    for op in range(8):
      for args in [(a,b),(a,3),(3,b)]:
        res=[]
        for api in (o,g):
          try: x=api.arith(op,*args); res.append((x.values.view(np.uint8).tolist(), x.validity_bits.tolist(), x.null_count, x.length))
          except StatusError as e: res.append((e.code,e.msg))
        if res[0]!=res[1]:
          bad+=1
          if bad<15: print("ARITH MISMATCH",t,op,[r if len(r)==2 else (r[2],r[3]) for r in res])
for t in INTS:
  info=np.iinfo(NP_OF[t]); a=Arr(t,[info.max,info.min,1,info.min],[1,1,1,0]); b=Arr(t,[1,1 if info.min==0 else -1,0,3])
  for op in range(8):
    res=[]
    for api in (o,g):
      try: x=api.arith(op,a,b); res.append((x.values.tolist(),x.validity().tolist(),x.null_count))
      except StatusError as e: res.append((e.code,e.msg))
    if res[0]!=res[1]: bad+=1; print("OVF MISMATCH",t,op,res)
print("TOTAL MISMATCHES", bad, time.time()-t0)
# quick perf probe
import torch
from arrow1_amd import compute, device
n = 100_000_000
vals = device.DeviceArray(INT64, torch.randint(-100,100,(n,),dtype=torch.int64,device='cuda').view(torch.uint8), (torch.randint(0,256,(n//8+64,),dtype=torch.uint8,device='cuda')|torch.randint(0,256,(n//8+64,),dtype=torch.uint8,device='cuda')), n, 0, -1)
for sel in [0.1,0.5,0.9]:
    m = (torch.rand(n, device='cuda') < sel)
    mb = torch.zeros(n//8+64, dtype=torch.uint8, device='cuda')
    w = torch.tensor([1,2,4,8,16,32,64,128],dtype=torch.uint8,device='cuda')
    mb[:n//8] = (m.view(-1,8).to(torch.uint8)*w).sum(1).to(torch.uint8)
    mask = device.DeviceArray(BOOL, mb, None, n, 0, 0)
    ctx = g.ctx
    for it in range(3):
        torch.cuda.synchronize(); t0=time.time(); r = compute.filter(ctx, vals, mask); torch.cuda.synchronize(); dt=time.time()-t0
    print("filter 100M sel=%.1f -> %d rows, %.3f ms, read %.0f GB/s total %.0f GB/s" % (sel, r.length, dt*1e3, 8.25*n/dt/1e9, (8.25+8.125*sel)*n/dt/1e9))
