"""Cycles per phase of k_window_sort (a library built with -DWS_PHASES, see mkvar_file.sh):
    python3 profiles/bench_src/ws_phases.py [sort_u64|sort_f64]
Prints the average cycles per workgroup and phase: 0 load + stage, 1 run heads, 2 bucket atomics, 3 count scan,
4 bucket starts, 5 placement, 6 in-bucket sort, 7 row indices to LDS, 8 output."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "sort_u64"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = bench.device.Context(0)
n = 100_000_000
inp = bench.make_inputs(workload, n, dev, 0, 0.5)
step = bench.make_step(workload, ctx, inp, 0.5, 1, 0)
for _ in range(2):
    step()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
fn = C.CDLL(bench._abi.LIB_PATH).amdarrow_debug_sort_phases  # (the library is loaded already: same handle, same counters)
fn2 = C.CDLL(bench._abi.LIB_PATH).amdarrow_debug_onesweep_phases
fn(out)  # clears
fn2(out)
for _ in range(3):
    step()
torch.cuda.synchronize()
fn(out)
wgs = max(1, out[15])
print(workload, "workgroups", wgs, "cycles/wg per phase:", " ".join("%d:%.0f" % (i, out[i] / wgs) for i in range(9)),
      "total %.0f" % (sum(out[:9]) / wgs))
fn2(out)
wgs = max(1, out[15])
print(workload, "one-sweep workgroups", wgs, "cycles/wg per phase (0 loads + count, 1 totals / scan, 2 keys staged, 3 look-back, 4 barrier behind it, "
      "5 keys out, 6 indices staged, 7 indices out):", " ".join("%d:%.0f" % (i, out[i] / wgs) for i in range(8)), "total %.0f" % (sum(out[:8]) / wgs))
