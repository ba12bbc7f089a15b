#!/usr/bin/env python3
"""Where the run kernel of the streamed MFMA route (sr_stream_mfma_kernel) spends its launch: per-workgroup time stamps of the
LAB build (100 MHz wall clock: start, first stage staged, loop done, partial product stored), summarised over the launch.
GPU box:  python scripts/stream_trace.py 5000 64"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: F401
import torch
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B, _lib
from _timing import timeit
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
Ts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64").split(",")]
lib = _lib.lib
fn = lib.sr_lab_stream_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
prob = workload.make_problem(9, N, 2, 1, max(Ts), sf2=0.01)
gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
gp.train(prob["Z"], prob["Y"], opt_hyp=False)
for T in Ts:
    x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device)
    wall = timeit(lambda: gp.predict_device(x, True), n=200, warmup=20)
    buf = np.zeros(8 * 4096, dtype=np.uint64)
    gp.predict_device(x, True); torch.cuda.synchronize()
    rc = fn(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    tr = buf.reshape(4096, 8)
    live = tr[:, 4] > 0
    tr = tr[live].astype(np.int64)
    t0 = tr[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    start, staged, loop, end, nsub, smid = us(tr[:, 0]), us(tr[:, 1]), us(tr[:, 2]), us(tr[:, 3]), tr[:, 4], tr[:, 5]
    print("N=%d T=%d  predict wall %.1f us; run kernel: %d workgroups on %d distinct CU ids, stages per workgroup %d..%d (sum %d)"
          % (N, T, wall, len(tr), len(set(smid.tolist())), nsub.min(), nsub.max(), nsub.sum()))
    q = lambda v: "min %.1f  median %.1f  max %.1f" % (v.min(), np.median(v), v.max())
    print("  start after the first workgroup's start [us]:   " + q(start))
    print("  prologue (start -> first stage in LDS) [us]:     " + q(staged - start))
    print("  per stage (loop time / stages) [us]:             " + q((loop - staged) / nsub))
    print("  loop [us]:                                       " + q(loop - staged))
    print("  epilogue (partial product stored) [us]:          " + q(end - loop))
    print("  end after the first workgroup's start [us]:      " + q(end))
    order = np.argsort(-nsub)
    for i in list(order[:3]) + list(order[-2:]):
        print("    workgroup of %d stages: start %.1f staged %.1f loop done %.1f end %.1f (CU id %d)"
              % (nsub[i], start[i], staged[i], loop[i], end[i], smid[i]))
    cus = {}
    for i in range(len(tr)):
        cus.setdefault(int(smid[i]), []).append(i)
    multi = [v for v in cus.values() if len(v) > 1]
    print("  CU ids holding more than one workgroup: %d (of %d)" % (len(multi), len(cus)))
