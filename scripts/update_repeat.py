import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from safe_exploration_amd import SimpleGPModel, workload
N = int(sys.argv[1]); reps = int(sys.argv[2])
prob = workload.make_problem(4, N, 2, 1, 16)
gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
out = []
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    torch.cuda.synchronize()
    out.append(time.perf_counter() - t0)
fl = 2 * (2.0 / 3.0) * float(N) ** 3
print("N=%d" % N, " ".join("%.3fs(%.1fTF)" % (t, fl / t / 1e12) for t in out), flush=True)
