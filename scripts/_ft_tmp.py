import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from safe_exploration_amd import SimpleGPModel, workload
for N in (500, 1000, 2000):
    prob = workload.make_problem(4, N, 2, 1, 16)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    for _ in range(3):
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    torch.cuda.synchronize()
    print("N=%d train() wall %.1f us" % (N, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
    os.environ["SR_FACT_TRACE_ON"] = "1"
