import cProfile, pstats, sys, io, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from safe_exploration_amd import SimpleGPModel, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
prob = workload.make_problem(4, N, 2, 1, 16)
gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
for _ in range(5): gp.train(prob["Z"], prob["Y"], opt_hyp=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): gp.train(prob["Z"], prob["Y"], opt_hyp=False)
torch.cuda.synchronize()
print("train() wall %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): gp.train(prob["Z"], prob["Y"], opt_hyp=False)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
