import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, "/root/repo")
from safe_exploration_amd import SimpleGPModel, workload
N = int(sys.argv[1])
prob = workload.make_problem(6, N + 400, 2, 1, 8)
gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
Z, Y = prob["Z"], prob["Y"]
gp.train(Z[:N], Y[:N], opt_hyp=False)
for i in range(10):
    gp.update_model(Z[N+i:N+i+1], Y[N+i:N+i+1], opt_hyp=False, replace_old=False); gp.predict(Z[:1]); gp.information_gain()
pr = cProfile.Profile()
pr.enable()
for i in range(10, 210):
    gp.predict(Z[N+i:N+i+1])
    gp.update_model(Z[N+i:N+i+1], Y[N+i:N+i+1], opt_hyp=False, replace_old=False)
    gp.information_gain()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
