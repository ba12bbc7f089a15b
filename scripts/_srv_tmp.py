import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from safe_exploration_amd import SimpleGPModel, workload
from safe_exploration_amd._lib import lib
prob = workload.make_problem(9, 100, 2, 1, 4, sf2=0.01)
gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
gp.train(prob["Z"], prob["Y"], opt_hyp=False)
assert gp.start_server(0.05)
hd = gp._handle; io = hd.single_io()
io["h_in_np"][:] = np.hstack((prob["p"][0], prob["k_ff"][0]))
def t(fn, n=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("raw C call, first order: %.1f us" % t(lambda: lib.sr_gp_server_call(hd.h, io["p_in"], 0, io["p_srv"], 5.0)))
print("raw C call, ping:        %.1f us" % t(lambda: lib.sr_gp_server_call(hd.h, io["p_in"], 2, io["p_srv"], 5.0)))
lib.sr_gp_server_call(hd.h, io["p_in"], 0, io["p_srv"], 5.0)
lib.sr_gp_server_call(hd.h, io["p_in"], 2, io["p_srv"], 5.0)
print("device-side evaluation time of the last query: %.2f us" % io["srv_np"][0])
print("python __call__: %.1f us" % t(lambda: gp(prob["p"][:1], prob["k_ff"][:1])))
