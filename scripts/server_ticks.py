#!/usr/bin/env python3
"""Where a resident-server call spends its time: wall time of a blocking call, of a PING (mailbox round trip, nothing
evaluated) and the device's own clock for one evaluation (request seen -> results fenced), per kernel identifier.
GPU box:  python scripts/server_ticks.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, _lib  # noqa: E402
from call_latency import kern_hyp  # noqa: E402
from _timing import timeit  # noqa: E402


def main():
    print("# resident server: blocking __call__ / linearize_predict wall [us], PING round trip [us], device ticks of one evaluation [us]")
    for kt in ("rbf", "mat52", "lin_rbf", "lin_mat52"):
        for n_out, n_in, n_u, N in ((2, 2, 1, 100), (4, 3, 1, 25), (4, 3, 1, 150), (4, 4, 1, 150), (2, 2, 1, 350), (2, 2, 1, 500)):
            rng = np.random.default_rng(N)
            D = n_in + n_u
            Z = rng.uniform(-1, 1, (N, D))
            Y = rng.standard_normal((N, n_out))
            gp = SimpleGPModel(n_out, n_in, n_u, kern_types=[kt] * n_out, hyp=kern_hyp(kt, rng, D, n_out), device="cuda:0")
            gp.train(Z, Y, opt_hyp=False)
            if not gp.start_server(idle_timeout_s=0.05):
                continue
            x = rng.uniform(-0.5, 0.5, (1, D))
            st, ac = x[:, :n_in], x[:, n_in:]
            t_call = timeit(lambda: gp(st, ac), n=300, warmup=20, sync=False)
            t_lin = timeit(lambda: gp.linearize_predict(st, ac, True), n=300, warmup=20, sync=False)
            hd = gp._handle
            io = hd.single_io()
            out = np.zeros(8)
            po = ctypes.c_void_p(out.ctypes.data)
            ping = lambda: _lib.lib.sr_gp_server_call(hd.h, io["p_in"], 2, po, ctypes.c_double(5.0))
            gp(st, ac)
            ping()
            dev_us = out[0]
            t_ping = timeit(ping, n=300, warmup=20, sync=False)
            print("%-9s n_out=%d D=%d N=%4d Np=%3d  __call__ %5.1f  linearize %5.1f  ping %4.1f  evaluation on the device %4.1f"
                  % (kt, n_out, D, N, hd.Np, t_call, t_lin, t_ping, dev_us), flush=True)
            gp.stop_server()
            del gp


if __name__ == "__main__":
    main()
