"""The reference's horse pair registered a few times (for rocprofv3 --kernel-trace --stats):
usage: python scripts/horse_trace.py [capacity] [repeats]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import icp_amd
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = os.path.join("tests", "golden", "data")
m, p = np.load(os.path.join(g, "horse_ref.npy")), np.load(os.path.join(g, "horse_tr1.npy"))
for k in range(rep):
    t0 = time.perf_counter()
    T, new_p, info = icp_amd.icp(m, p, 200, 1e-5, cap, return_info=True)
    print(f"run {k}: {1e3 * (time.perf_counter() - t0):.3f} ms, {info['iterations']} iterations", flush=True)
