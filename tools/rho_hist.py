"""Iteration histogram of the fixed-rho ADMM (CPU oracle, 8 threads) for candidate step sizes.
usage: rho_hist.py N B seed_offset rho@pos:vel:acc:input ...  (development aid)"""
import sys, time, itertools
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from concurrent.futures import ThreadPoolExecutor
O.build()
SEED=0x4D5043
N=int(sys.argv[1]); B=int(sys.argv[2]); seed=SEED+int(sys.argv[3])
def run(seed,rho,cls,first=0):
    O.set_rho_class_scale(*cls)
    p=O.mpc6_params(N=N); x0,ref=O.synth_mpc6(seed,N,0.1,B,first)
    s=O.fixed_rho_settings(rho,warm_start=0,linsys=1)
    e=np.linspace(0,B,9).astype(int)
    with ThreadPoolExecutor(8) as ex: parts=list(ex.map(lambda k:O.mpc6_batch(p,s,x0[e[k]:e[k+1]],ref[e[k]:e[k+1]]),range(8)))
    it=np.concatenate([q["iters"] for q in parts]); st=np.concatenate([q["status"] for q in parts])
    return st, it
for spec in sys.argv[4:]:
    rho, cls = spec.split("@")
    cls=tuple(float(x) for x in cls.split(":"))
    st,it=run(seed,float(rho),cls)
    u,c=np.unique(it,return_counts=True)
    print(rho,cls,"solved %.4f mean %.2f"%((st==1).mean(), it.mean()), dict(zip(u.tolist(),c.tolist())), flush=True)
