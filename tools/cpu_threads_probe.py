import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
for th in (8, 16, 32, 64):
    os.environ["RGCN_CPU_THREADS"] = str(th)
    r = bench.cpu_baseline(steps=1)
    print(th, r["value"], r["sample"][-14:], flush=True)
