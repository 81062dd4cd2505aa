"""Debug: is NSFFTrainer(graph=...) reproducible run to run in one process (no process group)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_dist_gpu as T
import nsff_pl_amd as A
A.set_precision("f16x3")
for graph in (False, True, True, True, False):
    losses, params = T._train(graph, steps=4)
    print("graph" if graph else "eager", " ".join(f"{l:.7f}" for l in losses), f"|p|={float(params.abs().sum()):.6f}", flush=True)
