#!/usr/bin/env python
"""Micro-benchmark of the field kernel alone (A/B of kernel variants on the GPU box).

    NSFF_LIB=path/to/variant.so python tools/bench_field.py [--rays 1024] [--iters 10]

Times the three launch shapes of a C2 train-mode call (coarse s+t, fine s+t+flows, warp t-only)
with HIP events (nsff_prof_*) and prints TFLOP/s per shape and for the C2 mix.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import nsff_pl_amd as A  # noqa: E402
from nsff_pl_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"])
    ap.add_argument("--tile-points", type=int, default=0)
    args = ap.parse_args()
    from nsff_pl_amd import config
    config.set_precision(args.precision)
    config.set_tile_points(args.tile_points)
    peak = {"f32": 157.3e12, "f16x3": 2500e12}[args.precision]
    dev = "cuda:0"
    cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=args.rays, seed=0)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    for m in list(models.values()) + [emb["t"]]:
        m.to(dev)
    freqs = [float(f) for f in emb["xyz"].freqs]
    n = args.rays
    t_emb = emb["t"](torch.randint(0, 30, (n,), device=dev)).detach().contiguous()
    shapes = {"coarse_s+t": ("coarse", 64, 2, 2, 0), "fine_s+t+flow": ("fine", 192, 2, 2, 2),
              "warp_t": ("fine", 192, 0, 2, 1)}
    res = {}
    for name, (typ, S, sm, tm, fh) in shapes.items():
        P = n * S
        xyz = (torch.rand(P, 3, device=dev) * 2 - 1).contiguous()
        raw = torch.empty(P, 16, device=dev)
        call = lambda: _lib.field_query(models[typ], raw, P, S, static_mode=sm, transient_mode=tm,
                                        flow_heads=fh, xyz=xyz, freqs=freqs, t_emb=t_emb)
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(args.iters):
            call()
        torch.cuda.synchronize()
        k, ms, fl, _ex, ghz = _lib.prof_collect()
        _lib.prof_enable(False)
        res[name] = (ms / k, fl / k)
        print(f"{name:14s} P={P:7d}  {ms / k:8.3f} ms  {fl / (ms * 1e-3) / 1e12:7.2f} TFLOP/s  clock {ghz if ghz is None else round(ghz, 3)} GHz")
    ms = res["coarse_s+t"][0] + res["fine_s+t+flow"][0] + 2 * res["warp_t"][0]
    fl = res["coarse_s+t"][1] + res["fine_s+t+flow"][1] + 2 * res["warp_t"][1]
    print(f"C2 mix [{args.precision}] {ms:8.3f} ms  {fl / (ms * 1e-3) / 1e12:7.2f} TFLOP/s  ({fl / (ms * 1e-3) / peak:.3f} of the dense MFMA peak "
          f"of the dtype, avg launch {ms / 4:.4f} ms)")


if __name__ == "__main__":
    main()
