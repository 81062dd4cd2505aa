#!/usr/bin/env python
"""Time `interpolate` on a full 512x288 frame with S planes of synthetic test-time renders (row N2, C5's
"time-interp x10").  Prints ms per interpolated frame and the accumulator-atomic rate."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsff_pl_amd as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=192)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--flow", type=float, default=0.02, help="NDC flow magnitude (0.02 ~ 5 px at 512 wide)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H, S = 512, 288, args.planes
    n = W * H
    g = torch.Generator(device=dev).manual_seed(0)
    K = torch.tensor([[400.0, 0, W / 2], [0, 400.0, H / 2], [0, 0, 1]])
    c2w = torch.eye(4)[:3]
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    zs = torch.linspace(0, 1, S, device=dev).expand(n, S).contiguous()
    # NDC points on the pixel rays (ray_utils.get_ndc_rays geometry for an identity pose)
    ox = -(xs.reshape(-1, 1) - W / 2) / (W / 2) * 0 + (xs.reshape(-1, 1) - W / 2) / (W / 2)
    oy = -(ys.reshape(-1, 1) - H / 2) / (H / 2)
    xyz = torch.stack([ox.expand(n, S), oy.expand(n, S), 2 * zs - 1], -1).contiguous()
    def res():
        return {"xyzs_fine": xyz, "zs_fine": zs,
                "static_rgbs_fine": torch.rand(n, S, 3, device=dev, generator=g),
                "static_alphas_fine": torch.rand(n, S, device=dev, generator=g) * 0.05,
                "transient_rgbs_fine": torch.rand(n, S, 3, device=dev, generator=g),
                "transient_alphas_fine": torch.rand(n, S, device=dev, generator=g) * 0.05,
                "transient_flows_fw": (torch.rand(n, S, 3, device=dev, generator=g) - 0.5) * 2 * args.flow,
                "transient_flows_bw": (torch.rand(n, S, 3, device=dev, generator=g) - 0.5) * 2 * args.flow}
    a, b = res(), res()
    for _ in range(2):
        A.interpolate(a, b, 0.3, K, c2w, (W, H))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        rgb, depth = A.interpolate(a, b, 0.3, K, c2w, (W, H))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.reps * 1e3
    atomics = 2 * n * S * 20
    print(f"interpolate 512x288x{S}: {ms:.2f} ms/frame, {atomics / ms / 1e6:.1f} G atomics/s, "
          f"rgb mean {float(rgb.mean()):.4f} finite {bool(torch.isfinite(rgb).all())}")


if __name__ == "__main__":
    main()
