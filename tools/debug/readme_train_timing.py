"""Time one training step of the reference's documented training configuration (README.md:226-233: use_viewdir, N_samples 128,
N_importance 0, batch_size 512) -- bench.py's aux.readme_train as a stand-alone script (for rocprofv3 runs).
    python tools/debug/readme_train_timing.py [steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                                    # noqa: E402


class _B:
    device = torch.device("cuda:0")


print(json.dumps(bench.readme_train(_B(), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 10), indent=1))
