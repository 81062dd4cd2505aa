"""Debug: device kernels / device time of one eager training step by phase (forward, loss, backward, optimizer)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
import bench
from nsff_pl_amd import config, field_grad

config.set_precision("f16x3")
dev = torch.device("cuda:0")
b = bench.Bench(0, 1, dev)
b.to_device()
step = b.train_step()
tr = b.trainer
batch = {k: v.to(dev) for k, v in b.scenes.synthetic_targets(bench.N_RAYS, b.ts.cpu(), 100).items()}
batch["rays"] = b.rays
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()


def one():
    with record_function("P0_zero"):
        tr.optimizer.zero_grad(set_to_none=True)
    kwargs = dict(output_transient=tr.output_transient, output_transient_flow=tr.output_transient_flow)
    with record_function("P1_forward"):
        results = tr.forward(batch["rays"], batch.get("ts"), **kwargs)
    with record_function("P2_loss"):
        loss_d = tr.loss(results, batch, epoch=0, **kwargs)
        loss = sum(loss_d.values())
    with record_function("P3_backward"):
        with field_grad.deferred_weight_grads():
            loss.backward()
    with record_function("P4_adam"):
        tr.optimizer.step()


with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        one()
    torch.cuda.synchronize()
ev = prof.events()
ranges = [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("P") and e.name[1].isdigit() and e.device_type.name == "CPU"]
# map each device kernel to the CPU range that launched it via correlation: use cpu-side launch events (children)
stats = collections.defaultdict(lambda: [0, 0.0])
names = collections.defaultdict(lambda: collections.Counter())
for e in ev:
    if e.device_type.name != "CPU":
        continue
    if not e.kernels:
        continue
    for (n, s, t) in ranges:
        if s <= e.time_range.start <= t:
            for k in e.kernels:
                stats[n][0] += 1
                stats[n][1] += k.duration
                names[n][k.name[:60]] += 1
            break
for n in sorted(stats):
    print(f"{n:12s} kernels/step {stats[n][0] / 3:7.1f}   device us/step {stats[n][1] / 3:9.1f}")
    for k, c in names[n].most_common(8):
        print(f"      {c / 3:6.1f}  {k}")
