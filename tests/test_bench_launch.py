"""bench.py --gpus N must be able to start its own ranks (the driver may run it without torchrun), and must also
work when torchrun already did.  CPU plumbing check with the stand-in step (gloo): spawn, rendezvous on 127.0.0.1,
barrier-bracketed timing, MAX over ranks, exactly ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_bench_self_launches_two_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--standin"], capture_output=True, text=True, timeout=600, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0 and "STAND-IN" in line["data"]
    # rank 0 explains the line: every rank's own time and the time inside the pixel all-gather, min / max over the ranks
    pr = line["per_rank"]
    assert 0 < pr["ms_per_step_min"] <= pr["ms_per_step_max"] <= line["ms_per_step"] * 1.001
    assert 0 < pr["gather_ms_per_step_min"] <= pr["gather_ms_per_step_max"] <= pr["ms_per_step_max"]


def test_bench_strong_scaling_form_launches_two_ranks():
    """`--workload eval --gpus N`: ONE 147 456-ray frame sharded by shard_bounds over the ranks + the (overlapped) pixel
    all-gather with uneven-shard counts -- the strong-scaling line of BASELINE.md section 4."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "eval", "--standin"], capture_output=True, text=True, timeout=600, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["rays_per_frame"] == 512 * 288
    assert line["value"] > 0 and "STAND-IN" in line["data"] and line["per_rank"]["gather_ms_per_step_max"] > 0


def test_bench_single_process_and_torchrun_forms():
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--standin"],
                         capture_output=True, text=True, timeout=600, env=_env())
    assert one.returncode == 0, one.stderr[-2000:]
    assert len(_json_lines(one.stdout)) == 1 and _json_lines(one.stdout)[0]["n_gpus"] == 1
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29655", os.path.join(ROOT, "bench.py"),
                         "--gpus", "2", "--steps", "2", "--warmup", "0", "--standin"],
                        capture_output=True, text=True, timeout=600, env=_env())
    assert tr.returncode == 0, tr.stderr[-2000:]
    lines = _json_lines(tr.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
    # torchrun with ONE rank: the process group exists, so the step still ends with the (one-rank) collective
    tr1 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29656", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "0", "--standin"],
                         capture_output=True, text=True, timeout=600, env=_env())
    assert tr1.returncode == 0, tr1.stderr[-2000:]
    lines = _json_lines(tr1.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["per_rank"]["gather_ms_per_step_max"] > 0
    assert "per_rank" not in _json_lines(one.stdout)[0]          # plain `python bench.py`: no group, no collective


def test_launcher_propagates_a_failing_rank():
    from nsff_pl_amd import dist as ndist
    code = "import os,sys,time; r=int(os.environ['RANK']); sys.exit(3) if r==1 else time.sleep(30)"
    rc = ndist.launch_local(2, [sys.executable, "-c", code], timeout=60)
    assert rc == 3
