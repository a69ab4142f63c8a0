"""bench.py keeps its contract: ONE JSON line on rank 0 with the agreed keys, for one process and -- rehearsed on this
one-GPU box over gloo (PTRT_BENCH_REHEARSE=1: every rank renders on cuda:0, host-staged gather; not a measurement) -- for
the N > 1 path with both ways of cutting the frame."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}
SMALL = ["--width", "320", "--height", "200", "--steps", "3", "--warmup", "1"]


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_single_process_line():
    d = _line(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL], cwd=ROOT, stderr=subprocess.DEVNULL))
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["metric"] == "Mrays/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and d["scaling"] == "strong"
    assert d["config"]["workload"].startswith("cornell 320x200 4spp 4-bounce") and d["config"]["parallelism"] == "single"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert r["valu_issue_frac"] is None  # (profile-derived fields only for the exact workload the profile was taken on)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["single_thread"]["cores"] == 1
    assert "configs3" not in d  # only the default (headline) run carries the 4K split


@pytest.mark.parametrize("layout", ["strips", "bands"])
def test_two_rank_rehearsal(layout):
    env = dict(os.environ, PTRT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541" if layout == "strips" else "29542", os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL,
           "--layout", layout]
    d = _line(subprocess.check_output(cmd, cwd=ROOT, env=env, stderr=subprocess.DEVNULL, timeout=300))
    assert KEYS <= set(d) and d["n_gpus"] == 2 and "rehearsal" in d and d["config"]["parallelism"] == f"tile2-{layout}"
    assert d["rays_per_frame"] > 320 * 200 * 4  # both ranks' rays


def test_plain_invocation_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE (how the driver starts it): the process -- before it
    imports anything that touches a device -- starts the two ranks itself (python -m torch.distributed.run as a child, a free
    rendezvous port on 127.0.0.1), passes rank 0's line through and exits with the child's code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PTRT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
    assert r.returncode == 0
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and "rehearsal" in d and d["config"]["parallelism"] == "tile2-strips"
    assert d["config"]["ramp_frames"] >= 10
