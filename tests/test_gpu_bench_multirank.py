"""Rehearsal of bench.py's N > 1 control flow on the 1-GPU box: two ranks share device 0, the
per-shard top-k travel over gloo (VROD_BENCH_BACKEND=gloo), and the merged result of the last
batch must equal a single-device search of the whole corpus.  (The measured multi-GPU runs use
RCCL, one GPU per rank; this checks sharding, id offsets, the packed exchange, the merge and the
pipelined loop, not performance.)
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_bench_two_ranks_on_one_device(world):
    env = dict(os.environ, VROD_BENCH_BACKEND="gloo", VROD_BENCH_VERIFY="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29560 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--rows", "300001"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3
    assert out["verify_merged_equals_single_device"] is True
    assert out["exactness"]["certificate_fallback_queries"] == 0
