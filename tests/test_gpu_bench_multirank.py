"""Rehearsal of bench.py's N > 1 control flow on the 1-GPU box: two ranks share device 0, the
per-shard top-k travel over gloo (VROD_BENCH_BACKEND=gloo), and the merged result of the last
batch must equal a single-device search of the whole corpus.  Every N > 1 line must carry the
metric's recall / exactness fields WITHOUT any opt-in switch (VROD_BENCH_VERIFY unset): the driver
sets none.  (The measured multi-GPU runs use
RCCL, one GPU per rank; this checks sharding, id offsets, the packed exchange, the merge and the
pipelined loop, not performance.)
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assert_multi_gpu_fields(out):
    """BASELINE.json's metric is QPS + recall@10 vs the CPU reference at 1/2/4/8 GPUs: the three fields of an N > 1 line"""
    assert out["verify_merged_equals_single_device"] is True
    assert "hipcc" in out["library"]          # vrod_version(): the compiler that built the device code
    assert out["recall_at_10"] == 1.0
    assert out["bit_exact_vs_oracle_on_sample"] is True
    assert "oracle_sample" in out and "parity" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_bench_two_ranks_on_one_device(world):
    env = dict(os.environ, VROD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VROD_BENCH_VERIFY", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29560 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--rows", "300001"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3
    assert_multi_gpu_fields(out)
    assert out["exactness"]["certificate_fallback_queries"] == 0


@pytest.mark.gpu
def test_bench_plain_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in the command (the way the driver starts the
    1-GPU run): the parent starts the ranks itself, relays ONE JSON line and the ranks' exit code."""
    env = dict(os.environ, VROD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VROD_BENCH_VERIFY", None)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "300001"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert_multi_gpu_fields(out)
    pr = out["per_rank"]
    assert pr["backend"] == "gloo" and pr["rccl_ranks"] == 0     # rehearsal backend: not an RCCL measurement
    assert len(pr["scan_ms_per_step"]["all"]) == 2
    assert 0 < pr["scan_ms_per_step"]["min"] <= pr["scan_ms_per_step"]["max"]
    assert pr["allgather_merge_ms_per_step"]["max"] > 0 and pr["wall_ms_per_step"]["max"] > 0


def test_bench_world_size_mismatch_is_an_error():
    """WORLD_SIZE=4 with --gpus 8 must not silently report n_gpus=4 (checked before torch is imported)."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("rccl", ["1", "0"])
def test_bench_inprocess_two_shards_one_handle(rccl):
    """`bench.py --inprocess --gpus 2`: one process, one multi-device handle, the exchange inside the
    library (RCCL all-gather; VROD_RCCL=0: peer copies).  On the 1-GPU box both shards sit on device 0."""
    env = dict(os.environ, VROD_BENCH_DEVICES="0,0", VROD_RCCL=rccl, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VROD_BENCH_VERIFY", None)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--inprocess", "--gpus", "2", "--steps", "4", "--warmup", "1", "--rows", "400001"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert_multi_gpu_fields(out)
    assert ("RCCL all-gather" in out["config"]["parallelism"]) == (rccl == "1")
    # what each shard's device spent (vrod_index_shard_stats)
    assert [d["shard"] for d in out["per_device"]] == [0, 1] and all(d["device"] == 0 and d["scan_ms_per_step"] > 0 for d in out["per_device"])
