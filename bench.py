#!/usr/bin/env python
"""bench.py -- the driver's measurement contract for the vRod similarity-scan hot path.

    python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 without a launcher this
        process starts the N ranks itself -- before importing torch or touching a GPU -- relays
        rank 0's JSON line and exits with the ranks' return code)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --inprocess                   (ONE process, one multi-device handle:
        vrod_index_create(n_devices = N), the exchange is the library's own RCCL all-gather)

One "step" = one batch of queries through the whole search (fast scan -> candidates ->
canonical re-score -> certified top-k), corpus resident in HBM before the timed region.
Two batches are in flight (begin(s+1) is enqueued before end(s)); every one of the K batches
is begun, completed, exchanged and merged inside the timed region.
Default workload = the configuration BASELINE.json's metric is quoted on:
10M x 768 bf16 cosine, batch 1024, top-10 ("cfg3").  With N GPUs the 10M rows are sharded
into contiguous ranges (strong scaling), every rank scans its shard for the same batch and
the per-shard top-k are all-gathered over RCCL and merged.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events recorded by the
library on its own stream around every launch of the dominant scan kernel; `cpu_baseline`
is the CPU oracle (a build-authored restatement: vRod itself has no scan) timed on this
host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: rows, dim, dtype, metric, batch, k, roofline bound
    "cfg3": dict(n=10_000_000, dim=768, dtype="bf16", metric="cosine", nq=1024, k=10, bound="mfma"),
    "cfg2": dict(n=1_000_000, dim=768, dtype="f32", metric="l2", nq=1, k=100, bound="hbm"),
    "cfg4": dict(n=40_000_000, dim=768, dtype="bf16", metric="cosine", nq=1024, k=10, bound="mfma"),
    "cfg5": dict(n=10_000_000, dim=1536, dtype="f32", metric="cosine", nq=256, k=1000, bound="mfma"),
    "tiny": dict(n=200_000, dim=768, dtype="bf16", metric="cosine", nq=1024, k=10, bound="mfma"),
    # cfg3 with exact duplicates (real embedding corpora hold them): 5 % of the rows are 64 copies each of 7812 rows.
    # cfg3dup: random queries (few land on a duplicated row); cfg3hot: EVERY query is a duplicated row -- all 65 copies tie
    # at the top, no certificate can hold, the whole batch takes the second chance (band pass)
    "cfg3dup": dict(n=10_000_000, dim=768, dtype="bf16", metric="cosine", nq=1024, k=10, bound="mfma", dup_groups=7812, dup_copies=64),
    "cfg3hot": dict(n=10_000_000, dim=768, dtype="bf16", metric="cosine", nq=1024, k=10, bound="mfma", dup_groups=7812, dup_copies=64, hot=True),
}
PEAK = {  # /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
    "hbm": (8000.0, "GB/s"),            # HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
    "mfma_bf16": (2500.0, "TFLOP/s"),   # dense bf16 MFMA
    "mfma_f32": (157.3, "TFLOP/s"),     # fp32-input MFMA
}
CORPUS_SEED, QUERY_SEED = 1, 2
DT = {"f32": 0, "bf16": 1}
ME = {"cosine": 0, "l2": 1}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override total corpus rows (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true", help="skip the extra 1-query HBM-roofline probe (cfg2)")
    ap.add_argument("--no-host-probe", action="store_true", help="skip the host-pointer (vrod_search, PCIe-inclusive) probe")
    ap.add_argument("--inprocess", action="store_true",
                    help="one process, one handle over --gpus devices (vrod_index_create n_devices > 1, RCCL inside the library)")
    return ap.parse_args()


def self_launch(args) -> None:
    """`python bench.py --gpus N` (N > 1) with no launcher: start the N ranks as CHILD processes
    (`python -m torch.distributed.run ...`), relay rank 0's one JSON line, exit with their code.
    This parent never imports torch and never touches a GPU (and never re-execs itself)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        sys.stderr.write(f"bench: {args.gpus} ranks ended with code {p.returncode} and {len(lines)} JSON line(s)\n")
        sys.stderr.write(p.stdout.decode("utf-8", "replace")[-4000:])
        raise SystemExit(p.returncode or 1)
    print(lines[0], flush=True)
    raise SystemExit(0)


def run_steps(ix, wl, steps, first_step, world, rank, dist, va, torch, dev, collective=None):
    """Run `steps` batches; returns accumulated library stats of this rank.

    Two batches are in flight (vrod_search_begin_* / vrod_search_end): batch s+1 is enqueued on
    the library's stream before batch s is completed, so the device goes from one scan straight
    into the next while the host reads batch s's certificate verdicts and this rank's top-k of
    batch s is all-gathered (RCCL) and merged on torch's stream.  Every batch that is begun is
    ended, exchanged and merged inside this function: the pipeline is drained before it returns.
    VROD_BENCH_PIPELINE=0 runs the batches strictly one after the other instead.
    """
    from vrod_amd.shard import all_gather_packed, alloc_packed
    nq, k = wl["nq"], wl["k"]
    depth = 1 if os.environ.get("VROD_BENCH_PIPELINE") == "0" else 2
    # this rank's results live in one packed block (ids | scores): the exchange is ONE all-gather
    bufs = [alloc_packed(nq, k, dev) for _ in range(depth)]
    collective = world > 1 if collective is None else collective
    if collective:
        gathered = torch.empty(world * bufs[0][0].numel(), dtype=torch.uint8, device=dev)
        mi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        ms = torch.empty((nq, k), dtype=torch.float32, device=dev)
    acc = dict(scan_ms=0.0, scan_flops=0.0, scan_bytes=0.0, launches=0, fallback=0, band=0, max_err=0.0, eps=0.0, split=0,
               exchange_ms=0.0, exchange_host_ms=0.0, sample_ms=0.0, sample_launches=0, overlap_ms=0.0)
    ex_events = []   # (start, stop) on torch's stream around all-gather + merge of every batch

    def begin(s):
        _, oi, osc = bufs[s % depth]
        if wl.get("hot"):   # the batch's queries ARE duplicated corpus rows (rows [r, r + nq) of the corpus stream)
            ix.search_begin_synthetic_device(CORPUS_SEED, ((first_step + s) * nq) % (wl["dup_groups"] - nq), nq, k, oi, osc)
        else:
            ix.search_begin_synthetic_device(QUERY_SEED, (first_step + s) * nq, nq, k, oi, osc)

    if steps > 0:
        begin(0)
    for s in range(steps):
        if depth == 2 and s + 1 < steps:
            begin(s + 1)
        ix.search_end()
        st = ix.last_stats()
        packed, oi, osc = bufs[s % depth]
        if collective:
            # per-shard top-k -> every rank (RCCL all-gather over xGMI), then the exact merge
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            e0.record()
            all_gather_packed(dist, packed, gathered)
            va.merge_topk_packed_device(dev.index, wl["metric"], gathered, world, nq, k, mi, ms)
            e1.record()
            acc["exchange_host_ms"] += (time.perf_counter() - h0) * 1e3
            ex_events.append((e0, e1))
        if depth == 1 and s + 1 < steps:
            begin(s + 1)
        acc["scan_ms"] += st["scan_ms"]
        acc["sample_ms"] += st["sample_ms"]                      # of scan_ms: the sample-pass launch (its own kernel form)
        acc["sample_launches"] += 1 if st["sample_ms"] > 0 else 0
        acc["overlap_ms"] += st["overlap_ms"]                    # of scan_ms: two scan launches in flight at once (counted twice in the sum)
        acc["scan_flops"] += st["scan_flops"]
        acc["scan_bytes"] += st["scan_bytes"]
        acc["launches"] += st["scan_launches"]
        acc["fallback"] += st["fallback_queries"]
        acc["band"] += st["band_queries"]
        acc["split"] += st["split_pass"]
        acc["max_err"] = max(acc["max_err"], st["max_fast_err"])
        acc["eps"] = st["eps_bound"]
    if ex_events:
        torch.cuda.synchronize(dev)
        acc["exchange_ms"] = sum(a.elapsed_time(b) for a, b in ex_events)
    final = (mi, ms) if collective else (bufs[(steps - 1) % depth][1], bufs[(steps - 1) % depth][2])
    return acc, final


def baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim (the config this bench measures by default)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "queries/sec + recall@10 vs CPU ref, 10M\u00d7768 cosine, batch=1024, 1/2/4/8 GPU"


def host_cores() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(wl, va, torch, dev, n_total):
    """Oracle on this host's cores on a bounded sample; also recall@k of the HIP path on it."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    cores = host_cores()
    dim, k = wl["dim"], wl["k"]
    ns = int(min(n_total, max(65536, (6 << 30) // (dim * 4))))          # <= 6 GB of fp32 rows
    ns = min(ns, 2_000_000)
    t = time.time()
    raw = O.synth_rows(CORPUS_SEED, 0, ns, dim, threads=cores)
    corpus = O.prepare(raw, DT[wl["dtype"]], ME[wl["metric"]], threads=cores)
    # bf16 workloads: also the oracle over the ORIGINAL fp32 data ("additionally
    # reported against the fp32-data oracle, labelled separately"), on a smaller query sample
    corpus32 = O.prepare(raw, DT["f32"], ME[wl["metric"]], threads=cores) if wl["dtype"] == "bf16" else None
    del raw
    # size the query samples for ~12 s (all cores) and ~10 s (one thread) at ~2.5e9 mul-add/s/thread
    q_all = int(max(2, min(512, 12.0 * 2.5e9 * cores / (ns * dim))))
    q_one = int(max(1, min(32, 10.0 * 2.5e9 / (ns * dim))))
    rq = O.synth_rows(QUERY_SEED, 0, q_all, dim)
    pq = O.prepare(rq, DT[wl["dtype"]], ME[wl["metric"]])
    gen_s = time.time() - t
    t = time.time()
    oi, osc = O.scan_topk(corpus, pq, k, ME[wl["metric"]], threads=cores)
    t_all = time.time() - t
    t = time.time()
    O.scan_topk(corpus, pq[:q_one], k, ME[wl["metric"]], threads=1)
    t_one = time.time() - t
    scale = ns / float(n_total)  # scan time is linear in rows
    out = {
        "value": round(q_all / t_all * scale, 4), "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": f"{q_all} queries x first {ns} of {n_total} rows, {cores} threads, {t_all:.2f}s; "
                  f"rate scaled by rows ({ns}/{n_total}); build-authored CPU restatement (vRod has no scan)",
        "value_1thread": round(q_one / t_one * scale, 4),
        "sample_1thread": f"{q_one} queries x {ns} rows, 1 thread (vRod is single-threaded: Rc<RefCell>), {t_one:.2f}s",
        "host_prep_s": round(gen_s, 2),
    }
    # recall@k of the HIP path against the oracle on the same sample corpus / queries
    with va.Index(dim, wl["dtype"], wl["metric"], device=dev.index) as sx:
        sx.add_synthetic(CORPUS_SEED, 0, ns)
        if wl["nq"] > 8:
            sx.set_path(va.PATH_MFMA)
        ids, sc = sx.search(rq, k)
    hits = sum(len(set(ids[i].tolist()) & set(oi[i].tolist())) for i in range(q_all))
    recall = hits / float(q_all * min(k, ns))
    bit_exact = bool(np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32)))
    extra = {}
    if corpus32 is not None:
        q32 = min(q_all, 64)
        pq32 = O.prepare(rq[:q32], DT["f32"], ME[wl["metric"]])
        fi, fs = O.scan_topk(corpus32, pq32, k, ME[wl["metric"]], threads=cores)
        hits32 = sum(len(set(ids[i].tolist()) & set(fi[i].tolist())) for i in range(q32))
        rel = np.abs(sc[:q32].astype(np.float64) - fs.astype(np.float64)) / np.maximum(np.abs(fs.astype(np.float64)), 1e-30)
        same = ids[:q32] == fi
        extra[f"recall_at_{k}_vs_fp32_oracle"] = round(hits32 / float(q32 * min(k, ns)), 6)
        extra["fp32_oracle_note"] = (f"{q32} queries, first {ns} rows: the HIP bf16 result against the oracle over the UN-rounded fp32 data "
                                     "(bf16 storage moves neighbours near the k-th boundary; the bf16-data oracle above is the parity target); "
                                     f"max relative score difference where the ids agree: {float(rel[same].max()) if same.any() else 0.0:.3e}")
    return out, recall, bit_exact, f"{q_all} queries vs oracle on the first {ns} rows", extra


def multi_gpu_checks(out, wl, va, torch, dev_index, n_total, final, last_batch):
    """Every N > 1 line carries what BASELINE.json's metric asks beside the QPS (recall@k vs the CPU reference, exactness),
    computed OUTSIDE the timed region on one device: (1) a single-device handle over the whole corpus searches the last
    timed batch -- the merged result must equal it bit for bit (`verify_merged_equals_single_device`; a false verdict
    makes the process exit non-zero) and recall@k of the merged ids against it is reported; (2) that single-device handle
    is checked against the oracle on a bounded sample (first rows, a few queries: seconds of host time).  The full
    `cpu_baseline` timing stays an N = 1 item."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    nq, k, dim = wl["nq"], wl["k"], wl["dim"]
    dev = torch.device("cuda", dev_index)
    need = n_total * dim * (2 if wl["dtype"] == "bf16" else 4) * (2.2 if wl["dtype"] == "f32" else 1.1)
    free, _total = torch.cuda.mem_get_info(dev)
    if need > free:
        out["verify_merged_equals_single_device"] = None
        out["verify_note"] = f"a single-device handle over all {n_total} rows needs {need / 2**30:.0f} GiB, {free / 2**30:.0f} GiB free on device {dev_index}: not run"
    else:
        with va.Index(dim, wl["dtype"], wl["metric"], device=dev_index) as fx:
            fx.add_synthetic(CORPUS_SEED, 0, n_total)
            vi = torch.empty((nq, k), dtype=torch.int64, device=dev)
            vs = torch.empty((nq, k), dtype=torch.float32, device=dev)
            fx.search_synthetic_device(QUERY_SEED, last_batch * nq, nq, k, vi, vs)
            fi, fs = final
            same = bool(torch.equal(vi, fi) and torch.equal(vs.view(torch.int32), fs.view(torch.int32)))
            out["verify_merged_equals_single_device"] = same
            a, b = vi.cpu().numpy(), fi.cpu().numpy()
            out[f"recall_at_{k}"] = round(sum(len(set(a[i].tolist()) & set(b[i].tolist())) for i in range(nq)) / float(nq * min(k, n_total)), 6)
            out["recall_sample"] = f"the merged result of the last timed batch ({nq} queries) against a single-device search of all {n_total} rows"
    # the single-device path against the oracle on a bounded sample (the N = 1 line's check, smaller)
    cores = host_cores()
    ns = int(min(n_total, 1_000_000))
    q = int(max(2, min(32, 4.0 * 2.5e9 * cores / (ns * dim))))
    raw = O.synth_rows(CORPUS_SEED, 0, ns, dim, threads=cores)
    rq = O.synth_rows(QUERY_SEED, 0, q, dim)
    oi, osc = O.search(raw, rq, k, DT[wl["dtype"]], ME[wl["metric"]], threads=cores)
    del raw
    with va.Index(dim, wl["dtype"], wl["metric"], device=dev_index) as sx:
        sx.add_synthetic(CORPUS_SEED, 0, ns)
        if nq > 8:
            sx.set_path(va.PATH_MFMA)
        ids, sc = sx.search(rq, k)
    out["bit_exact_vs_oracle_on_sample"] = bool(np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32)))
    out["oracle_sample"] = f"{q} queries x first {ns} rows, single-device handle vs the CPU oracle ({cores} threads)"
    out["parity"] = "unpinned by the reference (vRod holds no scan, tests or vectors): the oracle is a build-authored restatement"
    return out.get("verify_merged_equals_single_device") is not False and out["bit_exact_vs_oracle_on_sample"]


def main_inprocess(args):
    """ONE process, ONE handle over N devices (vrod_index_create with n_devices = N): the deployment a
    single-threaded host like vRod uses.  Rows are dealt to the devices in blocks of 65536; each batch
    is scanned on every device, the per-shard top-k travel through the library's own RCCL all-gather
    (ncclCommInitAll communicator, one ncclAllGather per device in a group call) and are merged on the
    first device.  Two batches in flight, as in the multi-process form.
    VROD_BENCH_DEVICES=0,0: the device list (rehearsal on fewer GPUs than shards)."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import vrod_amd as va
    va.load()
    devs = [int(x) for x in os.environ.get("VROD_BENCH_DEVICES", ",".join(str(i) for i in range(args.gpus))).split(",")]
    if len(devs) != args.gpus:
        raise SystemExit(f"bench: VROD_BENCH_DEVICES names {len(devs)} devices but --gpus {args.gpus}")
    wl = dict(WORKLOADS[args.workload])
    n_total = args.rows or wl["n"]
    nq, k = wl["nq"], wl["k"]
    dev = torch.device("cuda", devs[0])
    torch.cuda.set_device(dev)
    ix = va.Index(wl["dim"], wl["dtype"], wl["metric"], devices=devs) if len(devs) > 1 else va.Index(wl["dim"], wl["dtype"], wl["metric"], device=devs[0])
    ix.add_synthetic(CORPUS_SEED, 0, n_total)
    ix.set_profiling(True)
    outs = [(torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev)) for _ in range(2)]

    def run(steps, first):
        acc = dict(scan_ms=0.0, scan_flops=0.0, launches=0, fallback=0, exchange=0, split=0, dev_ms=[0.0] * len(devs))
        if steps:
            ix.search_begin_synthetic_device(QUERY_SEED, first * nq, nq, k, *outs[0])
        for s in range(steps):
            if s + 1 < steps:
                ix.search_begin_synthetic_device(QUERY_SEED, (first + s + 1) * nq, nq, k, *outs[(s + 1) % 2])
            ix.search_end()
            st = ix.last_stats()
            acc["scan_ms"] += st["scan_ms"]          # slowest device of the batch
            acc["scan_flops"] += st["scan_flops"]    # all devices
            acc["launches"] += st["scan_launches"]
            acc["fallback"] += st["fallback_queries"]
            acc["exchange"] = st["exchange"]
            acc["split"] += st["split_pass"]
            if len(devs) > 1:
                for g in range(len(devs)):
                    acc["dev_ms"][g] += ix.shard_stats(g)["scan_ms"]
        return acc

    def sync_all():
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    run(args.warmup, 0)
    sync_all()
    t0 = time.perf_counter()
    acc = run(args.steps, args.warmup)
    sync_all()
    elapsed = time.perf_counter() - t0
    G = len(devs)
    factor = 3.0 if acc["split"] else 1.0
    per_gpu = factor * acc["scan_flops"] / G / (acc["scan_ms"] * 1e-3) / 1e12 if acc["scan_ms"] else 0.0
    peak, unit = PEAK["mfma_bf16" if wl["dtype"] == "bf16" or acc["split"] else "mfma_f32"]
    if wl["bound"] == "hbm":
        peak, unit = PEAK["hbm"]
        per_gpu = 0.0
    out = {
        "metric": baseline_metric() if args.workload == "cfg3" and not args.rows else f"queries/sec, {args.workload}",
        "value": round(nq * args.steps / elapsed, 2), "unit": "queries/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": wl["dtype"], "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_total} x {wl['dim']} {wl['dtype']} {wl['metric']}, batch={nq}, top-{k}", "rows_total": n_total,
                   "dim": wl["dim"], "batch": nq, "k": k, "metric": wl["metric"], "devices": devs,
                   "parallelism": f"ONE process, one handle over {G} devices (rows dealt in blocks of 65536); exchange = "
                                  + {0: "none", 1: "RCCL all-gather inside the library (ncclCommInitAll)", 2: "peer copies (VROD_RCCL=0)"}[acc["exchange"]],
                   "corpus_seed": CORPUS_SEED, "query_seed": QUERY_SEED, "batches_in_flight": 2},
        "roofline": {"bound": "hbm" if wl["bound"] == "hbm" else "mfma", "achieved": round(per_gpu, 2), "peak": peak, "unit": unit,
                     "frac": round(per_gpu / peak, 4), "traffic": None, "traffic_source": "not measured in this run",
                     "kernel": "scan_mfma_w4_kernel", "note": "per GPU: all devices' algorithmic flops / N / the slowest device's scan time per batch",
                     "launches_per_step": acc["launches"] / max(args.steps, 1)},
        "library": va.version(),
        "exactness": {"certificate_fallback_queries": acc["fallback"]},
    }
    if len(devs) > 1:   # what each shard's device spent scanning, per batch (vrod_index_shard_stats)
        out["per_device"] = [{"shard": g, "device": devs[g], "scan_ms_per_step": round(acc["dev_ms"][g] / max(args.steps, 1), 4)} for g in range(len(devs))]
    final = tuple(t.clone() for t in outs[(args.steps - 1) % 2])
    ix.close()
    checks_ok = True
    if len(devs) > 1 and os.environ.get("VROD_BENCH_VERIFY") != "0":
        checks_ok = multi_gpu_checks(out, wl, va, torch, devs[0], n_total, final, args.warmup + args.steps - 1)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if not checks_ok:
        sys.stderr.write("bench: the merged multi-device result differs from the single-device search / the oracle\n")
        sys.exit(3)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.inprocess:
        self_launch(args)   # never returns; this process has not imported torch nor touched a GPU
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench: WORLD_SIZE={os.environ['WORLD_SIZE']} but --gpus {args.gpus}: launch exactly one rank per GPU")
    if args.inprocess:
        return main_inprocess(args)
    # stdout carries exactly ONE JSON line.  Native libraries write there too (RCCL prints a
    # version banner at communicator creation), so fd 1 points at stderr until the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # VROD_BENCH_BACKEND=gloo: rehearsal of the N>1 control flow on a box with fewer GPUs than
    # ranks (ranks share devices, the exchange is staged through the host) -- never a measurement
    backend = os.environ.get("VROD_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # VROD_BENCH_FORCE_COLLECTIVE=1: run the all-gather + merge even with one rank (rehearsal of
    # the N>1 code path on a 1-GPU box; the exchange is then part of the timed step)
    force_coll = os.environ.get("VROD_BENCH_FORCE_COLLECTIVE") == "1"
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import vrod_amd as va
    va.load()  # raises without the HIP library: there is no fallback path
    from vrod_amd.shard import shard_range

    wl = dict(WORKLOADS[args.workload])
    n_total = args.rows or wl["n"]
    lo, hi = shard_range(n_total, rank, world)
    ix = va.Index(wl["dim"], wl["dtype"], wl["metric"], device=dev_index)
    if wl.get("dup_groups"):
        if world != 1:
            raise SystemExit("bench: the duplicate workloads are single-GPU")
        g, c = wl["dup_groups"], wl["dup_copies"]
        ix.reserve(n_total)
        ix.add_synthetic(CORPUS_SEED, 0, n_total - g * c)
        for _ in range(c):
            ix.add_synthetic(CORPUS_SEED, 0, g)      # one more copy of rows [0, g)
    else:
        ix.add_synthetic(CORPUS_SEED, lo, hi - lo)   # shard rows [lo, hi) of the synthetic stream, generated in HBM
    ix.set_id_offset(lo)
    ix.set_profiling(True)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1 or force_coll:
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
            torch.cuda.synchronize(dev)

    coll = world > 1 or force_coll
    exit_code = 0
    run_steps(ix, wl, args.warmup, 0, world, rank, dist, va, torch, dev, coll)
    fence()
    t0 = time.perf_counter()
    acc, final = run_steps(ix, wl, args.steps, args.warmup, world, rank, dist, va, torch, dev, coll)
    fence()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    if world > 1:
        rdev = dev if backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # per-rank kernel stats: report the slowest rank's scan time, the sum of flops
        ks = torch.tensor([acc["scan_ms"], acc["scan_flops"], acc["scan_bytes"], float(acc["fallback"])], dtype=torch.float64, device=rdev)
        kmax = ks.clone(); dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        ksum = ks.clone(); dist.all_reduce(ksum, op=dist.ReduceOp.SUM)
        fallback_total = int(ksum[3].item())
        # where a scaling loss sits: every rank's own scan time, exchange time (all-gather + merge on
        # torch's stream, next batch's scan running beside it) and wall time of the timed region
        mine = torch.tensor([acc["scan_ms"], acc["exchange_ms"], acc["exchange_host_ms"], my_elapsed * 1e3],
                            dtype=torch.float64, device=rdev)
        allr = torch.empty(world * 4, dtype=torch.float64, device=rdev)
        dist.all_gather_into_tensor(allr, mine)
        per_rank_stats = allr.view(world, 4).cpu().tolist()
    else:
        fallback_total = acc["fallback"]
        per_rank_stats = [[acc["scan_ms"], acc["exchange_ms"], acc["exchange_host_ms"], elapsed * 1e3]]

    if rank == 0:
        nq = wl["nq"]
        value = nq * args.steps / elapsed
        # roofline of the dominant kernel on THIS rank (rank 0): algorithmic work / HIP-event time
        if wl["bound"] == "hbm":
            achieved = acc["scan_bytes"] / (acc["scan_ms"] * 1e-3) / 1e9 if acc["scan_ms"] else 0.0
            peak, unit = PEAK["hbm"]
            kernel = "scan_stream_kernel"
            per_launch = acc["scan_bytes"] / max(acc["launches"], 1)
            work_key = "algorithmic_bytes_per_launch"
        else:
            # split pass (the default for fp32 batches while memory allows, VROD_F32_SPLIT=0 switches it off):
            # an fp32 corpus scanned as three bf16 products (hi.hi + hi.lo + lo.hi) on the bf16 matrix cores:
            # the kernel executes 3x the algorithmic flops, priced against the bf16 peak
            split = acc["split"] > 0
            if split and acc["split"] != args.steps:
                raise SystemExit("bench: the fast pass changed between timed batches (split pass switched off mid-run)")
            factor = 3.0 if split else 1.0
            # time during which at least one scan launch was in flight: the sum of the launches' own times minus what
            # two of them spent side by side (0 unless the pipelined order overlaps a sample pass with a last stage)
            busy_ms = acc["scan_ms"] - acc["overlap_ms"]
            achieved = factor * acc["scan_flops"] / (busy_ms * 1e-3) / 1e12 if busy_ms > 0 else 0.0
            peak, unit = PEAK["mfma_bf16" if wl["dtype"] == "bf16" or split else "mfma_f32"]
            # bf16 rows and the bf16 planes of fp32 rows: the 4-wave kernel; fp32 rows without planes: the 8-wave phased one
            kernel = "scan_mfma_w4_kernel" if split or wl["dtype"] == "bf16" else "scan_mfma_phased_kernel"
            per_launch = factor * acc["scan_flops"] / max(acc["launches"], 1)
            work_key = "algorithmic_flops_per_launch"
            # the same split the way rocprofv3 --stats shows it: the filtered launches (the kernel that covers the
            # algorithmic work) and the sample-pass launch (another instantiation; covers none) are two kernel rows
            nf = acc["launches"] - acc["sample_launches"]
            tf = acc["scan_ms"] - acc["sample_ms"]
            by_kernel = {
                "filtered_launches": {"launches_per_step": nf / max(args.steps, 1), "avg_launch_ms": round(tf / max(nf, 1), 4),
                                      "frac_without_the_sample_launch": round(factor * acc["scan_flops"] / (tf * 1e-3) / 1e12 / peak, 4) if tf > 0 else None},
                "sample_launch": {"launches_per_step": acc["sample_launches"] / max(args.steps, 1),
                                  "avg_launch_ms": round(acc["sample_ms"] / max(acc["sample_launches"], 1), 4)},
            }
        # HBM bytes per launch come from a separate rocprofv3 --pmc FETCH_SIZE pass of this same command
        # (counters cannot be read from inside the run): the committed summary, never a live value
        traffic, traffic_source = None, "not measured in this run (rocprofv3 --pmc FETCH_SIZE is a separate pass)"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1 and not args.rows:
            try:
                tj = json.load(open(tpath)).get(args.workload, {})
                traffic = tj.get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = f"profiles/traffic.json ({tj.get('source', 'rocprofv3 --pmc FETCH_SIZE x2, own pass')}; NOT this run)"
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm" if wl["bound"] == "hbm" else "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel,
            "launches_per_step": acc["launches"] / max(args.steps, 1),
            "avg_launch_ms": round(acc["scan_ms"] / max(acc["launches"], 1), 4), work_key: per_launch,
            "timing": "HIP events attached to each scan dispatch on the library's stream (hipExtLaunchKernelGGL start/stop), timed steps only",
        }
        if wl["bound"] == "mfma":
            roofline["by_kernel"] = by_kernel
            if kernel == "scan_mfma_w4_kernel":   # context, NOT a value of this run
                roofline["context"] = ("peak = 2.5 PF nominal (2.4 GHz); under MFMA load with real data the chip holds 1.85-2.2 GHz (power): the scan "
                                       "loop's instruction mix alone measures 0.91 of peak with MFMAs only, 0.76 with the 256x256 tile's LDS fills "
                                       "from L2 and ~0.68 at a search's HBM share (scripts/ubench/loop_mix.hip, profiles/r03/x_loop_mix.log)")
        if wl["bound"] == "mfma" and acc["overlap_ms"] > 0:
            # (vrod_index.hip: up to 6M rows per handle the next batch's sample pass runs beside this batch's last stage)
            roofline["overlap"] = {
                "overlap_ms_per_step": round(acc["overlap_ms"] / max(args.steps, 1), 4),
                "scan_busy_ms_per_step": round((acc["scan_ms"] - acc["overlap_ms"]) / max(args.steps, 1), 4),
                "sum_of_launch_ms_per_step": round(acc["scan_ms"] / max(args.steps, 1), 4),
                "note": "the sample pass of the next batch runs beside this batch's last stage: each launch's own time (avg_launch_ms, "
                        "by_kernel, rocprofv3) includes waiting for compute units and the sum counts the shared time twice; achieved / "
                        "frac divide by the time at least one scan launch was in flight (vrod_search_stats: scan_ms - overlap_ms)"}
        out = {
            "metric": baseline_metric() if args.workload == "cfg3" and not args.rows else f"queries/sec, {args.workload}",
            "value": round(value, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n_total} x {wl['dim']} {wl['dtype']} {wl['metric']}, batch={nq}, top-{wl['k']}",
                       "rows_total": n_total, "rows_per_gpu": hi - lo, "dim": wl["dim"], "batch": nq, "k": wl["k"],
                       "metric": wl["metric"], "parallelism": f"row-shard x{world}, RCCL all-gather of per-shard top-k" if world > 1 else "single GPU",
                       "corpus_seed": CORPUS_SEED, "query_seed": QUERY_SEED,
                       "batches_in_flight": 1 if os.environ.get("VROD_BENCH_PIPELINE") == "0" else 2},
            "roofline": roofline,
            "library": va.version(),   # names the hipcc that built the device code (the register audit ran on ITS output)
            "exactness": {"certificate_fallback_queries": fallback_total, "resolved_by_band_pass": acc["band"],
                          "max_fast_err": acc["max_err"], "eps_bound": acc["eps"],
                          "note": "ids and score bits equal the CPU oracle by construction (canonical re-score + certificate)"},
        }
        if coll:
            steps_n = max(args.steps, 1)
            col = lambda j: [r[j] / steps_n for r in per_rank_stats]
            out["per_rank"] = {
                "rccl_ranks": world if backend == "nccl" else 0, "backend": backend,
                "scan_ms_per_step": {"min": round(min(col(0)), 4), "max": round(max(col(0)), 4), "all": [round(x, 4) for x in col(0)]},
                "allgather_merge_ms_per_step": {"min": round(min(col(1)), 4), "max": round(max(col(1)), 4),
                                                "note": "HIP events on the exchange stream around all-gather + merge; the next batch's scan runs beside it"},
                "allgather_merge_host_ms_per_step": {"min": round(min(col(2)), 4), "max": round(max(col(2)), 4)},
                "wall_ms_per_step": {"min": round(min(col(3)), 4), "max": round(max(col(3)), 4)},
                "roofline_rank": 0,
            }
        checks_ok = True
        if coll and not wl.get("dup_groups") and os.environ.get("VROD_BENCH_VERIFY") != "0":
            # N > 1: recall / exactness fields of the metric, outside the timed region (VROD_BENCH_VERIFY=0 skips them)
            ix.close()
            checks_ok = multi_gpu_checks(out, wl, va, torch, dev_index, n_total, final, args.warmup + args.steps - 1)
        if world == 1 and not args.no_cpu_baseline and not wl.get("dup_groups"):
            cb, recall, bit_exact, rs, extra = cpu_baseline(wl, va, torch, dev, n_total)
            out["cpu_baseline"] = cb
            out[f"recall_at_{wl['k']}"] = round(recall, 6)
            out["recall_sample"] = rs
            out["bit_exact_vs_oracle_on_sample"] = bit_exact
            out["parity"] = "unpinned by the reference (vRod holds no scan, tests or vectors): the oracle is a build-authored restatement"
            out.update(extra)
        if world == 1 and not args.no_host_probe and not wl.get("dup_groups"):
            # the boundary as SURVEY.md 8(b) specifies it: vrod_search, host pointers in, results in host
            # memory, synchronous (PCIe both ways inside the call).  Reported beside `value`, never as it.
            hq = va.synth_rows_device(dev_index, QUERY_SEED, 0, nq, wl["dim"]).cpu().numpy()
            ix.set_profiling(False)
            ix.search(hq, wl["k"])
            hsteps = max(3, min(10, args.steps))
            th = time.perf_counter()
            for _ in range(hsteps):
                ix.search(hq, wl["k"])
            th = time.perf_counter() - th
            out["host_pointer_qps"] = {"value": round(nq * hsteps / th, 2), "unit": "queries/s", "ms_per_batch": round(th / hsteps * 1e3, 4),
                                       "batches": hsteps, "entry_point": "vrod_search (host fp32 queries in, host ids + scores out, synchronous; "
                                       "one batch at a time, nothing in flight across calls)"}
        if world == 1 and not args.no_hbm_probe and args.workload == "cfg3":
            # the north_star's second roofline: the memory-bound 1-query scan (configs[1])
            ix.close()
            w2 = WORKLOADS["cfg2"]
            hx = va.Index(w2["dim"], w2["dtype"], w2["metric"], device=dev_index)
            hx.add_synthetic(CORPUS_SEED, 0, w2["n"])
            hx.set_profiling(True)
            run_steps(hx, w2, 3, 0, 1, 0, dist, va, torch, dev)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            a2, _ = run_steps(hx, w2, 20, 3, 1, 0, dist, va, torch, dev)
            torch.cuda.synchronize(dev)
            e2 = time.perf_counter() - t1
            gbps = a2["scan_bytes"] / (a2["scan_ms"] * 1e-3) / 1e9 if a2["scan_ms"] else 0.0
            out["extra_hbm_1query"] = {
                "workload": "cfg2: 1000000 x 768 f32 l2, batch=1, top-100", "queries_per_s": round(20 / e2, 2),
                "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK["hbm"][0], "unit": "GB/s",
                             "frac": round(gbps / PEAK["hbm"][0], 4), "kernel": "scan_stream_kernel",
                             "avg_launch_ms": round(a2["scan_ms"] / max(a2["launches"], 1), 4)}}
            hx.close()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
        if not checks_ok:
            sys.stderr.write("bench: the merged multi-GPU result differs from the single-device search / the oracle\n")
            exit_code = 3
    if world > 1 or force_coll:
        dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
