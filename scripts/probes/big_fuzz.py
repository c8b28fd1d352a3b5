"""Randomised searches on a corpus large enough for the staged batched scan with work stealing (1.2M rows), with duplicated
rows (failed certificates, band pass, the self-tuning margin), every result checked against the oracle (dev tool):
    python scripts/probes/big_fuzz.py [seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import vrod_amd as va
from oracle import oracle as O
O.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
bad = 0
for trial in range(2):
    dim = int(rng.choice([96, 192, 256]))
    dtype = ["bf16", "f32"][trial % 2]
    metric = ["cosine", "l2"][int(rng.integers(2))]
    DT, ME = (0 if dtype == "f32" else 1), (0 if metric == "cosine" else 1)
    n = 1_200_000
    raw = O.synth_rows(100 + seed, 0, n, dim, threads=16)
    ndup = 3000
    src = rng.integers(0, n, ndup)
    for c in range(1, 31):                       # 30 extra copies of 3000 rows, scattered
        raw[rng.integers(0, n, ndup)] = raw[src]
    prepared = O.prepare(raw, DT, ME, threads=16)
    with va.Index(dim, dtype, metric) as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        for step in range(10):
            k = int(rng.choice([1, 10, 100, 500]))
            nq = int(rng.choice([5, 64, 300, 1024]))
            hot = rng.random() < 0.5             # half of the batches: queries that ARE duplicated rows
            rq = raw[src[rng.integers(0, ndup, nq)]] if hot else O.synth_rows(200 + seed, step * 2048, nq, dim)
            pq = O.prepare(rq, DT, ME)
            if rng.random() < 0.5:
                ids, sc = ix.search(rq, k)
            else:
                dq = torch.from_numpy(rq).to(dev)
                oi_t = torch.empty((nq, k), dtype=torch.int64, device=dev); os_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
                ix.search_begin_device(dq, k, oi_t, os_t); ix.search_end()
                ids, sc = oi_t.cpu().numpy().view(np.uint64), os_t.cpu().numpy()
            st = ix.last_stats()
            oi, osc = O.scan_topk(prepared, pq, k, ME, threads=16)
            ok = np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
            bad += not ok
            print(f"trial {trial} {dtype}/{metric}/d{dim} step {step}: nq={nq} k={k} hot={hot} kprime={st['kprime']} launches={st['scan_launches']} "
                  f"fallback={st['fallback_queries']} band={st['band_queries']} split={st['split_pass']} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"seed {seed}: mismatches {bad}")
sys.exit(1 if bad else 0)
