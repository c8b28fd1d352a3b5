"""Randomised API-sequence fuzz against the oracle (dev tool): adds, synchronous and pipelined
searches of changing shapes / paths on ONE handle, every result checked."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import vrod_amd as va
from oracle import oracle as O
O.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
bad = 0
for trial in range(6):
    dim = int(rng.choice([32, 100, 192]))
    dtype = ["f32", "bf16"][int(rng.integers(2))]
    metric = ["cosine", "l2"][int(rng.integers(2))]
    mode = int(rng.integers(3))            # fp32 batches: fp32 MFMA pass / forced split pass / the library's default
    if mode == 2:
        os.environ.pop("VROD_F32_SPLIT", None)
    else:
        os.environ["VROD_F32_SPLIT"] = str(mode)
    DT = 0 if dtype == "f32" else 1
    ME = 0 if metric == "cosine" else 1
    rows = np.zeros((0, dim), np.float32)
    with va.Index(dim, dtype, metric) as ix:
        for step in range(25):
            op = rng.choice(["add", "sync", "pipe", "pipe", "sync"])
            if op == "add" or rows.shape[0] == 0:
                m = int(rng.choice([1, 300, 5000, 20000]))
                new = rng.standard_normal((m, dim)).astype(np.float32)
                ix.add(new); rows = np.concatenate([rows, new]); continue
            ix.set_path(int(rng.choice([0, 0, 1, 2])))
            k = int(rng.choice([1, 10, 50]))
            if op == "sync":
                nq = int(rng.choice([1, 3, 9, 40, 300]))
                rq = rng.standard_normal((nq, dim)).astype(np.float32)
                ids, sc = ix.search(rq, k)
                oi, osc = O.search(rows, rq, k, DT, ME, threads=8)
                ok = np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
            else:
                nb = int(rng.choice([2, 3, 5]))
                same = bool(rng.integers(2))
                nqs = [int(rng.choice([1, 2, 8, 40]))] * nb if same else [int(rng.choice([1, 2, 8, 40, 260])) for _ in range(nb)]
                hq = [rng.standard_normal((n, dim)).astype(np.float32) for n in nqs]
                dq = [torch.from_numpy(h).to(dev) for h in hq]
                outs = [(torch.empty((n, k), dtype=torch.int64, device=dev), torch.empty((n, k), dtype=torch.float32, device=dev)) for n in nqs]
                res = []
                ix.search_begin_device(dq[0], k, *outs[0])
                for s in range(nb):
                    if s + 1 < nb: ix.search_begin_device(dq[s + 1], k, *outs[s + 1])
                    ix.search_end()
                    res.append((outs[s][0].cpu().numpy().view(np.uint64), outs[s][1].cpu().numpy()))
                ok = True
                for s in range(nb):
                    oi, osc = O.search(rows, hq[s], k, DT, ME, threads=8)
                    ok &= np.array_equal(res[s][0], oi) and np.array_equal(res[s][1].view(np.uint32), osc.view(np.uint32))
            if not ok:
                bad += 1
                print(f"MISMATCH seed {seed} trial {trial} step {step} op {op} dim {dim} {dtype} {metric} rows {rows.shape[0]} k {k}", flush=True)
print(f"seed {seed}: mismatches {bad}")
