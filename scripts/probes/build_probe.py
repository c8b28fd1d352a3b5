import sys, time
sys.path.insert(0, ".")
import torch, vrod_amd as va
for dtype, n, dim in (("bf16", 10_000_000, 768), ("f32", 1_000_000, 768)):
    ix = va.Index(dim, dtype, "cosine")
    t = time.time(); ix.add_synthetic(1, 0, n); dt = time.time() - t
    print(f"build {dtype} {n}x{dim}: {dt:.3f} s ({n*dim*4/dt/1e9:.1f} GB/s of raw fp32)")
    ix.close()
