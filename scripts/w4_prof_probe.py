"""Where the filtered 4-wave kernel's cycles go (diagnostic build: scripts/build_variant.sh prof -DVROD_W4_PROF).

    VROD_HIP_LIB=$PWD/vrod_amd/libvrod_prof.so python scripts/w4_prof_probe.py [rows] [nq] [batches]
Counters are shader-clock totals over every wave of every filtered launch (see g_w4_prof in kernels_mfma.hip).
"""
import ctypes
import sys
import torch
sys.path.insert(0, ".")
import vrod_amd as va
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
k = 10
lib = va.load()
ix = va.Index(768, "bf16", "cosine")
ix.add_synthetic(1, 0, n)
ix.set_path(va.PATH_MFMA)
oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
buf = (ctypes.c_ulonglong * 16)()
ix.search_synthetic_device(2, 0, nq, k, oi, osc)     # warm
lib.vrod_debug_w4_prof(buf, 1)
for s in range(steps):
    ix.search_synthetic_device(2, (s + 1) * nq, nq, k, oi, osc)
assert lib.vrod_debug_w4_prof(buf, 1) == 0
c = [int(x) for x in buf]
tot = c[0]
print(f"rows {n} nq {nq} batches {steps}")
print(f"wave cycles in filtered launches      {tot:>16d}")
print(f"tile epilogues                        {c[1]:>16d}  {c[1] / tot:7.4f}   {c[4]} epilogues, {c[1] / max(c[4], 1):8.1f} cycles each")
print(f"  of which walk of hit columns        {c[2]:>16d}  {c[2] / tot:7.4f}   {c[5]} walked ({c[5] / max(c[4], 1):.3f} of epilogues), {c[6]} columns, {c[7]} appends")
print(f"     cycles per walking epilogue      {c[2] / max(c[5], 1):10.1f}   per column {c[2] / max(c[6], 1):10.1f}   per append {c[2] / max(c[7], 1):10.1f}")
print(f"wait at the barrier after an epilogue {c[3]:>16d}  {c[3] / tot:7.4f}   {c[3] / max(c[4], 1):8.1f} cycles each")
print(f"flushes                               {c[9]:>16d}  {c[9] / tot:7.4f}   {c[8]} flushes (wave count), {c[9] / max(c[8], 1):8.1f} cycles each")
kt = max(c[12], 1)
print(f"K-tiles {c[12]}: phases q0+q1 {c[13] / kt:8.1f} cycles  wait+barrier M {c[10] / kt:8.1f}  phases q2+q3 {c[14] / kt:8.1f}  wait+barrier E {c[11] / max(c[12] - c[4], 1):8.1f}  (64 MFMAs = 1024 cycles per pair of phases)")
print(f"  fractions of wave cycles: phases {(c[13] + c[14]) / tot:7.4f}  barrier M {c[10] / tot:7.4f}  barrier E {c[11] / tot:7.4f}")
print(ix.last_stats())
