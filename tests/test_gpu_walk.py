"""The 4-wave kernel's walk of hit columns reads a 16 x 16 tile's accumulator registers through a computed jump
(`w4_read_acc_dyn`, kernels_mfma.hip): one table entry per (row tile m, query tile n) of a wave's 128 x 128 block.
Plant every query's best row so that the 256 queries of a block cover all 64 (m, n) entries of all four waves,
inside a filtered stage (behind the dense sample): a wrong entry loses the planted row and the result is no
longer the oracle's.  Parity unpinned by the reference (vRod holds no scan): the oracle is build-authored."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = {"f32": 0, "bf16": 1}
ME = {"cosine": 0, "l2": 1}


@pytest.fixture(scope="module")
def va():
    import torch
    assert torch.cuda.is_available()
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("dtype,metric,split", [("bf16", "cosine", None), ("bf16", "l2", None), ("f32", "cosine", "1")])
def test_every_accumulator_tile_of_every_wave_is_walked(va, oracle, dtype, metric, split):
    from conftest import f32_split
    dim, n, nq, k = 128, 200_000, 512, 10
    raw = oracle.synth_rows(41, 0, n, dim, threads=8)
    rq = oracle.synth_rows(42, 0, nq, dim, threads=8)
    # query q of a 256-query block: wave column wc = (q % 256) / 128, query tile n = (q % 128) / 16, lane q % 16.
    # Its planted row sits in row tile m = q % 8 of wave row wr = (q / 8) % 2, row-in-tile (q / 16) % 16:
    # per wave column the 128 queries cover the 8 x 8 (m, n) pairs twice, once per wave row.
    tile0 = 150_000 // 256          # a 256-row corpus tile well inside the last filtered stage; a second one per query block
    planted = np.empty(nq, dtype=np.int64)
    for q in range(nq):
        ql = q % 256
        m, wr, r = ql % 8, (ql // 8) % 2, (ql // 16) % 16
        row = (tile0 + 3 * (q // 256)) * 256 + wr * 128 + m * 16 + r
        planted[q] = row
    assert len(set(planted.tolist())) == nq
    raw[planted] = rq                                    # cosine 1 / distance 0: the query's best row by a wide margin
    oi, osc = oracle.search(raw, rq, k, DT[dtype], ME[metric], threads=8)
    assert np.array_equal(oi[:, 0], planted)
    with f32_split(split), va.Index(dim, dtype, metric) as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    assert st["path"] == va.PATH_MFMA and st["scan_launches"] >= 3, st     # dense sample + filtered stages
    assert np.array_equal(ids, oi), np.argwhere(ids != oi)[:5]
    assert np.array_equal(bits(sc), bits(osc))
