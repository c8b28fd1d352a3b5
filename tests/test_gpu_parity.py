"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bar: ids AND score bits identical to the oracle for every dtype/metric/path (the
canonical re-score makes even the bf16 path bit-exact, which is stricter than the
1e-4 relative tolerance north_star allows for bf16 distances).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = {"f32": 0, "bf16": 1}
ME = {"cosine": 0, "l2": 1}


@pytest.fixture(scope="module")
def va():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    import vrod_amd
    vrod_amd.load()  # raises if the HIP library is missing: no fallback
    return vrod_amd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same(ids, sc, oi, osc, what=""):
    assert np.array_equal(ids, oi), f"{what}: ids differ at {np.argwhere(ids != oi)[:5]}"
    assert np.array_equal(bits(sc), bits(osc)), f"{what}: score bits differ"


def run_case(va, O, raw, rq, k, dtype, metric, path, id_offset=0):
    from conftest import f32_split
    oi, osc = O.search(raw, rq, k, DT[dtype], ME[metric], id_offset=id_offset)
    # a batched search over an fp32 corpus has two fast passes (the fp32 MFMA kernel, and by default
    # the bf16 split pass over [hi | lo] planes): both must give the oracle's bits
    modes = ("0", None) if dtype == "f32" and path in (0, 2) and rq.shape[0] > 4 else (None,)
    for mode in modes:
        with f32_split(mode), va.Index(raw.shape[1], dtype, metric) as ix:
            ix.add(raw)
            ix.set_id_offset(id_offset)
            ix.set_path(path)
            ids, sc = ix.search(rq, k)
            st = ix.last_stats()
        assert_same(ids, sc, oi, osc, f"{dtype}/{metric}/path{path}/split={mode}")
        if mode == "0":
            assert st["split_pass"] == 0
    return st


# ---------------------------------------------------------------- building blocks
def test_synth_stream_matches_oracle(va, oracle):
    d = va.synth_rows_device(0, 1, 12345, 300, 768).cpu().numpy()
    assert np.array_equal(bits(d), bits(oracle.synth_rows(1, 12345, 300, 768)))
    d = va.synth_rows_device(0, 2, 0, 17, 100).cpu().numpy()
    assert np.array_equal(bits(d), bits(oracle.synth_rows(2, 0, 17, 100)))


def test_fp64_sqrt_div_are_correctly_rounded_at_scale(va, oracle):
    """The generator and the cosine normalisation rely on IEEE fp64 sqrt and divide on the
    device; a 1-ulp difference would show up as rare fp32 mismatches.  256k x 64 elements."""
    d = va.synth_rows_device(0, 7, 1 << 20, 1 << 18, 64).cpu().numpy()
    assert np.array_equal(bits(d), bits(oracle.synth_rows(7, 1 << 20, 1 << 18, 64, threads=8)))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_prepare_matches_oracle(va, oracle, dtype, metric):
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal((1000, 100)) * rng.uniform(1e-3, 1e3, (1000, 1))).astype(np.float32)
    raw[5] = 0.0  # zero row stays zero
    raw[6, :] = 1e-30  # tiny values
    with va.Index(100, dtype, metric) as ix:
        ix.add(raw[:400])
        ix.add(raw[400:])  # second add appends
        got = ix.get_rows(0, 1000)
    assert np.array_equal(bits(got), bits(oracle.prepare(raw, DT[dtype], ME[metric])))


# ---------------------------------------------------------------- search parity
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("metric", ["cosine", "l2"])
@pytest.mark.parametrize("path", [1, 2, 3])  # stream, mfma, exact
def test_search_parity_small(va, oracle, dtype, metric, path):
    rng = np.random.default_rng(11)
    raw = rng.standard_normal((5000, 96)).astype(np.float32)
    rq = rng.standard_normal((5, 96)).astype(np.float32)
    st = run_case(va, oracle, raw, rq, 10, dtype, metric, path)
    assert st["path"] == path


def test_config1_10k_x_128_cosine_top10(va, oracle):
    """BASELINE.json configs[0]: 10k x 128 fp32 cosine, batch 1, top-10 (synthetic stream)."""
    raw = oracle.synth_rows(1, 0, 10000, 128)
    rq = oracle.synth_rows(2, 0, 1, 128)
    run_case(va, oracle, raw, rq, 10, "f32", "cosine", 0)


@pytest.mark.parametrize("path", [1, 2])
def test_768d_l2_top100(va, oracle, path):
    raw = oracle.synth_rows(1, 0, 20000, 768, threads=8)
    rq = oracle.synth_rows(2, 0, 4, 768)
    st = run_case(va, oracle, raw, rq, 100, "f32", "l2", path)
    assert st["max_fast_err"] <= st["eps_bound"]


def test_bf16_cosine_batched_mfma(va, oracle):
    raw = oracle.synth_rows(1, 0, 30000, 768, threads=8)
    rq = oracle.synth_rows(2, 0, 300, 768)
    st = run_case(va, oracle, raw, rq, 10, "bf16", "cosine", 0)
    assert st["path"] == 2
    assert st["max_fast_err"] <= st["eps_bound"]


def test_f32_cosine_batched_mfma_large_k(va, oracle):
    raw = oracle.synth_rows(1, 0, 12000, 1536, threads=8)
    rq = oracle.synth_rows(2, 0, 16, 1536)
    run_case(va, oracle, raw, rq, 1000, "f32", "cosine", 2)


# ---------------------------------------------------------------- edge cases (SURVEY.md 8c list)
@pytest.mark.parametrize("path", [1, 2])
def test_duplicates_tie_break_by_id(va, oracle, path):
    rng = np.random.default_rng(5)
    base = rng.standard_normal((50, 64)).astype(np.float32)
    raw = np.concatenate([base] * 60)  # every row 60 times: massive exact ties
    rq = base[:3] + 0.01 * rng.standard_normal((3, 64)).astype(np.float32)
    st = run_case(va, oracle, raw, rq, 25, "f32", "cosine", path)
    # every query sits next to a base row that exists 60 times -- more than the k' = 33 ... 57 candidates the
    # fast passes keep: the group of exact ties straddles the candidate cut, T equals s_k, no certificate
    # can separate the kept copies from the left-out ones -> all three queries took the second chance
    assert st["fallback_queries"] == 3
    # 30 copies fit inside every k': the certificate holds (ties among candidates are ordered by id) -> no fallback
    st = run_case(va, oracle, np.concatenate([base] * 30), rq, 25, "f32", "cosine", path)
    assert st["fallback_queries"] == 0


@pytest.mark.parametrize("path", [1, 2])
def test_k_equals_n_and_k_greater_than_n(va, oracle, path):
    rng = np.random.default_rng(6)
    raw = rng.standard_normal((37, 33)).astype(np.float32)  # d not a multiple of 32, N not of the tile
    rq = rng.standard_normal((3, 33)).astype(np.float32)
    run_case(va, oracle, raw, rq, 37, "f32", "l2", path)
    run_case(va, oracle, raw, rq, 50, "f32", "cosine", path)  # slots past N are (ID_NONE, NaN)
    run_case(va, oracle, raw, rq, 1, "bf16", "cosine", path)


def test_zero_vector_and_zero_query(va, oracle):
    rng = np.random.default_rng(7)
    raw = rng.standard_normal((300, 40)).astype(np.float32)
    raw[17] = 0.0
    rq = rng.standard_normal((2, 40)).astype(np.float32)
    rq[1] = 0.0  # zero query: all cosine scores are 0 -> ids 0..k-1
    for path in (1, 2):
        run_case(va, oracle, raw, rq, 12, "f32", "cosine", path)


def test_id_offset_and_empty_index(va, oracle):
    rng = np.random.default_rng(8)
    raw = rng.standard_normal((500, 64)).astype(np.float32)
    rq = rng.standard_normal((2, 64)).astype(np.float32)
    run_case(va, oracle, raw, rq, 5, "f32", "cosine", 0, id_offset=10_000_000_000)
    with va.Index(64, "f32", "cosine") as ix:
        ids, sc = ix.search(rq, 4)
        assert (ids == va.ID_NONE).all() and np.isnan(sc).all()


def test_rejects_nan_and_bad_args(va):
    with va.Index(16, "f32", "cosine") as ix:
        bad = np.zeros((4, 16), np.float32)
        bad[2, 3] = np.nan
        with pytest.raises(va.VrodError) as e:
            ix.add(bad)
        assert e.value.code == 2 and ix.count == 0
        ix.add(np.ones((4, 16), np.float32))
        q = np.ones((1, 16), np.float32)
        q[0, 0] = np.inf
        with pytest.raises(va.VrodError) as e:
            ix.search(q, 1)
        assert e.value.code == 2
        with pytest.raises(va.VrodError):
            ix.search(np.ones((1, 16), np.float32), 0)
        with pytest.raises(va.VrodError):
            ix.search(np.ones((1, 16), np.float32), va.MAX_K + 1)


@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_rejected_add_leaves_the_certificate_bound_alone(va, oracle, dtype, metric):
    """A rejected add must not leave its rows' norms in the handle's max |x|^2: on an L2 index one
    Inf row used to void every later certificate (every query of every later search took the exact
    path, silently), and large finite rejected rows widened the bound the same way."""
    dim, n, nq = 96, 30000, 24
    raw = oracle.synth_rows(1, 0, n, dim, threads=4)
    rq = oracle.synth_rows(2, 0, nq, dim)
    bad = (raw[:512] * np.float32(3e18)).astype(np.float32)     # huge but finite rows ...
    bad[100, 5] = np.inf                                        # ... and one Inf: the add is rejected as a whole
    oi, osc = oracle.search(raw, rq, 10, DT[dtype], ME[metric])
    for path in (1, 2):
        with va.Index(dim, dtype, metric) as ix:
            ix.add(raw[:20000])
            with pytest.raises(va.VrodError) as e:
                ix.add(bad)
            assert e.value.code == 2 and ix.count == 20000
            ix.add(raw[20000:])
            ix.set_path(path)
            ids, sc = ix.search(rq, 10)
            st = ix.last_stats()
        assert_same(ids, sc, oi, osc, f"after a rejected add, path {path}")
        assert st["fallback_queries"] == 0, st
        assert st["eps_bound"] < 1e-2, st


def test_denormal_products_follow_ieee(va, oracle):
    raw = (np.random.default_rng(9).standard_normal((200, 32)) * 1e-22).astype(np.float32)
    rq = (np.random.default_rng(10).standard_normal((2, 32)) * 1e-22).astype(np.float32)
    run_case(va, oracle, raw, rq, 7, "f32", "l2", 1)  # products ~1e-44: subnormal range


# ---------------------------------------------------------------- shard merge
def test_merge_topk_device_matches_oracle(va, oracle):
    import torch
    rng = np.random.default_rng(12)
    raw = rng.standard_normal((4000, 64)).astype(np.float32)
    rq = rng.standard_normal((9, 64)).astype(np.float32)
    k, G = 10, 4
    ids_l, sc_l = [], []
    for g in range(G):
        lo, hi = g * 1000, (g + 1) * 1000
        with va.Index(64, "f32", "cosine") as ix:
            ix.add(raw[lo:hi])
            ix.set_id_offset(lo)
            i, s = ix.search(rq, k)
        ids_l.append(i)
        sc_l.append(s)
    ids = torch.from_numpy(np.stack(ids_l).view(np.int64)).cuda()
    sc = torch.from_numpy(np.stack(sc_l)).cuda()
    mi, ms = va.merge_topk_device(0, "cosine", ids, sc)
    oi, osc = oracle.search(raw, rq, k, 0, 0)
    assert_same(mi.cpu().numpy().view(np.uint64), ms.cpu().numpy(), oi, osc, "merge")
    # and against the oracle's own merge of the oracle's own shard results
    mi2, ms2 = oracle.merge_topk(np.stack(ids_l), np.stack(sc_l), 0)
    assert_same(mi2, ms2, oi, osc, "oracle merge")


# ---------------------------------------------------------------- scale / structure cases
@pytest.mark.parametrize("dtype,metric,nq,split", [("bf16", "cosine", 12, None), ("bf16", "l2", 12, None), ("bf16", "l2", 70, None),
                                                   ("f32", "cosine", 40, "1"), ("f32", "l2", 70, "1"), ("f32", "l2", 12, "1")])
def test_mfma_path_splits_launches_past_2pow24_rows(va, oracle, dtype, metric, nq, split):
    """Rows are addressed relative to a launch's first tile with 24 bits: a 19M-row shard needs
    two launches for its last stage (sample 64k rows -> stage to ~1.1M -> 19M rows).  64-d rows
    keep it small (2.4 GB bf16, 4.9 GB fp32 + as much for the [hi | lo] planes); oracle on 8 threads.
    Covered: the skinny form (12 queries) and the 256-query tile (70), both metrics, and the SPLIT
    form of both over an fp32 corpus."""
    from conftest import f32_split
    n, dim, k = (1 << 24) + 2_300_001, 64, 10
    with f32_split(split), va.Index(dim, dtype, metric) as ix:
        ix.add_synthetic(1, 0, n)
        ix.set_path(va.PATH_MFMA)
        rq = oracle.synth_rows(2, 0, nq, dim)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    raw = oracle.synth_rows(1, 0, n, dim, threads=8)
    oi, osc = oracle.search(raw, rq, k, DT[dtype], ME[metric], threads=8)
    assert_same(ids, sc, oi, osc, "19M rows")
    assert st["fallback_queries"] == 0 and st["scan_launches"] >= 4
    assert st["split_pass"] == (1 if split == "1" else 0)


@pytest.mark.parametrize("dtype,metric", [("bf16", "l2"), ("f32", "l2"), ("bf16", "cosine")])
def test_mfma_padded_batch_300k_rows(va, oracle, dtype, metric):
    """Batch of 300 pads to 512 query columns; unnormalised rows exercise the L2 norm expansion."""
    rng = np.random.default_rng(21)
    raw = (oracle.synth_rows(1, 0, 300_000, 128, threads=8) * rng.uniform(0.5, 2.0, (300_000, 1))).astype(np.float32)
    rq = (oracle.synth_rows(2, 0, 300, 128) * rng.uniform(0.5, 2.0, (300, 1))).astype(np.float32)
    with va.Index(128, dtype, metric) as ix:
        ix.add(raw)
        ids, sc = ix.search(rq, 10)
        st = ix.last_stats()
    oi, osc = oracle.search(raw, rq, 10, DT[dtype], ME[metric], threads=8)
    assert_same(ids, sc, oi, osc, f"{dtype}/{metric} 300x300k")
    assert st["path"] == 2 and st["max_fast_err"] <= st["eps_bound"]


def test_synthetic_add_equals_host_add(va, oracle):
    """vrod_index_add_synthetic (generated in HBM) == vrod_index_add of the oracle's stream."""
    raw = oracle.synth_rows(1, 1000, 5000, 96)
    with va.Index(96, "bf16", "cosine") as a, va.Index(96, "bf16", "cosine") as b:
        a.add_synthetic(1, 1000, 5000)
        b.add(raw)
        assert np.array_equal(bits(a.get_rows(0, 5000)), bits(b.get_rows(0, 5000)))


def test_mfma_small_corpus_everything_appended(va, oracle):
    """N <= list capacity: no sample pass, thresholds stay 'worst', every (row, query) pair goes
    through the LDS log (overflowing it: direct global appends) and the lists hold all N rows."""
    rng = np.random.default_rng(33)
    raw = rng.standard_normal((8000, 64)).astype(np.float32)
    rq = rng.standard_normal((300, 64)).astype(np.float32)
    for dtype in ("bf16", "f32"):
        st = run_case(va, oracle, raw, rq, 10, dtype, "cosine", 2)
        assert st["scan_launches"] == 1 and st["fallback_queries"] == 0


def test_merge_topk_packed_device(va, oracle):
    """The single-all-gather layout: per shard nq*k ids (u64) then nq*k scores (f32) in one block."""
    import torch
    from vrod_amd.shard import alloc_packed
    rng = np.random.default_rng(13)
    raw = rng.standard_normal((3000, 32)).astype(np.float32)
    rq = rng.standard_normal((6, 32)).astype(np.float32)
    k, G = 10, 3
    blocks = []
    for g in range(G):
        lo, hi = g * 1000, (g + 1) * 1000
        with va.Index(32, "f32", "l2") as ix:
            ix.add(raw[lo:hi])
            ix.set_id_offset(lo)
            packed, pi, ps = alloc_packed(6, k, torch.device("cuda", 0))
            ix.search_device(torch.from_numpy(rq).cuda(), k, pi, ps)
            blocks.append(packed)
    gathered = torch.cat(blocks)
    mi = torch.empty((6, k), dtype=torch.int64, device="cuda")
    ms = torch.empty((6, k), dtype=torch.float32, device="cuda")
    va.merge_topk_packed_device(0, "l2", gathered, G, 6, k, mi, ms)
    oi, osc = oracle.search(raw, rq, k, 0, 1)
    assert_same(mi.cpu().numpy().view(np.uint64), ms.cpu().numpy(), oi, osc, "packed merge")


@pytest.mark.parametrize("path", [1, 2])
def test_max_k(va, oracle, path):
    """k = VROD_MAX_K (3584): k' = 4032 candidates per query, the widest select windows."""
    raw = oracle.synth_rows(1, 0, 30000, 64, threads=8)
    rq = oracle.synth_rows(2, 0, 3, 64)
    run_case(va, oracle, raw, rq, va.MAX_K, "f32", "cosine", path)


@pytest.mark.parametrize("metric", ["cosine", "l2"])
@pytest.mark.parametrize("dim,nq", [(40, 5), (129, 16), (300, 17), (768, 33), (768, 64), (1100, 64), (1536, 32), (1536, 40), (2300, 9)])
def test_bf16_skinny_kernel_batches_up_to_64(va, oracle, dim, nq, metric):
    """5..64 queries over bf16 rows: the skinny MFMA kernel (rows streamed from HBM into A fragments,
    queries resident in LDS; 32- or 64-query form by what fits in LDS -- (1536, 40) and (2300, 9) do
    not fit and take the tiled kernel).  Odd line counts per row, staged plan (N > list capacity: dense
    sample + filtered stages), a last 64-row block that is partly padding, unnormalised rows for L2."""
    rng = np.random.default_rng(dim * 100 + nq)
    n, k = 20333, 10
    raw = (rng.standard_normal((n, dim)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    rq = rng.standard_normal((nq, dim)).astype(np.float32)
    st = run_case(va, oracle, raw, rq, k, "bf16", metric, 2)
    assert st["path"] == 2 and st["scan_launches"] >= 2
    st = run_case(va, oracle, raw[:700], rq, 50, "bf16", metric, 2)      # everything appended, one launch
    assert st["path"] == 2


def test_bf16_skinny_kernel_heavy_appends_and_ties(va, oracle):
    """Thousands of identical rows beat every threshold at once: the wave-private log segments
    overflow into direct appends, the lists overflow their capacity, and the certificate must
    send those queries to the exact path."""
    rng = np.random.default_rng(91)
    base = rng.standard_normal((3000, 128)).astype(np.float32)
    dup = np.repeat(rng.standard_normal((1, 128)).astype(np.float32), 9000, axis=0)
    raw = np.concatenate([base, dup, base[:2000] * 1.5])
    rq = rng.standard_normal((40, 128)).astype(np.float32)
    rq[::3] = dup[0] + 0.05 * rng.standard_normal((14, 128)).astype(np.float32)
    for metric in ("cosine", "l2"):
        st = run_case(va, oracle, raw, rq, 20, "bf16", metric, 2)
        assert st["fallback_queries"] >= 14


@pytest.mark.parametrize("metric", ["cosine", "l2"])
@pytest.mark.parametrize("dim", [40, 129, 300, 768])   # bf16 K-tiles per row: 1, 3 (odd), 5, 12
def test_bf16_4wave_kernel_ktile_counts(va, oracle, dim, metric):
    """The 4-wave MFMA kernel alternates two register roles per K-tile and peels the first K-tile
    of every corpus tile (C = 0 form): odd and single K-tile rows, two query blocks (one padded),
    staged plan (N > list capacity), unnormalised rows for the L2 norm path."""
    rng = np.random.default_rng(dim)
    n, nq, k = 21000, 300, 10
    raw = (rng.standard_normal((n, dim)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    rq = rng.standard_normal((nq, dim)).astype(np.float32)
    st = run_case(va, oracle, raw, rq, k, "bf16", metric, 2)
    assert st["path"] == 2 and st["scan_launches"] >= 2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_denormal_score_range_is_inside_the_bound(va, oracle, dtype):
    """Squared distances around 1e-40 are denormal: rounding errors there are absolute (half a
    denormal ulp each), not relative, and the certificate's bound carries a term for them."""
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal((20000, 96)) * 1e-21).astype(np.float32)
    rq = (rng.standard_normal((40, 96)) * 1e-21).astype(np.float32)
    st = run_case(va, oracle, raw, rq, 10, dtype, "l2", 2)
    assert st["max_fast_err"] <= st["eps_bound"], st


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("scale", [1e18, 3e18, 1e19])
def test_l2_distances_that_overflow_fp32(va, oracle, path, scale):
    """Squared distances beyond FLT_MAX are +inf in the canonical chain too (ties -> smaller id).
    The fast scores are inf / NaN there and T = +inf stops meaning "nothing was left out": the
    certificate must refuse and the exact path must answer."""
    rng = np.random.default_rng(11)
    raw = (rng.standard_normal((9000, 96)) * scale).astype(np.float32)
    raw[17] = 0.0                      # one row at a finite distance from the zero query
    rq = (rng.standard_normal((5, 96)) * scale).astype(np.float32)
    rq[0] = 0.0
    st = run_case(va, oracle, raw, rq, 10, "f32", "l2", path)
    if scale >= 3e18:                  # (1e18: distances ~1e38, still finite: the fast passes may certify)
        assert st["fallback_queries"] == 5


@pytest.mark.parametrize("dim", [4096, 12288, 32768])
def test_long_rows_every_path(va, oracle, dim):
    """A scan pass keeps its queries in LDS: 8 of them up to d = 4096, fewer for longer rows
    (VROD_MAX_DIM = 32768: one fp32 row next to the tiles of the prepare / re-score kernels)."""
    rng = np.random.default_rng(dim)
    n = 2500 if dim <= 12288 else 700
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    for dtype, metric, path, nq in (("f32", "l2", 1, 9), ("bf16", "cosine", 1, 3), ("f32", "cosine", 2, 9), ("bf16", "l2", 2, 9), ("f32", "cosine", 3, 1)):
        rq = rng.standard_normal((nq, dim)).astype(np.float32)
        run_case(va, oracle, raw, rq, 5, dtype, metric, path)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_exact_path_batches_its_queries(va, oracle, dtype, metric):
    """The exact path walks the corpus once per group of 8 / 4 / 2 / 1 uncertified queries
    (15 = one group of each size); rows and dims off every tile boundary."""
    rng = np.random.default_rng(77)
    raw = rng.standard_normal((9001, 203)).astype(np.float32)
    rq = rng.standard_normal((15, 203)).astype(np.float32)
    st = run_case(va, oracle, raw, rq, 33, dtype, metric, 3)
    assert st["fallback_queries"] == 15


def test_exact_path_groups_of_scattered_failures(va, oracle):
    """Only some queries of a batch fail their certificate (those aimed at a block of > k'
    identical rows): the groups of the exact path gather non-adjacent query rows."""
    rng = np.random.default_rng(78)
    base = rng.standard_normal((3000, 64)).astype(np.float32)
    dup = np.repeat(rng.standard_normal((1, 64)).astype(np.float32), 200, axis=0)
    raw = np.concatenate([base[:1500], dup, base[1500:]])
    rq = rng.standard_normal((21, 64)).astype(np.float32)
    hit = [0, 3, 4, 9, 10, 11, 17, 20]
    rq[hit] = dup[0] + 0.01 * rng.standard_normal((len(hit), 64)).astype(np.float32)
    for path in (1, 2):
        st = run_case(va, oracle, raw, rq, 20, "f32", "cosine", path)
        assert st["fallback_queries"] == len(hit)


def test_exact_path_long_rows_fewer_queries_per_pass(va, oracle):
    """d = 12288: two fp32 query rows no longer fit beside the staging tile in 64 KB of LDS,
    so a pass of the exact path takes one query."""
    rng = np.random.default_rng(79)
    raw = rng.standard_normal((700, 12288)).astype(np.float32)
    rq = rng.standard_normal((3, 12288)).astype(np.float32)
    run_case(va, oracle, raw, rq, 5, "bf16", "l2", 3)
    raw = rng.standard_normal((900, 3000)).astype(np.float32)   # 4 rows of 3008 floats fit, 8 do not
    rq = rng.standard_normal((7, 3000)).astype(np.float32)
    run_case(va, oracle, raw, rq, 5, "f32", "cosine", 3)


def test_dim_beyond_the_limit_is_rejected(va):
    with pytest.raises(va.VrodError) as e:
        va.Index(32769, "f32", "cosine")
    assert e.value.code == 1
