"""GPU parity against the COMMITTED golden fixtures (tests/golden/golden.npz): ids and score
bits produced by the HIP path through the C ABI must equal the stored vectors -- no oracle call
on the comparison side.  Inputs are re-created from the stored seeds by the device-side
generator (itself pinned by `synth_*` fixtures) or taken from the fixture (adversarial sets)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"), allow_pickle=False)
META = json.load(open(os.path.join(HERE, "golden", "golden_meta.json")))
DTYPE = {0: "f32", 1: "bf16"}
METRIC = {0: "cosine", 1: "l2"}


@pytest.fixture(scope="module")
def va():
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


def test_device_generator_matches_golden(va):
    a = va.synth_rows_device(0, 1, 0, 3, 8).cpu().numpy().view(np.uint32)
    assert np.array_equal(a, GOLD["synth_seed1_rows0_3_dim8"])
    b = va.synth_rows_device(0, 2, 123456789, 1, 5).cpu().numpy().view(np.uint32)
    assert np.array_equal(b, GOLD["synth_seed2_row123456789_dim5"])


@pytest.mark.parametrize("name", sorted(META))
@pytest.mark.parametrize("path", [0, 1, 2])  # auto, stream, mfma
def test_hip_path_reproduces_golden(va, name, path):
    m = META[name]
    if m.get("raw"):
        raw, rq = GOLD[name + "__raw"], GOLD[name + "__queries"]
        dim = raw.shape[1]
    else:
        dim = m["dim"]
        rq = va.synth_rows_device(0, m["query_seed"], 0, m["nq"], dim).cpu().numpy()
    with va.Index(dim, DTYPE[m["dtype"]], METRIC[m["metric"]]) as ix:
        if m.get("raw"):
            ix.add(raw)
        else:
            ix.add_synthetic(m["corpus_seed"], 0, m["n"])
        ix.set_path(path)
        ids, sc = ix.search(rq, m["k"])
    assert np.array_equal(ids, GOLD[name + "__ids"]), name
    assert np.array_equal(sc.view(np.uint32), GOLD[name + "__score_bits"]), name
