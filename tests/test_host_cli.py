"""Host mirror of vRod's command API (vrod_amd/host, C++) through the `vrod` CLI.

CPU tests: the parts the reference actually implements or shapes -- --init-database
(src/main.rs:51-62, src/database/setup.rs:3-26), the flag set (main.rs:10-34), the
CommandBuilder dispatch and its error (src/command/builder.rs:22-81) -- plus the loud
failure of device commands without a GPU.  GPU test: BULKINSERT + SEARCHSIMILAR end to end
against the oracle.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VROD = os.path.join(ROOT, "vrod_amd", "vrod")


def run(*args, cwd=None, env=None):
    return subprocess.run([VROD, *args], capture_output=True, text=True, cwd=cwd, env=env)


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(VROD):
        subprocess.run(["make", "-C", os.path.join(ROOT, "vrod_amd", "host")], check=True)


def test_no_args_prints_help():
    r = run()
    assert r.returncode == 2 and "--init-database" in r.stdout and "--command-arg" in r.stdout


def test_init_database_layout_and_errors(tmp_path):
    r = run("-i", str(tmp_path), "-n", "db1")
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(tmp_path / "db1")) == ["vr_config", "vr_wal"]
    assert os.path.getsize(tmp_path / "db1" / "vr_wal") == 0
    r = run("--init-database", str(tmp_path), "--init-database-name", "db1")
    assert r.returncode == 1
    assert f"Directory with the name 'db1' already exists in '{tmp_path}'" in r.stderr
    r = run("-i", str(tmp_path))
    assert r.returncode == 1
    assert "Missing '--init_database_name' flag with argument for '--init_database' flag." in r.stderr


def test_command_dispatch_names_and_error(tmp_path):
    assert run("-i", str(tmp_path), "-n", "d").returncode == 0
    db = str(tmp_path / "d")
    r = run("-d", db, "-e", "NoSuchCommand")
    assert r.returncode == 1 and "Unrecognized command: NoSuchCommand" in r.stderr
    # case-insensitive names (builder.rs:29 upper-cases); the stubbed ones are accepted and do nothing
    for name in ("truncatewal", "Update", "DELETE", "search", "ReIndex"):
        assert run("-d", db, "-c", "x", "-e", name, "-a", "y").returncode == 0, name
    assert run("-d", db, "-e", "create", "-a", "words metric=l2 dtype=bf16").returncode == 0
    assert open(tmp_path / "d" / "words" / "vr_config").read().split() == ["dim=0", "metric=l2", "dtype=bf16", "count=0"]
    r = run("-d", db, "-e", "LISTCOLLECTIONS")
    assert r.returncode == 0 and r.stdout.split() == ["words"]
    assert run("-d", db, "-e", "CREATE", "-a", "words").returncode == 1       # already exists
    assert run("-d", db, "-e", "DROP", "-a", "words").returncode == 0
    assert run("-d", db, "-e", "LISTCOLLECTIONS").stdout.strip() == ""
    assert run("-d", db, "-c", "nope", "-e", "SEARCHSIMILAR", "-a", "1,2").returncode == 1
    assert run("-d", str(tmp_path / "missing"), "-e", "LISTCOLLECTIONS").returncode == 1


def test_device_commands_fail_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert run("-i", str(tmp_path), "-n", "d").returncode == 0
    db = str(tmp_path / "d")
    assert run("-d", db, "-e", "CREATE", "-a", "c").returncode == 0
    r = run("-d", db, "-c", "c", "-e", "INSERT", "-a", "1,0,0;x")
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_bulkinsert_and_searchsimilar_match_oracle(tmp_path, oracle):
    n, dim, nq, k = 3000, 48, 5, 7
    raw = oracle.synth_rows(1, 0, n, dim)
    rq = oracle.synth_rows(2, 0, nq, dim)
    # the reference's text format: f32::to_string values, comma-joined, ';' + word (embeddings.rs:55-61)
    emb = tmp_path / "alice_embeddings.txt"
    with open(emb, "w") as f:
        for i in range(n):
            f.write(",".join(repr(float(v)) for v in raw[i]) + f";word{i}\n")
    assert run("-i", str(tmp_path), "-n", "d").returncode == 0
    db = str(tmp_path / "d")
    assert run("-d", db, "-e", "CREATE", "-a", "alice metric=cosine dtype=f32").returncode == 0
    r = run("-d", db, "-c", "alice", "-e", "BULKINSERT", "-a", str(emb))
    assert r.returncode == 0, r.stderr
    assert "count=3000" in open(tmp_path / "d" / "alice" / "vr_config").read()
    # a second process: Database::load + lazy load of the collection into HBM
    qarg = f"k={k};" + ";".join(",".join(repr(float(v)) for v in rq[i]) for i in range(nq))
    r = run("-d", db, "-c", "alice", "-e", "SEARCHSIMILAR", "-a", qarg)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in r.stdout.strip().split("\n")]
    assert len(rows) == nq * k
    ids = np.array([int(x[2]) for x in rows], dtype=np.uint64).reshape(nq, k)
    sc = np.array([float(x[3]) for x in rows], dtype=np.float32).reshape(nq, k)
    oi, osc = oracle.search(raw, rq, k, 0, 0)
    assert np.array_equal(ids, oi)
    assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32))   # %.9g round-trips fp32
    assert rows[0][4] == f"word{int(oi[0, 0])}"
    # INSERT appends one more vector (a copy of query 0 -> it becomes query 0's best hit, id n)
    line = ",".join(repr(float(v)) for v in rq[0]) + ";the-query"
    assert run("-d", db, "-c", "alice", "-e", "INSERT", "-a", line).returncode == 0
    r = run("-d", db, "-c", "alice", "-e", "SEARCHSIMILAR", "-a", "k=1;" + ",".join(repr(float(v)) for v in rq[0]))
    assert r.returncode == 0 and r.stdout.split("\t")[2] == str(n) and r.stdout.strip().endswith("the-query")
    # query file form
    qf = tmp_path / "q.txt"
    qf.write_text("\n".join(",".join(repr(float(v)) for v in rq[i]) + ";ignored" for i in range(nq)) + "\n")
    r = run("-d", db, "-c", "alice", "-e", "SEARCHSIMILAR", "-a", f"k=3;@{qf}")
    assert r.returncode == 0 and len(r.stdout.strip().split("\n")) == nq * 3
    # leftovers of an insert that never reached its commit point (rows and payload lines beyond vr_config's
    # count): the next insert must land at id n + 1, not behind the orphans
    cdir = tmp_path / "d" / "alice"
    with open(cdir / "vr_vectors", "ab") as f:
        f.write(np.full(3 * dim + 5, 7.0, np.float32).tobytes())       # three orphan rows and a torn fourth
    with open(cdir / "vr_payloads", "a") as f:
        f.write("orphan-a\norphan-b\ntorn")
    # (a search BEFORE that insert must not see the orphan lines as payloads either)
    r = run("-d", db, "-c", "alice", "-e", "SEARCHSIMILAR", "-a", "k=1;" + ",".join(repr(float(v)) for v in rq[0]))
    assert r.returncode == 0 and r.stdout.split("\t")[2] == str(n) and r.stdout.strip().endswith("the-query")
    line = ",".join(repr(float(v)) for v in rq[1]) + ";second-query"
    assert run("-d", db, "-c", "alice", "-e", "INSERT", "-a", line).returncode == 0
    assert os.path.getsize(cdir / "vr_vectors") == (n + 2) * dim * 4
    assert open(cdir / "vr_payloads").read().split("\n")[n:] == ["the-query", "second-query", ""]
    assert "count=3002" in open(cdir / "vr_config").read() and not os.path.exists(cdir / "vr_config.tmp")
    r = run("-d", db, "-c", "alice", "-e", "SEARCHSIMILAR", "-a", "k=1;" + ",".join(repr(float(v)) for v in rq[1]))
    assert r.returncode == 0 and r.stdout.split("\t")[2] == str(n + 1) and r.stdout.strip().endswith("second-query")


@pytest.mark.gpu
def test_searchsimilar_over_a_multi_device_collection(tmp_path, oracle):
    """VROD_DEVICES=0,0: the collection is one multi-device handle (two shards on device 0 here);
    SEARCHSIMILAR prints the same ids and scores as the single-device run."""
    n, dim, nq, k = 70000, 8, 2, 5          # 70000 rows: both shards hold rows
    raw = oracle.synth_rows(3, 0, n, dim, threads=8)
    rq = oracle.synth_rows(4, 0, nq, dim)
    f32 = tmp_path / "rows.f32"
    raw.tofile(f32)
    assert run("-i", str(tmp_path), "-n", "d").returncode == 0
    db = str(tmp_path / "d")
    env = dict(os.environ, VROD_DEVICES="0,0")
    assert run("-d", db, "-e", "CREATE", "-a", "c metric=l2 dtype=f32", env=env).returncode == 0
    r = run("-d", db, "-c", "c", "-e", "BULKINSERT", "-a", f"{f32}:{dim}", env=env)
    assert r.returncode == 0, r.stderr
    qarg = f"k={k};" + ";".join(",".join(repr(float(v)) for v in rq[i]) for i in range(nq))
    outs = []
    for e in (env, None):
        r = run("-d", db, "-c", "c", "-e", "SEARCHSIMILAR", "-a", qarg, env=e)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1]
    rows = [l.split("\t") for l in outs[0].strip().split("\n")]
    ids = np.array([int(x[2]) for x in rows], dtype=np.uint64).reshape(nq, k)
    oi, _ = oracle.search(raw, rq, k, 0, 1)
    assert np.array_equal(ids, oi)
    # one query, k = 5: nq*k odd (the per-shard blocks of the exchange must stay aligned)
    r = run("-d", db, "-c", "c", "-e", "SEARCHSIMILAR", "-a", f"k=5;" + ",".join(repr(float(v)) for v in rq[1]), env=env)
    assert r.returncode == 0, r.stderr
    assert [int(l.split("\t")[2]) for l in r.stdout.strip().split("\n")] == oi[1].tolist()
