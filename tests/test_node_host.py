"""The NodeJS host (lib/emf/*.js + the N-API addon) -- the language the reference's host is
written in.  CPU: API shape, error behaviour, partitioner parity with the Python mirror.
GPU: a warm-started train() must reproduce the Python host's result files bit for bit (both
drive the same libycnr_als.so)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import make_problem
from ycnr_als.emf import split_to_portions

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ADDON = os.path.join(ROOT, "you-can-not-recommend_amd", "addon", "ycnr_als.node")

pytestmark = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists(ADDON),
                                reason="node or the built addon is missing")


def problem(seed=4, users=40, items=25):
    bu, bi, U, V = make_problem(users, items, 8, density=0.35, seed=seed, min_per_row=1, empty_rows=(3,))
    user = np.repeat(np.arange(users), bu.counts()).astype(np.int32)
    rng = np.random.default_rng(seed)
    typ = rng.choice([1, 1, 1, 1, 1, 1, 2, 3], size=bu.nnz).astype(np.int8)
    return bu, user, typ, U, V


def test_js_host_on_cpu(tmp_path):
    bu, user, typ, U, V = problem()
    inp = {"dir": str(tmp_path), "rip": 30, "threads": 2, "users": bu.rows, "items": bu.cols,
           "user": user.tolist(), "item": bu.indx.tolist(), "rating": bu.vals.tolist(), "type": typ.tolist()}
    r = subprocess.run(["node", os.path.join(HERE, "js", "host_cpu.js"), json.dumps(inp)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    train = typ <= 2
    cnt_u = np.bincount(user[train], minlength=bu.rows)
    cnt_i = np.bincount(bu.indx[train], minlength=bu.cols)
    assert out["trainNnz"] == int(train.sum())
    for step, cnt, rows_cnt, pct in (("byUser", cnt_u, (cnt_u > 0).sum(), 0), ("byItem", cnt_i, (cnt_i > 0).sum(), 0),
                                     ("rmseValidate", cnt_u, (cnt_u > 0).sum(), 11), ("rmseTest", cnt_u, (cnt_u > 0).sum(), 6)):
        ends, rip, mrows = split_to_portions(cnt, int(rows_cnt), 30, 2, pct)
        assert out["portionsRowIdTo"][step] == ends.tolist(), step
        assert out["maxRatingsInPortion"][step] == rip and out["maxRowsInPortion"][step] == mrows
    assert abs(out["totalRatingsAvg"] - float(bu.vals.mean())) < 1e-6
    from ycnr_als.emf import rebalanced_ranges, shard_ranges
    for key, got in out["shards"].items():
        w, k = (int(x) for x in key.split("_"))
        assert got == shard_ranges(cnt_u, w, k).tolist(), key
    for key, got in out["recut"].items():  # the feedback re-cut: both hosts cut alike from the same measured times
        w, k = (int(x) for x in key.split("_"))
        ms = [1.0 + 0.5 * r for r in range(w)]
        want = rebalanced_ranges(cnt_u, shard_ranges(cnt_u, w, k), ms, k)
        assert got == want.tolist(), key
        assert got[1] > shard_ranges(cnt_u, w, k)[1]  # the rank that ran shortest takes rows from the others
    # N2's opt-in flag: the reference packer's row tables and the data set its passes really consume, against the Python mirror
    from ycnr_als.data import Csr, csr_to_portion, drop_last_rating_per_portion, transpose_csr
    import torch
    tu = Csr(bu.rows, bu.cols, np.concatenate([[0], np.cumsum(cnt_u)]).astype(np.int64), bu.indx[train], bu.vals[train])
    ends_u, _, _ = split_to_portions(cnt_u, int((cnt_u > 0).sum()), 30, 2)
    b = 0
    for p, e in enumerate(ends_u):
        assert out["quirk"]["byUser"][p] == csr_to_portion(tu, b, int(e), dropLastRatingPerPortion=True)[0].tolist(), p
        b = int(e)
    assert out["quirk"]["byUserRowPtr"] == drop_last_rating_per_portion(tu, ends_u).rowPtr.tolist()
    ti = transpose_csr(tu.to("cpu")).numpy()
    ends_i, _, _ = split_to_portions(cnt_i, int((cnt_i > 0).sum()), 30, 2)
    di = drop_last_rating_per_portion(ti, ends_i)
    assert out["quirk"]["byItemRowPtr"] == di.rowPtr.tolist() and out["quirk"]["byItemIndx"] == di.indx.tolist()
    assert ti.nnz - di.nnz == len(ends_i)
    if "prepareError" in out:  # no GPU here: loud failure, carrying the library's message
        assert "hip" in out["prepareError"].lower()


def test_js_lord_rejects_when_a_gpu_process_dies():
    """EmfLord.trainOnGpus: a per-GPU process killed by a signal reports code === null; the Lord must
    reject and kill the survivors instead of waiting for 'trained' forever (and an overall time limit
    covers a process that neither answers nor dies)."""
    script = os.path.join(HERE, "js", "lord_death.js")
    r = subprocess.run(["node", script], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["outcome"] == "rejected" and "SIGKILL" in out["error"] and out["ms"] < 20000, out
    env = dict(os.environ, YCNR_TEST_DIE_SIGNAL="SIGSTOP", YCNR_TEST_TIMEOUT_MS="1500")
    r = subprocess.run(["node", script], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["outcome"] == "rejected" and "no result within" in out["error"], out


@pytest.mark.gpu
@pytest.mark.parametrize("double", [False, True])
def test_js_train_matches_python_host(tmp_path, double):
    from ycnr_als.data import Csr
    from ycnr_als.emf import Dataset, EmfLord
    import torch
    dt = np.float64 if double else np.float32
    bu, user, typ, U, V = problem(seed=6, users=60, items=45)
    bu.vals = bu.vals.astype(dt)

    def sub(mask):
        cnt = np.bincount(user[mask], minlength=bu.rows)
        rp = np.zeros(bu.rows + 1, np.int64)
        rp[1:] = np.cumsum(cnt)
        return Csr(bu.rows, bu.cols, rp, bu.indx[mask].copy(), bu.vals[mask].copy())

    tr = sub(typ <= 2)
    # by item: sort the train triplets by (item, user)
    tu, ti, tv = user[typ <= 2], bu.indx[typ <= 2], bu.vals[typ <= 2]
    o = np.lexsort((tu, ti))
    rp = np.zeros(bu.cols + 1, np.int64)
    rp[1:] = np.cumsum(np.bincount(ti, minlength=bu.cols))
    tri = Csr(bu.cols, bu.rows, rp, tu[o].astype(np.int32), tv[o].copy())
    ds = Dataset(tr, tri, sub(typ == 2), sub(typ == 3), float(bu.vals.astype(np.float64).mean()))
    pydir, jsdir = tmp_path / "py", tmp_path / "js"
    opts = {"factorsCount": 8, "trainIters": 3, "useDoublePrecision": double, "dbType": "ml",
            "ratingsInPortionForRmse": 30, "numThreadsForTrain": {"als": 2}}
    seedlord = EmfLord(options=dict(opts, dataDir=str(jsdir), trainIters=0))
    seedlord.prepareToTrain(ds, U.astype(dt), V.astype(dt))
    seedlord.train()  # zero iterations: just writes the starting factors as ml_factors_ready
    seedlord.destroy()
    lord = EmfLord(options=dict(opts, dataDir=str(pydir)))
    lord.prepareToTrain(ds, U.astype(dt), V.astype(dt))
    hist = lord.train()
    lord.destroy()
    inp = {"dir": str(jsdir), "k": 8, "iters": 3, "rip": 30, "threads": 2, "useDoublePrecision": double,
           "users": bu.rows, "items": bu.cols, "user": user.tolist(), "item": bu.indx.tolist(),
           "rating": bu.vals.tolist(), "type": typ.tolist()}
    (tmp_path / "in.json").write_text(json.dumps(inp))
    r = subprocess.run(["node", os.path.join(HERE, "js", "train_gpu.js"), str(tmp_path / "in.json")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for name in ("user_factors", "item_factors"):
        a = np.fromfile(pydir / "ml_factors_ready" / name, dt)
        b = np.fromfile(jsdir / "ml_factors_ready" / name, dt)
        assert np.array_equal(a, b), name
    for hp, hj in zip(hist, out["history"]):
        for key in ("rmseValidate", "rmseTest", "rmseTestShifted", "globalAvgShift"):
            assert abs(hp[key] - hj[key]) < 1e-12, key
    cj = json.loads((jsdir / "ml_factors_ready" / "calc_info.json").read_text())
    cp = json.loads((pydir / "ml_factors_ready" / "calc_info.json").read_text())
    assert list(cj) == list(cp)
    assert cj["calcCnt"] == 2 and cp["calcCnt"] == 1  # the JS run warm-started from a saved calc
    assert out["checkpoints"] == len(out["history"])    # one save per iteration: n - 1 checkpoints + the final one
    assert out["portionRatings"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("double,transport,sharding", [(False, "shm", "rows"), (True, "shm", "rows"), (False, "ipc", "rows"), (False, "ipc", "bands"), (True, "ipc", "bands")])
def test_js_train_on_per_gpu_processes(tmp_path, double, transport, sharding):
    """The NodeJS Lord forks 2 and 3 per-GPU processes (here sharing cuda:0: the host-staged stand-in 'shm' and the
    device-to-device transport 'ipc'): sharded upload, pipelined pieces, exchange, all-reduce and the re-cut of the
    shards after the first iteration through the addon.  Result files and RMSE history must equal the single-process run's.
    sharding = 'bands': options.itemStepSharding -- the users in 8 bands, the items' Gramians reduce-scattered, the user matrix exchanged
    once at the end (setRatingsBanded / deferExchange of the addon); 1, 2 and 4 processes."""
    dt = np.float64 if double else np.float32
    bu, user, typ, U, V = problem(seed=9, users=120, items=70)
    res = {}
    worlds = (1, 2, 4) if sharding == "bands" else (1, 2, 3)
    for world in worlds:
        d = tmp_path / f"w{world}"
        inp = {"dir": str(d), "k": 12, "iters": 3, "rip": 40, "threads": 2, "useDoublePrecision": double, "world": world, "transport": transport, "sharding": sharding,
               "users": bu.rows, "items": bu.cols, "user": user.tolist(), "item": bu.indx.tolist(),
               "rating": bu.vals.astype(dt).tolist(), "type": typ.tolist()}
        (tmp_path / f"in{world}.json").write_text(json.dumps(inp))
        r = subprocess.run(["node", os.path.join(HERE, "js", "train_gpu_ranks.js"), str(tmp_path / f"in{world}.json")],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[world] = json.loads(r.stdout.strip().splitlines()[-1])
    for world in worlds[1:]:
        for name in ("user_factors", "item_factors"):
            a = np.fromfile(tmp_path / "w1" / "ml_factors_ready" / name, dt)
            b = np.fromfile(tmp_path / f"w{world}" / "ml_factors_ready" / name, dt)
            assert a.size == (bu.rows if name[0] == "u" else bu.cols) * 12 and np.array_equal(a, b), (world, name)
        for h1, hw in zip(res[1]["history"], res[world]["history"]):
            for key in ("rmseValidate", "rmseTest", "rmseTestShifted", "globalAvgShift"):
                assert abs(h1[key] - hw[key]) < 1e-12, (world, key)
        assert res[world]["stepInfo"]["exchangeBytes"] > 0


@pytest.mark.gpu
def test_js_train_falls_back_to_the_other_transport(tmp_path):
    """EmfLord.trainOnGpus with two processes on ONE device asking for 'rccl' (which refuses duplicate devices): the Lord starts
    the processes afresh over 'ipc' and says so in its result; with strictTransport the same train is an error."""
    bu, user, typ, U, V = problem(seed=9, users=120, items=70)
    inp = {"dir": str(tmp_path / "w2"), "k": 12, "iters": 2, "rip": 40, "threads": 2, "useDoublePrecision": False, "world": 2, "transport": "rccl",
           "users": bu.rows, "items": bu.cols, "user": user.tolist(), "item": bu.indx.tolist(),
           "rating": bu.vals.astype(np.float32).tolist(), "type": typ.tolist()}
    (tmp_path / "in.json").write_text(json.dumps(inp))
    r = subprocess.run(["node", os.path.join(HERE, "js", "train_gpu_ranks.js"), str(tmp_path / "in.json")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["commTransport"] == "ipc" and out["commFallback"] and out["commFallback"][0].startswith("rccl:")
    assert len(out["history"]) == 2 and out["stepInfo"]["exchangeBytes"] > 0
    inp["strict"] = True
    inp["dir"] = str(tmp_path / "w2s")
    (tmp_path / "in_strict.json").write_text(json.dumps(inp))
    r = subprocess.run(["node", os.path.join(HERE, "js", "train_gpu_ranks.js"), str(tmp_path / "in_strict.json")], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


@pytest.mark.gpu
def test_node_split_and_stats_on_gpu(tmp_path):
    """N1 through the addon: the same split as the oracle, statistics as numpy."""
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 60, 300)
    rowPtr = np.zeros(301, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    types = rng.choice(np.array([0, 0, 0, 1, 2, 4], np.int8), rowPtr[-1])
    vals = rng.integers(1, 6, rowPtr[-1]).astype(np.float32)
    k, items = 12, 90
    Vf = (rng.standard_normal((items, k)) * 0.9).astype(np.float32)
    uf = (rng.standard_normal(k) * 0.9).astype(np.float32)
    skip1 = [5, 1, 17, 5, 60]   # 1-based, unsorted, one duplicate: the wrapper sorts and de-duplicates
    (tmp_path / "in.json").write_text(json.dumps({"rowPtr": rowPtr.tolist(), "types": types.tolist(), "vals": vals.tolist(),
                                                  "dataSetDistr": [85, 10, 5], "seed": 77,
                                                  "rec": {"user": uf.tolist(), "items": Vf.ravel().tolist(), "k": k, "skip1": skip1,
                                                          "shift": 0.2, "min": 0.5, "limit": 8}}))
    r = subprocess.run(["node", os.path.join(HERE, "js", "prep_gpu.js"), str(tmp_path / "in.json")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    want = orc.split_to_sets(rowPtr, types, (85, 10, 5), 77)
    assert np.array_equal(np.array(out["types"], np.int8), want)
    cnt, sm = orc.rating_stats(rowPtr, vals, want)
    assert np.array_equal(np.array(out["cnt"]), cnt)
    avg = np.divide(sm, cnt, out=np.zeros_like(sm), where=cnt > 0)
    assert np.allclose(out["avg"], avg, rtol=1e-13)
    assert out["max"] == cnt.max() and out["total"] == cnt.sum()
    assert np.isclose(out["totalRatingsAvg"], sm.sum() / cnt.sum(), rtol=1e-13)
    assert out["wrongTypeMessage"] == "invalid type!"
    oid, opr = orc.recommend(uf, Vf, np.array([0, 4, 16, 59], np.int32), 0.2, 0.5, 8)
    assert [r["id"] for r in out["rec"]] == (oid + 1).tolist() and len(oid) <= 7
    assert np.allclose([r["predict"] for r in out["rec"]], opr, atol=1e-5)

