"""world_size-2 run of the sharded train loop over gloo on CPU: both ranks must end with the
same replicated factor matrices as a single-process run, bit for bit.  The oracle stands in
for the GPU backend (test only); what is under test is the product's host logic: shard
ranges, the padded all-gather of solved shards, and the RMSE all-reduce."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[3])
import torch, torch.distributed as dist
from helpers import OracleBackend, make_problem
from test_host import small_dataset
from ycnr_als.emf import EmfLord
world = int(os.environ.get("WORLD_SIZE", "1"))
d = None
if world > 1:
    dist.init_process_group("gloo")
    d = dist
ds, U, V = small_dataset(seed=5, users=61, items=44, dt=np.float64)
lord = EmfLord(options={"factorsCount": 8, "trainIters": 3, "useDoublePrecision": True, "dataDir": sys.argv[4],
                        "ratingsInPortionForRmse": 30}, backend_factory=lambda o, u, i, dv: OracleBackend(o, u, i, dv), dist=d)
lord.prepareToTrain(ds, U.astype(np.float64), V.astype(np.float64))
hist = lord.train()
rank = lord.rank
np.savez(os.path.join(sys.argv[4], f"out_w{world}_r{rank}.npz"), U=lord.backend.get_factors(0), V=lord.backend.get_factors(1),
         rmse=np.array([[h["rmseValidate"], h["rmseTest"], h["rmseTestShifted"], h["globalAvgShift"]] for h in hist]),
         shards=np.concatenate([lord.shards[0], lord.shards[1]]))
if d: dist.destroy_process_group()
'''


def run(world, tmp):
    script = os.path.join(tmp, "worker.py")
    open(script, "w").write(WORKER)
    root = os.path.dirname(HERE)
    args = [script, HERE, os.path.join(root, "you-can-not-recommend_amd", "python"), root, tmp]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    if world == 1:
        subprocess.check_call([sys.executable] + args, env=env, timeout=300)
    else:
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                               "--master-addr", "127.0.0.1", "--master-port", "29531"] + args, env=env, timeout=600)


def test_two_ranks_equal_one_rank():
    with tempfile.TemporaryDirectory() as tmp:
        run(1, tmp)
        run(2, tmp)
        one = np.load(os.path.join(tmp, "out_w1_r0.npz"))
        for r in (0, 1):
            two = np.load(os.path.join(tmp, f"out_w2_r{r}.npz"))
            assert np.array_equal(two["U"], one["U"]) and np.array_equal(two["V"], one["V"])
            assert np.allclose(two["rmse"], one["rmse"], rtol=1e-12, atol=1e-12)
        s = np.load(os.path.join(tmp, "out_w2_r0.npz"))["shards"]
        assert s[0] == 0 and s[2] == 61 and 0 < s[1] < 61  # two non-empty user shards
        assert s[3] == 0 and s[5] == 44 and 0 < s[4] < 44
