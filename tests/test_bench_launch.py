"""bench.py's own launcher (CPU side): `python bench.py --gpus N` with N > 1 and no launcher around it must start the per-GPU
processes as a CHILD (`python -m torch.distributed.run ...`), never hang, and hand the child's exit code on -- here, without a
GPU, every rank refuses to compute ("needs a HIP device (no CPU fallback)") and the parent's code is non-zero.  What the ranks
then do on a GPU is tests/test_gpu_parity.py::test_bench_starts_its_own_ranks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_call_with_gpus_starts_ranks_and_relays_their_exit_code():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu suite")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "ml100k",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert "launching 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert r.returncode != 0
    assert "needs a HIP device" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no line without a measurement


def test_a_launched_rank_does_not_launch_again():
    """Inside a launcher's job (WORLD_SIZE set) bench.py must not start another one."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.argv = ['bench.py', '--gpus', '2']; import bench; bench.self_launch(); print('not launched')"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 0 and "not launched" in r.stdout, r.stderr[-500:]
