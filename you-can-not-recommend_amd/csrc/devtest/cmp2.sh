set -e
C="--workload ml100k --steps 1 --warmup 0 --no-cpu-baseline"
python bench.py $C --dump-factors /tmp/one.npz > /dev/null 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --backend gloo --same-device --dump-factors /tmp/two.npz $C > /tmp/two.log 2>&1 || tail -5 /tmp/two.log
python - <<'PY'
import numpy as np
a=np.load('/tmp/one.npz'); b=np.load('/tmp/two.npz')
for n in 'UV':
    d=np.abs(a[n]-b[n]).max(1); bad=np.nonzero(d>0)[0]
    print(n, 'rows differing', len(bad), 'of', len(d), 'first', bad[:10], 'last', bad[-5:], 'max', d.max())
PY
