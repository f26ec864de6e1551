"""bench.py prints one JSON line with the contract's fields, for one process and for the two-process code path (both ranks on
GPU 0 over gloo: RGP_BENCH_ONE_GPU=1), at tiny sizes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--T', '128', '--n-total', '8', '--minibatch', '4', '--steps', '3', '--warmup', '1']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_process_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + SMALL, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 3 and j['dtype'] == 'f64' and j['value'] > 0
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(j['roofline'])
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(j['cpu_baseline'])
    assert j['elbo_rel_err_vs_oracle'] < 1e-10


def test_two_process_path_on_one_gpu():
    env = dict(os.environ, RGP_BENCH_ONE_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j['n_gpus'] == 2 and j['scaling'] == 'strong' and j['value'] > 0


def test_wide_window_line():
    """--wide: config 4 read with 16-dimensional hidden layers (windows of 340 / 640 / 320 columns), at a tiny size."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--wide', '--T', '96', '--n-total', '4', '--minibatch', '2',
                        '--steps', '2', '--warmup', '1'], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j['n_gpus'] == 1 and j['value'] > 0 and j['roofline']['bound'] == 'mfma' and j['roofline']['launches'] == 6
    assert 'config4-wide' in j['config']['workload']
    assert j['elbo_rel_err_vs_oracle'] < 1e-10
