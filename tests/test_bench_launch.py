"""bench.py launch contract: `python bench.py --gpus N` without a launcher must start N ranks itself (it used to run ONE rank
and print n_gpus: 1).  CPU: the rendezvous-only --dry-run over gloo; GPU: two ranks sharing cuda:0 over gloo run real steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout, attempts=1):
    env = dict(os.environ, **env_extra)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    for _ in range(attempts):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=env)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]           # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_dry_run():
    out = _run(['--gpus', '2', '--dry-run'], {'DBX_DIST_BACKEND': 'gloo'}, 300)
    assert out['n_gpus'] == 2 and out['rank_sum'] == 3 and out['backend'] == 'gloo'


def test_bench_single_rank_dry_run():
    out = _run(['--gpus', '1', '--dry-run'], {}, 300)
    assert out['n_gpus'] == 1


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_gloo():
    out = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--no-cpu-baseline', '--no-inference'],
               {'DBX_DIST_BACKEND': 'gloo'}, 1200)
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 8 and out['value'] > 0
    assert out['roofline']['kernel'] and 0 < out['roofline']['frac'] < 1
