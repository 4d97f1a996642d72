"""bench.py launch contract: `python bench.py --gpus N` without a launcher must start N ranks itself (it used to run ONE rank
and print n_gpus: 1).  CPU: the rendezvous-only --dry-run over gloo; GPU: two ranks sharing cuda:0 over gloo run real steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DRIVER_FAULT = ('GPU core dump created', 'Memory access fault', 'HSA_STATUS_ERROR', 'hipErrorIllegal')


def _run(args, env_extra, timeout, attempts=1):
    """Runs bench.py; with attempts > 1 (several ranks time-slicing ONE GPU) a run that died in DRIVER code -- a GPU fault signature in
    its stderr, typically in torch's fill kernel while eight contexts initialise -- is retried; any other failure fails at once, and a
    test whose every attempt died that way is reported as xfail (the box cannot time-slice that many contexts), never as a pass."""
    env = dict(os.environ, **env_extra)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    faults = []
    for _ in range(attempts):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=env)
        if r.returncode == 0:
            break
        sig = [m for m in DRIVER_FAULT if m in r.stderr or m in r.stdout]
        if not sig or attempts == 1:
            break
        kern = [l.strip() for l in (r.stderr + r.stdout).splitlines() if 'Kernel Name' in l]
        faults.append('%s %s' % (sig[0], kern[0] if kern else ''))
    if r.returncode != 0 and len(faults) == attempts:
        pytest.xfail('every one of %d attempts died in driver code: %s' % (attempts, '; '.join(faults)))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if faults:
        import warnings
        warnings.warn('bench.py %s needed %d attempts (%s)' % (' '.join(args), len(faults) + 1, '; '.join(faults)))
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


def _run_expect_failure(args, env_extra, timeout):
    import time
    env = dict(os.environ, **env_extra)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=env)
    dt = time.time() - t0
    assert r.returncode != 0, r.stdout[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    errs = [l for l in lines if 'error' in l]
    assert errs, 'no parsable failure line:\n' + r.stdout[-2000:] + r.stderr[-3000:]
    assert not any('value' in l for l in lines)         # no bench line next to a failure line
    return errs[0], dt


def test_bench_failed_collective_setup_reports_one_json_line():
    """The N > 1 path must never hang or die with only a traceback: here the RCCL backend is asked for on a box where it cannot
    come up (no GPU: the process group constructor refuses) -- rank 0 prints {"error": ..., "rccl": {...}} and the job exits non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('CPU-only variant (the GPU box runs test_bench_two_ranks_nccl_on_one_gpu_fails_fast)')
    err, dt = _run_expect_failure(['--gpus', '2', '--dry-run'], {'DBX_DIST_BACKEND': 'nccl', 'DBX_DIST_TIMEOUT_S': '30'}, 300)
    assert err['rccl']['world'] == 2 and err['rccl']['timeout_s'] == 30.0 and 'DBX_DIST_BACKEND' in err['rccl']['env']
    assert dt < 150


@pytest.mark.gpu
def test_bench_two_ranks_nccl_on_one_gpu_fails_fast():
    """Two RCCL ranks on ONE GPU cannot form a communicator (duplicate device): the first collective fails or stalls -- the bounded
    process-group timeout / the watchdog around the first all-reduce turn that into one JSON failure line within the bound, not the
    ten-minute default hang.  (The first real N > 1 RCCL run happens on the driver's multi-GPU box with nobody watching.)"""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip('needs exactly one visible GPU')
    err, dt = _run_expect_failure(['--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '2', '--no-cpu-baseline', '--no-inference'],
                                  {'DBX_DIST_BACKEND': 'nccl', 'DBX_DIST_TIMEOUT_S': '60'}, 400)
    assert err['rccl']['world'] == 2 and err['rccl']['gpus_visible'] == 1
    assert dt < 150, dt
