"""World size 8 (BASELINE configs[3]) on the ONE GPU a test box has: eight ranks over gloo sharing cuda:0.  Functional coverage of the
N = 8 path only (rank slices, the positive-count all-reduce, bucket order, every rank's parameters equal after a step, the bench's
launch contract) -- no scaling number.  This file sorts last on purpose and retries: eight device contexts time-slicing one (virtual)
device died in driver code (a fault in torch's fill kernel at start-up, an illegal-instruction abort mid-run) about once in three runs,
with kernels that pass at world sizes 1 and 2 in the same call."""
import pytest

from test_bench_launch import _run
from test_hip_dist import run_ranks_equal_one_process

pytestmark = pytest.mark.gpu


def test_eight_ranks_equal_one_process(tmp_path):
    run_ranks_equal_one_process(tmp_path, 8, attempts=4)


def test_bench_eight_ranks_on_one_gpu_gloo():
    """configs[3]'s world size on the one GPU a builder box has: 8 ranks over gloo sharing cuda:0, 2 patches each (global batch 16):
    rank slices, the positive-count all-reduce, bucket order and the max-over-ranks timing at world 8.  No scaling number."""
    out = _run(['--gpus', '8', '--steps', '2', '--warmup', '1', '--batch', '2', '--no-cpu-baseline', '--no-inference'],
               {'DBX_DIST_BACKEND': 'gloo'}, 1200, attempts=4)     # (eight contexts on one virtual device: start-up faulted once in three runs)
    assert out['n_gpus'] == 8 and out['config']['global_batch'] == 16 and out['value'] > 0
    assert out['rccl']['world'] == 8 and out['rccl']['collective']
