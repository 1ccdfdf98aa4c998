"""N > 1 path on CPU: two processes over `gloo` exercise bench.py's distributed plumbing
(process group, barrier, max-over-ranks timing) and the batch split.  The data path has no
collective: each rank convolves its own contiguous image range (here with the CPU oracle
standing in for the HIP kernel, which tests may do) and the concatenation of the shards must
equal the whole-batch result."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    import __graft_entry__ as ge
    from oracle import oracle as O
    pkg = ge.load_package()
    r, lr, w = bench.dist_init("gloo")
    assert (r, w) == (rank, world)
    rng = np.random.RandomState(0)            # same data on every rank
    N, C, K = 5, 8, 64
    x = (rng.rand(N, 16, 16, C) - 0.5).astype(np.float32)
    wgt = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    s = (rng.rand(K) - 0.5).astype(np.float32)
    b = (rng.rand(K) - 0.5).astype(np.float32)
    n0, n1 = pkg.shard_range(N, rank, world)
    calls = {"n": 0}

    def step():
        calls["n"] += 1
        return O.conv3x3_bn_relu_direct(x[n0:n1], wgt, s, b)

    import time
    secs = bench.timed_steps(step, 3, 1, lambda: None, dist.barrier)
    assert calls["n"] == 4
    slow = secs + (0.25 if rank == 1 else 0.0)   # pretend rank 1 was slower
    mx = bench.max_over_ranks(slow, world)
    assert mx >= slow - 1e-9 and (rank == 1 or mx > secs + 0.2)
    np.save(os.path.join(tmp, f"shard{rank}.npy"), step())
    dist.barrier()
    if rank == 0:
        whole = O.conv3x3_bn_relu_direct(x, wgt, s, b)
        parts = np.concatenate([np.load(os.path.join(tmp, f"shard{i}.npy")) for i in range(world)])
        # BLAS may block a 2-image and a 5-image GEMM differently: equal to fp64 rounding, not bitwise
        assert parts.shape == whole.shape and np.abs(parts - whole).max() < 1e-12
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_batch_split_two_ranks_gloo(tmp_path):
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_flop_accounting():
    sys.path.insert(0, ROOT)
    import bench
    # SURVEY.md section 8d / BASELINE.md section 3
    assert abs(bench.algorithmic_flops("3x3", 128, 256, 256) - 29.59e9) < 0.01e9
    assert abs(bench.algorithmic_flops("3x3", 128, 128, 128) - 7.399e9) < 0.001e9
    assert abs(bench.algorithmic_flops("1x1", 128, 1024, 256) - 13.15e9) < 0.01e9
    assert abs(bench.executed_mfma_flops("3x3", 128, 256, 256) - 13.15e9) < 0.01e9
