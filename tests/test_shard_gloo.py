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


def test_bench_plumbing_failures_name_the_rank():
    """A failure of the process-group plumbing (RCCL init, barrier, max-reduce) exits non-zero with one JSON line on
    stderr that names the rank and the stage: a red SCALE record must be diagnosable without a rerun."""
    import json
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "try:\n    raise RuntimeError('no xGMI peer')\n"
            "except Exception as e:\n    bench.die(5, 'process group init (nccl)', e)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    first = json.loads(r.stderr.splitlines()[0])
    assert first == {"bench_error": "process group init (nccl)", "rank": 5, "error": "RuntimeError: no xGMI peer"}


def test_flop_accounting():
    sys.path.insert(0, ROOT)
    import bench
    # SURVEY.md section 8d / BASELINE.md section 3
    assert abs(bench.algorithmic_flops("3x3", 128, 256, 256) - 29.59e9) < 0.01e9
    assert abs(bench.algorithmic_flops("3x3", 128, 128, 128) - 7.399e9) < 0.001e9
    assert abs(bench.algorithmic_flops("1x1", 128, 1024, 256) - 13.15e9) < 0.01e9
    assert abs(bench.executed_mfma_flops("3x3", 128, 256, 256) - 13.15e9) < 0.01e9


def test_bench_shard_plans(pkg):
    """bench.py --scaling weak / strong: the per-rank batches and the job's FLOP accounting.  cfg 5 of
    BASELINE.json (block, N = 1024 over 8 GPUs) is 128 images per rank either way."""
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.plan_shard("weak", 128, r, 8) for r in range(8)] == [(128, 1024)] * 8
    assert [bench.plan_shard("strong", 1024, r, 8) for r in range(8)] == [(128, 1024)] * 8
    # uneven strong split: contiguous, covers the batch once, equals the package's shard_range
    for batch, world in ((130, 4), (7, 8), (128, 3)):
        got = [bench.plan_shard("strong", batch, r, world) for r in range(world)]
        assert sum(n for n, _ in got) == batch and all(g == batch for _, g in got)
        for r in range(world):
            a, b = pkg.shard_range(batch, r, world)
            assert got[r][0] == b - a
    # the whole job's FLOPs are the sum of the shards' (the kernel is linear in N)
    kind, C, K, _ = bench.LAYERS["residual_block"]
    per = sum(bench.algorithmic_flops(kind, bench.plan_shard("strong", 1024, r, 8)[0], C, K) for r in range(8))
    assert abs(per - bench.algorithmic_flops(kind, 1024, C, K)) < 1.0
    assert abs(bench.algorithmic_flops(kind, 1024, C, K) - 447.2e9) < 0.1e9


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("ranks,scaling,extra", [(2, "weak", ["--layer", "conv3x3_256"]),
                                                 (2, "strong", ["--layer", "residual_block", "--batch", "64"]),
                                                 (4, "strong", ["--layer", "residual_block", "--batch", "512"])])
def test_bench_ranks_on_one_gpu(ranks, scaling, extra, tmp_path):
    """The multi-rank path of bench.py itself (process group, barrier, max over ranks, strong / weak
    shards) rehearsed with several ranks sharing the one visible GPU over gloo -- the command the driver
    launches for N GPUs, minus RCCL (which needs one device per rank).  The 4-rank case is BASELINE
    configs[4]'s per-GPU share (128 images of the bottleneck block per rank, a strong split of 512); the GPU
    boxes admit at most 6 processes on the card, so the 8-rank job itself cannot be rehearsed on one GPU."""
    import json
    import subprocess
    env = dict(os.environ, WINO_BENCH_BACKEND="gloo")
    port = 29700 + (os.getpid() + 7 * ranks) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--steps", "3" if ranks > 2 else "5", "--warmup", "2", "--preheat-ms", "20", "--scaling", scaling,
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 alone prints
    js = json.loads(lines[0])
    assert js["n_gpus"] == ranks and js["scaling"] == scaling and js["value"] > 0
    if ranks == 4:
        assert js["config"]["global_batch"] == 512 and js["config"]["per_gpu_batch"] == 128
    elif scaling == "strong":
        assert js["config"]["global_batch"] == 64 and js["config"]["per_gpu_batch"] == 32
    else:
        assert js["config"]["global_batch"] == 256 and js["config"]["per_gpu_batch"] == 128
