"""The multi-GPU path on CPU: two gloo ranks shard the samples by global index, reduce the double
accumulators to rank 0, and rank 0's sums equal the single-process render of all samples.
The renderer inside each rank is the CPU oracle (the HIP kernels need a GPU); what is under test is
the sharding + reduction logic bench.py uses (monte-carlo-ray-tracer_amd/sharding.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import mcrt_pkg
    import pyoracle as po
    from mcrt_amd import sharding

    pkg = mcrt_pkg.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = po.Oracle()
    flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
    cam = orc.camera(24, 16, 90.0, (0, 0, 5), (0, 0, 0))
    begin, count = sharding.rank_sample_range(rank, world, 6)
    sums, _ = orc.render(orc.scene(flat), cam, begin, count, 100, 0x5EED0001, threads=2, portable_math=True, forward_sum=True)
    acc = torch.from_numpy(sums.reshape(-1).copy())
    sharding.reduce_accumulator(acc, dist, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_render_equals_single_process(tmp_path, oracle, pkg, scene_of):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out).reshape(16, 24, 3)
    cam = oracle.camera(24, 16, 90.0, (0, 0, 5), (0, 0, 0))
    sc = oracle.scene(scene_of("cornell"))
    a, _ = oracle.render(sc, cam, 0, 6, 100, 0x5EED0001, threads=2, portable_math=True, forward_sum=True)
    b, _ = oracle.render(sc, cam, 6, 6, 100, 0x5EED0001, threads=2, portable_math=True, forward_sum=True)
    assert np.array_equal(got, a + b)
    whole, _ = oracle.render(sc, cam, 0, 12, 100, 0x5EED0001, threads=2, portable_math=True, forward_sum=True)
    assert np.allclose(got, whole, rtol=1e-12, atol=1e-12)      # independent of N up to summation order


def test_rank_ranges_partition_the_samples(pkg):
    from mcrt_amd import sharding

    for world in (1, 2, 4, 8):
        covered = []
        for r in range(world):
            b, n = sharding.rank_sample_range(r, world, 256)
            covered += list(range(b, b + n))
        assert covered == list(range(256 * world))
        assert sharding.total_spp(world, 256) == 256 * world
    with pytest.raises(ValueError):
        sharding.rank_sample_range(2, 2, 4)
