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


def _progressive_worker(rank, world, port, out_path, scaling):
    """bench.py's step(): passes, one reduce per pass, non-root ranks restart from zero, rank 0 keeps the running image."""
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
    sc = orc.scene(pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz")))
    cam = orc.camera(16, 12, 90.0, (0, 0, 5), (0, 0, 0))
    plan = sharding.SamplePlan(rank, world, 7, scaling, pass_spp=3)
    acc = torch.zeros(12 * 16 * 3, dtype=torch.float64)
    seen = []
    for k, (first, count) in enumerate(plan.passes):
        if count:
            sums, _ = orc.render(orc.scene(sc.keep), cam, first, count, 100, 0x5EED0001, threads=1, portable_math=True, forward_sum=True,
                                 sums=acc.numpy().reshape(12, 16, 3).copy())
            acc = torch.from_numpy(sums.reshape(-1).copy())
        sharding.reduce_accumulator(acc, dist, dst=0)
        if k + 1 < len(plan.passes):
            sharding.after_reduce(acc, dist, dst=0)
        seen.append(plan.spp_after(k + 1))
    dist.barrier()
    if rank == 0:
        np.savez(out_path, acc=acc.numpy(), seen=np.array(seen), total=plan.total_spp)
    dist.destroy_process_group()


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_progressive_passes_over_two_ranks(tmp_path, oracle, scene_of, scaling):
    """Strong scaling (7 spp in total, 3-spp pass windows split over 2 ranks, one rank idle in the last window) and weak
    scaling (7 spp per rank): rank 0 ends with the sum over every global sample index exactly once."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "acc.npz")
    mp.spawn(_progressive_worker, args=(2, port, out, scaling), nprocs=2, join=True)
    z = np.load(out)
    total = int(z["total"])
    assert total == (7 if scaling == "strong" else 14) and z["seen"][-1] == total
    assert list(z["seen"]) == ([3, 6, 7] if scaling == "strong" else [6, 12, 14])
    cam = oracle.camera(16, 12, 90.0, (0, 0, 5), (0, 0, 0))
    whole, _ = oracle.render(oracle.scene(scene_of("cornell")), cam, 0, total, 100, 0x5EED0001, threads=2, portable_math=True, forward_sum=True)
    assert np.allclose(z["acc"].reshape(12, 16, 3), whole, rtol=1e-12, atol=1e-12)


def test_sample_plans_cover_every_index_once(pkg):
    from mcrt_amd import sharding

    for world in (1, 2, 3, 8):
        for scaling, pass_spp in (("weak", None), ("strong", None), ("strong", 64), ("weak", 5), ("strong", 5)):
            plans = [sharding.SamplePlan(r, world, 100, scaling, pass_spp) for r in range(world)]
            assert len({len(p.passes) for p in plans}) == 1                  # every pass ends in a collective
            covered = sorted(i for p in plans for b, n in p.passes for i in range(b, b + n))
            total = plans[0].total_spp
            assert covered == list(range(total)) and total == (100 * world if scaling == "weak" else 100)
            assert plans[0].spp_after(len(plans[0].passes)) == total and plans[0].spp_after(0) == 0
    assert sharding.split_range(10, 7, 3) == [(10, 3), (13, 2), (15, 2)]
    with pytest.raises(ValueError):
        sharding.SamplePlan(2, 2, 4)


def test_bench_refuses_a_world_that_is_not_gpus():
    """bench.py --gpus N under a launcher that started another number of ranks exits non-zero before touching a GPU."""
    import subprocess

    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


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


def test_baseline_plans_for_eight_gpus(pkg):
    """The sample plans bench.py builds for BASELINE configs 3 and 5 on 8 GPUs (strong scaling) and for config 4 (weak): same number of
    passes on every rank (each ends in the one reduce), every global sample index rendered once, rank 0's running count per pass."""
    import importlib.util

    from mcrt_amd import sharding

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for workload, scaling, want_passes, want_total in (("config3_1024_1024", "strong", 1, 1024), ("cornell_4k_4096", "strong", 64, 4096),
                                                       ("config4_2048_256", "weak", 1, 2048), ("config2_1024_256", "weak", 1, 2048)):
        spp = bench.WORKLOADS[workload][3]
        plans = [sharding.SamplePlan(r, 8, spp, scaling, pass_spp=bench.PASS_SPP.get(workload)) for r in range(8)]
        assert all(len(p.passes) == want_passes for p in plans) and plans[0].total_spp == want_total
        covered = np.zeros(want_total, np.int32)
        for p in plans:
            for b, n in p.passes:
                covered[b:b + n] += 1
        assert (covered == 1).all()
        if workload == "cornell_4k_4096":
            assert all(n == 8 for p in plans for _, n in p.passes)              # a 64-spp window over 8 ranks
            assert [plans[0].spp_after(k) for k in (1, 2, 64)] == [64, 128, 4096]
        if workload == "config3_1024_1024":
            assert [p.passes[0] for p in plans] == [(128 * r, 128) for r in range(8)]
