"""The N>1 path on CPU: two processes over gloo shard the samples of one image exactly as bench.py's
ranks do (pathtracer_amd.distributed), all-reduce the linear accumulation buffers, and finish the film.
The per-rank renderer here is the CPU oracle standing in for the GPU (no GPU exists in this test); what
is under test is the sharding (distributed.shard, both modes, the function bench.py's ranks call), the reduction and
the order 'reduce first, gamma after'.  pt_render_device + RCCL itself runs in the GPU tests (one rank: the box has one
GPU) and in the driver's multi-GPU bench."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import api, distributed, scenes
from tests import util

W, H, SPP, DEPTH = 48, 27, 3, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        osc = util.oracle_scene(scenes.default_scene())
        off, n = distributed.shard(mode, rank, world, SPP)  # what bench.py's ranks call
        img, rays = osc.render(W, H, n, DEPTH, sample0=off, threads=1)
        acc = torch.from_numpy(img.copy())
        distributed.all_reduce_sum(acc)
        r = torch.tensor([rays], dtype=torch.int64)
        dist.all_reduce(r)
        if rank == 0:
            np.save(os.path.join(out_dir, "acc_%s.npy" % mode), acc.numpy())
            np.save(os.path.join(out_dir, "rays_%s.npy" % mode), r.numpy())
    finally:
        dist.destroy_process_group()


def _run(mode, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    return np.load(tmp_path / ("acc_%s.npy" % mode)), int(np.load(tmp_path / ("rays_%s.npy" % mode))[0])


def test_weak_sharding_two_ranks(tmp_path):
    acc, rays = _run("weak", tmp_path)
    osc = util.oracle_scene(scenes.default_scene())
    full, full_rays = osc.render(W, H, 2 * SPP, DEPTH)  # the image has world * spp samples
    assert rays == full_rays
    np.testing.assert_allclose(acc, full, rtol=1e-5, atol=1e-6)  # only the fp32 sum association differs
    g_sharded, q_sharded = api.film_finalize(acc, 2 * SPP)  # gamma AFTER the reduction (main.cu:95-98 is non-linear)
    g_full, q_full = api.film_finalize(full, 2 * SPP)
    assert np.abs(q_sharded.astype(int) - q_full.astype(int)).max() <= 1


def test_strong_sharding_two_ranks(tmp_path):
    acc, rays = _run("strong", tmp_path)
    osc = util.oracle_scene(scenes.default_scene())
    full, full_rays = osc.render(W, H, SPP, DEPTH)
    assert rays == full_rays
    np.testing.assert_allclose(acc, full, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("spp", [8, 64, 100, 1024, 4096])
def test_shards_tile_the_sample_range(world, spp):
    """Strong: the ranks' ranges are contiguous, disjoint, cover [0, spp) and differ by at most one sample (config 3:
    1024 spp over 1/2/4/8 GPUs, config 5: 4096 over 8).  Weak: rank r starts where rank r-1 ends and every rank has spp."""
    at, sizes = 0, []
    for r in range(world):
        off, n = distributed.shard("strong", r, world, spp)
        assert off == at and n >= 1
        at += n
        sizes.append(n)
    assert at == spp and max(sizes) - min(sizes) <= 1
    for r in range(world):
        assert distributed.shard("weak", r, world, spp) == (r * spp, spp)
    with pytest.raises(ValueError):
        distributed.shard("strong", 0, spp + 1, spp)  # a rank without a sample
    with pytest.raises(ValueError):
        distributed.shard("strong", world, world, spp)
    with pytest.raises(ValueError):
        distributed.shard("neither", 0, world, spp)


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("spp,offset", [(8, 0), (64, 0), (100, 7), (1024, 0), (4096, 1 << 20), (4095, 3)])
def test_the_library_group_cuts_samples_like_the_ranks_do(world, spp, offset):
    """pt_group_render's own split (pt_group_shard, the C++ arithmetic its GPUs get their ranges from) tiles
    [offset, offset + spp) and equals distributed.shard("strong"), the function bench.py's ranks call: a wrong range
    there would duplicate or skip samples without any GPU test noticing (one GPU is all they have)."""
    at = offset
    for g in range(world):
        off, n = api.group_shard(spp, offset, world, g)
        want_off, want_n = distributed.shard("strong", g, world, spp)
        assert (off, n) == (offset + want_off, want_n)
        assert off == at and n >= 1
        at += n
    assert at == offset + spp
    for bad in ((spp, offset, world, world), (spp, offset, world, -1), (spp, offset, 0, 0), (world - 1, offset, world, 0)):
        with pytest.raises(api.PtError):
            api.group_shard(*bad)
