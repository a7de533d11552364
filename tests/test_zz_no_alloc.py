"""pt_render_device_noalloc against pt_render_device (GPU).

The entry point for callers that cannot allocate renders with the library's fixed per-GPU scratch array instead of
a workspace grown on demand.  Same paths, same arithmetic, same film order: the linear sums must equal
pt_render_device's bit for bit (the bar of every other pipeline-equivalence test here), on two calls in a row, the
second with a smaller input (the array keeps the first call's contents), and beyond one pass of the array.
"""
import numpy as np
import pytest
import torch

from pathtracer_amd import api, scenes

pytestmark = pytest.mark.gpu


def _both(ctx, w, h, spp, depth, **kw):
    p = api.Context.params(w, h, spp, depth, **kw)
    ref = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    got = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    st = ctx.render_device(p, ref.data_ptr(), stream, want_stats=True)
    ctx.render_device_noalloc(p, got.data_ptr(), stream)
    torch.cuda.synchronize()
    return got.cpu().numpy(), ref.cpu().numpy(), st


@pytest.mark.parametrize("name", ["random_spheres", "mixed_1k"])
def test_noalloc_equals_render_device_twice_in_a_row(ctx, name):
    b = scenes.random_spheres_scene() if name == "random_spheres" else scenes.mixed_scene(n=1000, seed=3, extent=20.0, n_materials=64)
    ctx.upload(b.desc())
    for (w, h, spp, depth) in ((320, 180, 6, 50), (96, 54, 2, 12)):  # the second call is the smaller one
        got, ref, st = _both(ctx, w, h, spp, depth)
        assert st["rays"] > w * h * spp
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (name, w, h, spp)


def test_noalloc_frame_larger_than_the_scratch_array(ctx):
    """1280x720 at 8 spp is 7.4M paths, 2.6 passes of the 32 MiB array (2.8M paths): bands of pixels, same image."""
    ctx.upload(scenes.random_spheres_scene().desc())
    assert 1280 * 720 * 8 * 12 > api.NOALLOC_SCRATCH_BYTES
    got, ref, _ = _both(ctx, 1280, 720, 8, 50, sample_offset=3)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_noalloc_limits_and_arguments(ctx):
    ctx.upload(scenes.default_scene().desc())
    buf = torch.zeros((8, 8, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(api.PtError) as e:  # more than 2^32 paths
        ctx.render_device_noalloc(api.Context.params(16384, 16384, 17, 5), buf.data_ptr(), 0)
    assert e.value.status == 4
    with pytest.raises(api.PtError) as e:  # a tile of that many samples does not fit the array
        ctx.render_device_noalloc(api.Context.params(8, 8, 100000, 5, samples_per_pass=100000), buf.data_ptr(), 0)
    assert e.value.status == 4
    with pytest.raises(api.PtError) as e:  # flags that need events or another pipeline
        ctx.render_device_noalloc(api.Context.params(8, 8, 1, 5, flags=api.RENDER_STAGE_TIMES), buf.data_ptr(), 0)
    assert e.value.status == 1


def test_noalloc_is_capturable_into_a_graph(ctx):
    """Stream capture refuses allocation, synchronisation and event queries: a capture that succeeds, runs nothing while
    it records and reproduces pt_render_device's sums on every replay shows the entry point does none of them."""
    ctx.upload(scenes.random_spheres_scene().desc())
    w, h, spp, depth = 160, 90, 4, 20
    p = api.Context.params(w, h, spp, depth)
    ref = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    ctx.render_device(p, ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    got = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g.capture_begin()
        ctx.render_device_noalloc(p, got.data_ptr(), s.cuda_stream)
        g.capture_end()
    torch.cuda.synchronize()
    assert float(got.abs().max()) == 0.0  # recorded, not run
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32))
    g.replay()  # the film accumulates: a second replay adds the same samples again, as a second pt_render_device call does
    ctx.render_device(p, ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32))
