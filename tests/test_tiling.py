"""Tile-parallel driver (resselt_amd/tiling.py): planning logic, single-process tiling and the world_size-2 path over
gloo on CPU with the oracle as the per-tile model (the GPU engine is not involved here)."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from resselt_amd.tiling import TileParallel, choose_grid, plan_tiles, upscale_tiled
from resselt_amd.utils import synth


def _model():
    from oracle.rrdbnet import rrdbnet_forward

    sd = synth.rrdbnet_state_dict(nb=1, nf=16, scale=2, seed=4)

    def run(x):
        with torch.no_grad():
            return rrdbnet_forward(sd, x)

    return run


def test_grid_and_plan_cover_the_image_exactly():
    assert choose_grid(8, 4320, 7680) == (2, 4) and choose_grid(4, 1080, 1920) == (2, 2) and choose_grid(1, 10, 10) == (1, 1)
    assert choose_grid(2, 1080, 1920) == (1, 2) and choose_grid(3, 100, 100) in ((1, 3), (3, 1))
    for (h, w, r, c, halo, align) in [(4320, 7680, 2, 4, 32, 1), (50, 70, 3, 2, 8, 8), (17, 5, 4, 1, 3, 1), (64, 64, 2, 2, 12, 8)]:
        tiles = plan_tiles(h, w, r, c, halo, align)
        cover = torch.zeros(h, w, dtype=torch.int32)
        for t in tiles:
            cover[t.y0 : t.y1, t.x0 : t.x1] += 1
            assert t.ry0 <= t.y0 <= t.y1 <= t.ry1 <= h and t.rx0 <= t.x0 <= t.x1 <= t.rx1 <= w
            assert t.ry0 in (0, max(t.y0 - (halo if halo % align == 0 else (halo // align + 1) * align), 0))
            if align > 1:  # interior edges and read windows stay on the alignment grid
                assert t.y0 % align == 0 and t.x0 % align == 0 and t.ry0 % align == 0 and t.rx0 % align == 0
        assert int(cover.min()) == 1 and int(cover.max()) == 1
    with pytest.raises(ValueError):
        plan_tiles(10, 10, 0, 1)


def test_single_process_tiling_matches_full_frame():
    model = _model()
    x = synth.synth_input((1, 3, 45, 61), seed=8)
    full = model(x)
    # halo 24 exceeds the receptive-field radius of this 1-block net (18 LR px): tiles agree to fp32 rounding
    tiled = upscale_tiled(model, x, scale=2, tile=(20, 32), halo=24)
    assert tiled.shape == full.shape and (tiled - full).abs().max().item() <= 1e-6
    # without halo the seams differ (this is what the halo is for)
    assert (upscale_tiled(model, x, scale=2, tile=(20, 32), halo=0) - full).abs().max().item() > 1e-4
    # world_size 1 TileParallel is the same computation
    assert torch.equal(TileParallel(model, 2, halo=24, grid=(2, 2))(x), upscale_tiled(model, x, scale=2, tile=(23, 31), halo=24))


def _u8_model():
    """uint8 [N, H, W, C] in, uint8 [N, H*2, W*2, C] out (what an engine model with supports_u8 does), on the oracle."""
    run = _model()

    def f(img):
        y = run(img.permute(0, 3, 1, 2).float() / 255)
        return (y.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()

    return f


def _u8_input():
    g = torch.Generator().manual_seed(11)
    return torch.randint(0, 256, (1, 48, 40, 3), generator=g, dtype=torch.uint8)


def test_upscale_tiled_takes_u8_images():
    """An 8-bit [N, H, W, C] image through the single-device tiler: tiles are planned over (H, W), not over (W, C), and the result
    equals the whole-image run wherever the halo covers the receptive field (advisor finding, round 3)."""
    from resselt_amd.tiling import run_tile, Tile

    model, img = _u8_model(), _u8_input()
    full = model(img)
    tiled = upscale_tiled(model, img, scale=2, tile=(20, 16), halo=24)
    assert tiled.shape == full.shape == (1, 96, 80, 3) and tiled.dtype == torch.uint8
    assert (tiled.int() - full.int()).abs().max().item() <= 1  # rounding of values that differ by 1e-6 at a .5 boundary
    assert torch.equal(TileParallel(model, 2, halo=24, grid=(3, 3))(img), upscale_tiled(model, img, scale=2, tile=(16, 14), halo=24))
    # an empty tile keeps the rank and the channel axis of its layout
    empty = Tile(0, 5, 5, 0, 0, 5, 5, 0, 0)
    assert run_tile(model, img, empty, 2).shape == (1, 0, 0, 3)
    assert run_tile(_model(), synth.synth_input((1, 3, 8, 8), seed=1), empty, 2).shape == (1, 3, 0, 0)
    with pytest.raises(ValueError):
        upscale_tiled(model, img[0], scale=2, tile=(20, 16))


def _u8_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        model, img = _u8_model(), _u8_input()
        out = {}
        for name, kw in (('strips', dict(grid=(world, 1))), ('strips2', dict(grid=(2 * world, 1))), ('grid', dict(grid=(world, 2))),
                         ('ragged', dict(grid=(3, 1))), ('sync', dict(grid=(world, 1), overlap=False))):  # fmt: skip
            tp = TileParallel(model, scale=2, halo=24, **kw)
            y = tp(img)
            out[name] = (y.numpy(), dict(tp.last_stats))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_tile_parallel_u8_images_async_and_in_place(world):
    """8-bit images through the tile-parallel driver over gloo: one asynchronous all-gather per round of tiles; equal full-width row bands
    land in the result itself (no receive buffer, no copy); every rank ends with an image bit-identical to the single-process tiling."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_u8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model, img = _u8_model(), _u8_input()
    for name, grid in (('strips', (world, 1)), ('strips2', (2 * world, 1)), ('grid', (world, 2)), ('ragged', (3, 1)), ('sync', (world, 1))):
        single = TileParallel(model, scale=2, halo=24, grid=grid)(img)  # no process group here: the world-size-1 path, same tile plan
        assert single.dtype == torch.uint8 and tuple(single.shape) == (1, 96, 80, 3)
        for rank in range(world):
            y, stats = results[rank][name]
            assert torch.equal(torch.from_numpy(y), single), (name, rank)
            assert stats['tile_dtype'] == 'torch.uint8'
            assert stats['in_place'] == (name in ('strips', 'strips2', 'sync')), (name, stats)
            assert stats['rounds'] == {'strips': 1, 'strips2': 2, 'grid': 2, 'ragged': -(-3 // world), 'sync': 1}[name]
            assert stats['overlap'] == (name != 'sync')
    # uint8 tiles are a quarter of the bytes of the same image as float32 tensors
    assert results[0]['strips'][1]['bytes_per_round'] == 96 * 80 * 3


def _bench_like_cases(model, u8_model):
    """The tile-parallel calls `bench.py --gpus 8` makes, at toy size (same structure: grids, halo clamping, sub-tiling, dtypes):
      c2      8 x 1 full-width row bands of one image, fp32 tensors (padded receive buffer) and 8-bit images (bands gathered in place);
      c5      ONE image, 2 x 4 tiles dealt round-robin, every tile run as sub-tiles by an inner single-device tiler (fp32 and 8-bit);
      uneven  3 x 3 = 9 tiles on 8 ranks (a last round with one tile), idle: 2 x 2 = 4 tiles on 8 ranks (four ranks only join the collective)."""
    from resselt_amd.tiling import upscale_tiled as ut

    g = torch.Generator().manual_seed(21)
    x_c2 = synth.synth_input((1, 3, 8 * 10, 18), seed=21)
    img_c2 = torch.randint(0, 256, (1, 8 * 10, 18, 3), generator=g, dtype=torch.uint8)
    x_c5 = synth.synth_input((1, 3, 32, 64), seed=22)
    img_c5 = torch.randint(0, 256, (1, 32, 64, 3), generator=g, dtype=torch.uint8)
    inner = lambda crop: ut(model, crop, 2, tile=(8 + 16, 16 + 16), halo=8)  # noqa: E731
    inner_u8 = lambda crop: ut(u8_model, crop, 2, tile=(8 + 16, 16 + 16), halo=8)  # noqa: E731
    return {
        'c2_f32': (model, x_c2, dict(grid=(8, 1), halo=8)),
        'c2_u8': (u8_model, img_c2, dict(grid=(8, 1), halo=8)),
        'c5_f32': (inner, x_c5, dict(grid=(2, 4), halo=8)),
        'c5_u8': (inner_u8, img_c5, dict(grid=(2, 4), halo=8)),
        'uneven': (model, x_c5, dict(grid=(3, 3), halo=8)),
        'idle': (u8_model, img_c5, dict(grid=(2, 2), halo=8)),
    }


def _bench_like_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        out = {}
        for name, (fn, x, kw) in _bench_like_cases(_model(), _u8_model()).items():
            tp = TileParallel(fn, scale=2, **kw)
            y = tp(x)
            out[name] = (y.numpy(), dict(tp.last_stats))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_tile_parallel_eight_ranks_as_bench_runs_them():
    """World size 8 over gloo on the CPU: exactly what `bench.py --config c2 --gpus 8` and `--config c5 --gpus 8` execute (no 8-GPU node has
    run them yet), plus an uneven last round and idle ranks; every rank must end with the single-process result, bit for bit."""
    world = 8
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_like_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect_stats = {
        'c2_f32': dict(rounds=1, in_place=False, tile_dtype='torch.float32'),
        'c2_u8': dict(rounds=1, in_place=True, tile_dtype='torch.uint8'),
        'c5_f32': dict(rounds=1, in_place=False, tile_dtype='torch.float32'),
        'c5_u8': dict(rounds=1, in_place=False, tile_dtype='torch.uint8'),
        'uneven': dict(rounds=2, in_place=False, tile_dtype='torch.float32'),
        'idle': dict(rounds=1, in_place=False, tile_dtype='torch.uint8'),
    }
    for name, (fn, x, kw) in _bench_like_cases(_model(), _u8_model()).items():
        single = TileParallel(fn, scale=2, **kw)(x)  # no process group: the world-size-1 path over the same tile plan
        for rank in range(world):
            y, stats = results[rank][name]
            assert torch.equal(torch.from_numpy(y), single), (name, rank)
            for k, v in expect_stats[name].items():
                assert stats[k] == v, (name, rank, k, stats)
            assert stats['overlap'] is True
    # the bytes that cross the links per round: an 8-bit band is a quarter of the same band in fp32
    assert results[0]['c2_f32'][1]['bytes_per_round'] == 4 * results[0]['c2_u8'][1]['bytes_per_round']


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        model = _model()
        x = synth.synth_input((1, 3, 40, 52), seed=9)  # every rank builds the same input
        y = TileParallel(model, scale=2, halo=24)(x)
        y3 = TileParallel(model, scale=2, halo=24, grid=(3, 1))(x)  # 3 uneven tiles on 2 ranks: round-robin + padding
        y1 = TileParallel(model, scale=2, halo=24, grid=(1, 1))(x)  # fewer tiles than ranks: rank 1 idles but joins the collective
        # by value (numpy): a torch tensor goes through the queue as a file descriptor its producer must outlive
        q.put((rank, y.numpy(), y3.numpy(), y1.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_tile_parallel_all_gather_gloo(world):
    """world_size 2 and 4 over gloo: one tile per rank (N x 1 and the chooser's grid), uneven tiles, fewer tiles than ranks."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict()
    for _ in range(world):
        rank, y, y3, y1 = q.get(timeout=180)
        results[rank] = (torch.from_numpy(y), torch.from_numpy(y3), torch.from_numpy(y1))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _model()
    full = model(synth.synth_input((1, 3, 40, 52), seed=9))
    for rank in range(world):
        y, y3, y1 = results[rank]
        assert y.shape == full.shape
        assert (y - full).abs().max().item() <= 1e-6, rank  # every rank ends with the whole image
        assert (y3 - full).abs().max().item() <= 1e-6, rank
        assert (y1 - full).abs().max().item() <= 1e-6, rank


def test_bench_launcher_starts_ranks_and_relays_one_line():
    """`python bench.py --gpus N` without torch.distributed.run starts the N ranks itself (before touching the GPU) and prints
    exactly rank 0's JSON line; a failing child gives a non-zero exit status.  --dry-run keeps the ranks off the GPU."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    ok = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=env, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [ln for ln in ok.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['dry_run'] is True
    # --gpus that does not divide the 8 tiles of config c5: every rank exits non-zero, and so does the launcher
    bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '3', '--config', 'c5', '--dry-run'], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith('{')]


def test_splits_never_yield_empty_tiles():
    from resselt_amd.tiling import _splits

    assert _splits(10, 4, 8) == [0, 8, 10]
    assert _splits(5, 3, 8) == [0, 5]
    tiles = plan_tiles(10, 20, 3, 1, halo=2, align=8)
    assert all(t.y1 > t.y0 and t.x1 > t.x0 for t in tiles) and tiles[-1].y1 == 10
    x = torch.arange(10 * 20, dtype=torch.float32).reshape(1, 1, 10, 20)
    y = upscale_tiled(lambda t: t.repeat_interleave(2, -1).repeat_interleave(2, -2), x, 2, tile=(3, 20), halo=2, align=8)
    assert torch.equal(y, x.repeat_interleave(2, -1).repeat_interleave(2, -2))


def test_choose_grid_keeps_tile_aspect():
    assert [choose_grid(n, 1080, 1920) for n in (1, 2, 4, 8)] == [(1, 1), (1, 2), (2, 2), (2, 4)]


def test_bench_power_sampler_matches_samples_by_wall_clock(tmp_path, monkeypatch):
    """bench.py's power leg: a child process samples `rocm-smi -d <this GPU>` (stubbed here) once the bench reaches its power leg -- it stays
    idle before ``begin()`` -- and the bench keeps the samples inside its window; without rocm-smi on the PATH the field is null instead of an error."""
    import importlib.util
    import stat
    import sys
    import time

    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fake = tmp_path / 'rocm-smi'
    # (the stub answers for the device the sampler asks for with -d: device 0 here)
    fake.write_text('#!/bin/sh\n[ "$1" = "-d" ] && [ "$2" = "0" ] || exit 1\n'
                    'echo "GPU[0]\t\t: sclk clock level: 1: (1970Mhz)"\necho "GPU[0]\t\t: Max Graphics Package Power (W): 1400.0"\n'
                    'echo "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1366.0"\n')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv('PATH', f'{tmp_path}{os.pathsep}{os.environ["PATH"]}')
    monkeypatch.delenv('HIP_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    s = bench.PowerSampler()
    assert s.proc is not None
    time.sleep(0.7)
    assert os.path.getsize(s.path) == 0  # idle until the power leg begins: nothing sampled during load / warm-up / the timed region
    s.begin()
    t0 = time.time()
    time.sleep(1.5)
    got = s.stop(t0, time.time())
    assert got is not None and got['samples'] >= 2 and got['avg_w'] == 1366.0 and got['cap_w'] == 1400.0 and got['sclk_mhz_avg'] == 1970  # the busy GPU
    assert s.proc.poll() is not None  # the child is gone
    monkeypatch.setenv('PATH', str(tmp_path / 'nothing'))
    none = bench.PowerSampler()
    assert none.proc is None and none.stop(0.0, time.time()) is None
    assert sys.executable
