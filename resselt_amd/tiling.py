"""Tile-parallel whole-image inference.

The reference has no tiler and no distributed code; every SR model in scope is translation-equivariant up to its
receptive field, so an image shards into independent *input* tiles (3 channels: tiny) that are widened by a halo,
upscaled independently and cropped.  No activation ever crosses a GPU boundary; the only collective is the final
reassembly: an all-gather of equal-sized (padded) output tiles over RCCL/xGMI (``torch.distributed`` backend "nccl"
on ROCm), after which every rank holds the whole upscaled image.

Host logic only: works with any callable ``model(x) -> y`` (the CPU tests drive it with the oracle over gloo).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Sequence

import torch


@dataclass(frozen=True)
class Tile:
    """One work item in input-pixel coordinates. ``y0:y1, x0:x1`` is the region this tile OWNS in the output grid;
    ``ry0:ry1, rx0:rx1`` is what it reads (owned region + halo, clamped to the image)."""

    index: int
    y0: int
    y1: int
    x0: int
    x1: int
    ry0: int
    ry1: int
    rx0: int
    rx1: int

    @property
    def shape(self) -> tuple[int, int]:
        return self.y1 - self.y0, self.x1 - self.x0


def _splits(total: int, parts: int, align: int) -> list[int]:
    """Boundaries of at most `parts` near-equal chunks of [0, total), interior boundaries rounded to `align`.
    A dimension too small for `parts` aligned chunks yields fewer (never an empty chunk)."""
    edges = [0]
    for i in range(1, parts):
        e = round(total * i / parts / align) * align
        if edges[-1] < e < total:
            edges.append(e)
    edges.append(total)
    return edges


def choose_grid(n_tiles: int, height: int, width: int) -> tuple[int, int]:
    """rows x cols = n_tiles with tile aspect closest to square (e.g. 8 tiles of a 16:9 frame -> 2 x 4)."""
    best = None
    for rows in range(1, n_tiles + 1):
        if n_tiles % rows:
            continue
        cols = n_tiles // rows
        th, tw = height / rows, width / cols
        score = max(th / tw, tw / th)
        if best is None or score < best[0]:
            best = (score, rows, cols)
    return best[1], best[2]


def plan_tiles(height: int, width: int, rows: int, cols: int, halo: int = 32, align: int = 1) -> list[Tile]:
    """Partition an H x W input into rows x cols tiles.  ``align`` keeps interior tile edges AND halo-extended read
    windows on multiples of e.g. the SwinIR window, so window partitions of a tile coincide with the full frame's."""
    if rows < 1 or cols < 1 or halo < 0 or align < 1:
        raise ValueError('rows, cols >= 1, halo >= 0, align >= 1 required')
    if halo % align:
        halo = (halo // align + 1) * align
    ys, xs = _splits(height, rows, align), _splits(width, cols, align)
    tiles = []
    for r in range(len(ys) - 1):
        for c in range(len(xs) - 1):
            y0, y1, x0, x1 = ys[r], ys[r + 1], xs[c], xs[c + 1]
            tiles.append(Tile(len(tiles), y0, y1, x0, x1, max(y0 - halo, 0), min(y1 + halo, height), max(x0 - halo, 0), min(x1 + halo, width)))
    return tiles


def _is_image(x: torch.Tensor) -> bool:
    """uint8 [N, H, W, C] images (what a decoder delivers; models with ``supports_u8`` read and write them) vs float [N, C, H, W] tensors."""
    return x.dtype == torch.uint8


def _finish(y: torch.Tensor) -> torch.Tensor:
    """Last step of every helper that hands an image back to the user: synchronise and ask the engine whether any kernel reported a
    failed hand-off (``EngineModule.forward`` itself never synchronises, so a failure of the LAST launches would otherwise surface one
    call late).  A no-op for CPU tensors (the CPU tests drive this module with the oracle)."""
    if isinstance(y, torch.Tensor) and y.is_cuda:
        from .engine import lib as L

        torch.cuda.synchronize(y.device)
        L.check_status('upscale')
    return y


def run_tile(model: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor, tile: Tile, scale: int) -> torch.Tensor:
    """Upscale one tile (with its halo) and crop the halo off the result (a view of the model's output, not a copy)."""
    img = _is_image(x)
    if tile.y1 <= tile.y0 or tile.x1 <= tile.x0:
        return x.new_zeros((x.shape[0], 0, 0, x.shape[3]) if img else (x.shape[0], x.shape[1], 0, 0))
    crop = (x[:, tile.ry0 : tile.ry1, tile.rx0 : tile.rx1] if img else x[:, :, tile.ry0 : tile.ry1, tile.rx0 : tile.rx1]).contiguous()
    y = model(crop)
    oy, ox = (tile.y0 - tile.ry0) * scale, (tile.x0 - tile.rx0) * scale
    th, tw = tile.shape
    return y[:, oy : oy + th * scale, ox : ox + tw * scale] if img else y[:, :, oy : oy + th * scale, ox : ox + tw * scale]


def upscale_tiled(model, x: torch.Tensor, scale: int, tile: tuple[int, int], halo: int = 32, align: int = 1, check: bool = True) -> torch.Tensor:
    """Single-device tiling of a large image: bounds the engine's activation buffers (e.g. 8K inputs).  ``x``: a float ``[N, C, H, W]``
    tensor, or a uint8 ``[N, H, W, C]`` image for models with ``supports_u8`` (the layouts ``run_tile`` / ``TileParallel`` take)."""
    if x.dim() != 4:
        raise ValueError(f'expected a [N, C, H, W] tensor or a uint8 [N, H, W, C] image, got shape {tuple(x.shape)}')
    img = _is_image(x)
    n = x.shape[0]
    h, w = (x.shape[1], x.shape[2]) if img else (x.shape[2], x.shape[3])
    rows, cols = -(-h // tile[0]), -(-w // tile[1])
    out = None
    for t in plan_tiles(h, w, rows, cols, halo, align):
        y = run_tile(model, x, t, scale)
        if out is None:
            shape = (n, h * scale, w * scale, y.shape[3]) if img else (n, y.shape[1], h * scale, w * scale)
            out = torch.empty(shape, dtype=y.dtype, device=y.device)
        if img:
            out[:, t.y0 * scale : t.y1 * scale, t.x0 * scale : t.x1 * scale] = y
        else:
            out[:, :, t.y0 * scale : t.y1 * scale, t.x0 * scale : t.x1 * scale] = y
    return _finish(out) if check else out


def upscale(model, image: torch.Tensor, tile: tuple[int, int] | None = None, halo: int = 32, align: int = 1,
            dtype: torch.dtype = torch.float16, scale: int | None = None) -> torch.Tensor:
    """uint8 image in, uint8 image out: ``image`` is [H, W, C] or [N, H, W, C] uint8 on the GPU (what an image decoder delivers).

    ``/255`` and the NHWC->NCHW transpose run in one kernel, the model runs on ``dtype`` tensors (whole image, or ``tile``-sized tiles
    with ``halo`` pixels of context when the image is larger than ``tile``), and ``clamp(0, 1) * 255`` + round-half-even + NCHW->NHWC
    run in one kernel.  The reference leaves both conversions and the tiling to its callers.
    ``align``: SwinIR-family models want tile origins on a window multiple.
    """
    from .engine import ops

    if getattr(model, 'supports_u8', False) and image.dtype == torch.uint8:
        return _upscale_u8(model, image, tile, halo, align, scale)
    squeeze = image.dim() == 3
    x = ops.image_u8_to_nchw(image, dtype)
    if scale is None:
        scale = model.parameters_info.upscale
        if not isinstance(scale, int):
            raise ValueError('this model has several output scales: pass scale=')
    _, _, h, w = x.shape
    if tile is None or (h <= tile[0] and w <= tile[1]):
        y = model(x)
    else:
        y = upscale_tiled(model, x, scale, tile, halo, align, check=False)
    out = _finish(ops.nchw_to_image_u8(y))
    return out[0] if squeeze else out


def _upscale_u8(model, image: torch.Tensor, tile, halo: int, align: int, scale) -> torch.Tensor:
    """``upscale`` for models that read and write 8-bit images themselves (no separate conversion passes): whole image, or tiles."""
    squeeze = image.dim() == 3
    img = image.unsqueeze(0) if squeeze else image
    if scale is None:
        scale = model.parameters_info.upscale
    n, h, w, c = img.shape
    if tile is None or (h <= tile[0] and w <= tile[1]):
        out = model(img)
    else:
        out = upscale_tiled(model, img, scale, tile, halo, align, check=False)
    out = _finish(out)
    return out[0] if squeeze else out


class TileParallel:
    """``TileParallel(model, scale)(x)``: every rank of the process group upscales its share of the tiles of ``x`` and
    all ranks return the complete upscaled image.

    ``x`` must be the same full input on every rank (a 3-channel image is cheap to replicate; only the outputs and
    the activations are large): a float ``[N, C, H, W]`` tensor, or -- for models with ``supports_u8`` -- a uint8
    ``[N, H, W, C]`` image, in which case the tiles that cross the links are 8-bit (a quarter of the fp32 bytes).
    With ``world_size`` ranks the image is cut into ``world_size`` tiles (one per rank) unless ``grid`` says otherwise;
    tiles are dealt round-robin: rank r owns tiles r, r + world, ...

    The collective: one ``all_gather_into_tensor`` per ROUND of tiles (round k = tiles k*world .. k*world + world - 1),
    issued asynchronously as soon as this rank's tile of the round is finished, so it runs (on the backend's own stream)
    beside the computation of the next round's tile; all of them are waited for once, at the end.  When the tiles of a
    round are equal full-width row bands of a channel-interleaved image (uint8 ``[1, H, W, C]``: the bands of a round are
    one contiguous slab of the result) the gather writes straight into the result; otherwise tiles are padded to the
    largest tile, gathered into a receive buffer and copied into place.
    """

    def __init__(self, model: Callable[[torch.Tensor], torch.Tensor], scale: int, halo: int = 32, align: int = 1,
                 grid: tuple[int, int] | None = None, group=None, overlap: bool = True):  # fmt: skip
        self.model, self.scale, self.halo, self.align, self.grid, self.group, self.overlap = model, scale, halo, align, grid, group, overlap
        self.last_stats: dict = {}

    def tiles_for(self, height: int, width: int, world: int) -> list[Tile]:
        rows, cols = self.grid if self.grid is not None else choose_grid(world, height, width)
        return plan_tiles(height, width, rows, cols, self.halo, self.align)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            world, rank = 1, 0
        else:
            world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        img = _is_image(x)
        n = x.shape[0]
        h, w = (x.shape[1], x.shape[2]) if img else (x.shape[2], x.shape[3])
        s = self.scale
        tiles = self.tiles_for(h, w, world)
        mine = tiles[rank::world]

        def full_shape(c_out):
            return (n, h * s, w * s, c_out) if img else (n, c_out, h * s, w * s)

        def place(full, t, o):
            if img:
                full[:, t.y0 * s : t.y1 * s, t.x0 * s : t.x1 * s] = o
            else:
                full[:, :, t.y0 * s : t.y1 * s, t.x0 * s : t.x1 * s] = o

        if world == 1:
            full = None
            for t in mine:
                o = run_tile(self.model, x, t, s)
                if full is None:
                    full = torch.empty(full_shape(o.shape[3] if img else o.shape[1]), dtype=o.dtype, device=o.device)
                place(full, t, o)
            return _finish(full)

        rounds = -(-len(tiles) // world)
        # Can the gather of a round land in the result itself?  Channel-interleaved single image, every tile a full-width row band,
        # every round complete and made of equal bands.
        in_place = img and n == 1 and len(tiles) == rounds * world and all(t.x0 == 0 and t.x1 == w for t in tiles) and all(
            len({t.shape[0] for t in tiles[k * world : (k + 1) * world]}) == 1 for k in range(rounds))  # fmt: skip
        mh = max(t.shape[0] for t in tiles) * s
        mw = max(t.shape[1] for t in tiles) * s
        full = recv = None
        works, keep = [], []
        c_out = dtype = device = None
        for k in range(rounds):
            t = mine[k] if k < len(mine) else None
            o = run_tile(self.model, x, t, s) if t is not None else None
            if c_out is None:
                ref = o if (o is not None and o.numel()) else None
                c_out, dtype, device = self._out_meta(ref, x, img, idle_ranks=len(tiles) < world)
                full = torch.empty(full_shape(c_out), dtype=dtype, device=device)
                if not in_place:
                    recv = torch.empty((rounds, world, n, mh, mw, c_out) if img else (rounds, world, n, c_out, mh, mw), dtype=dtype, device=device)
            if in_place:
                band = tiles[k * world]
                dst = full[0, band.y0 * s : tiles[k * world + world - 1].y1 * s].reshape(-1)  # the round's bands: one contiguous slab
                src = o.reshape(-1) if o.is_contiguous() else o.contiguous().reshape(-1)
                out_t, in_t = dst, src
            else:
                send = torch.zeros((n, mh, mw, c_out) if img else (n, c_out, mh, mw), dtype=dtype, device=device)
                if o is not None and o.numel():
                    if img:
                        send[:, : o.shape[1], : o.shape[2]] = o
                    else:
                        send[:, :, : o.shape[2], : o.shape[3]] = o
                out_t, in_t = recv[k].reshape(-1), send.reshape(-1)
            keep.append(in_t)  # the input of an asynchronous collective must stay untouched (and alive) until it is waited for
            work = dist.all_gather_into_tensor(out_t, in_t, group=self.group, async_op=self.overlap)
            if self.overlap:
                works.append(work)
        for wk in works:
            wk.wait()
        if not in_place:
            for t in tiles:
                k, r = t.index // world, t.index % world
                th, tw = t.shape
                place(full, t, recv[k, r, :, : th * s, : tw * s] if img else recv[k, r, :, :, : th * s, : tw * s])
        self.last_stats = dict(rounds=rounds, in_place=bool(in_place), bytes_per_round=int(keep[0].numel() * keep[0].element_size()) * world,
                               tile_dtype=str(dtype), overlap=bool(self.overlap))  # fmt: skip
        return _finish(full)

    def _out_meta(self, ref, x, img: bool, idle_ranks: bool):
        """(channels, dtype, device) of the output tiles.  Only when there are fewer tiles than ranks can a rank have no
        output of its own; ``idle_ranks`` is the same on every rank, so all of them enter the exchange together."""
        import torch.distributed as dist

        if not idle_ranks:
            return (ref.shape[3] if img else ref.shape[1]), ref.dtype, ref.device
        mine = None if ref is None else ((ref.shape[3] if img else ref.shape[1]), ref.dtype)
        metas = [None] * dist.get_world_size(self.group)
        dist.all_gather_object(metas, mine, group=self.group)
        c_out, dtype = next(m for m in metas if m is not None)
        return c_out, dtype, (ref.device if ref is not None else x.device)
