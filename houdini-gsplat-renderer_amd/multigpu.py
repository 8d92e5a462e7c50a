"""Tile-row sharding across ranks: one process per GPU, ONE collective per frame.

Rank g renders the tile rows r with r % G == g (interleaved for load balance) into a compact
"band" image; rank 0 gathers the G bands (RCCL gather over xGMI when the tensors live on
GPUs, gloo on CPU for tests) and de-interleaves them into the full frame.  Per-pixel work is
identical to the unsharded frame, so the stitched image is BIT-identical to a 1-GPU render.
The reference has no multi-GPU path (single GL context, /root/reference/README.md:68-71).
"""
from __future__ import annotations

import numpy as np

TILE = 16


def tiles_y(height: int) -> int:
    return (height + TILE - 1) // TILE


def band_rows(height: int, count: int) -> int:
    """pixel rows of every rank's band image (uniform, padded) -- mirrors gsr_band_rows()"""
    return ((tiles_y(height) + count - 1) // count) * TILE


def owned_tile_rows(height: int, index: int, count: int, layout: int = 0) -> list[int]:
    """layout 0: interleaved rows index, index+count, ...; layout 1: the contiguous band [index*rpb, (index+1)*rpb)"""
    ty = tiles_y(height)
    if layout == 1 and count > 1:
        rpb = (ty + count - 1) // count
        return list(range(index * rpb, min((index + 1) * rpb, ty)))
    return list(range(index, ty, count))


def extract_band(full: np.ndarray, index: int, count: int, layout: int = 0) -> np.ndarray:
    """what rank `index` would render: its owned tile rows of `full` stacked bottom-up"""
    h, w = full.shape[0], full.shape[1]
    out = np.zeros((band_rows(h, count), w) + full.shape[2:], dtype=full.dtype)
    for lrow, trow in enumerate(owned_tile_rows(h, index, count, layout)):
        y0, y1 = trow * TILE, min(trow * TILE + TILE, h)
        out[lrow * TILE: lrow * TILE + (y1 - y0)] = full[y0:y1]
    return out


def stitch_bands_host(gathered: np.ndarray, height: int, layout: int = 0) -> np.ndarray:
    """gathered [G, band_rows, W, C] -> [H, W, C]; host mirror of the k_stitch_bands kernel"""
    count = gathered.shape[0]
    out = np.zeros((height,) + gathered.shape[2:], dtype=gathered.dtype)
    for g in range(count):
        for lrow, trow in enumerate(owned_tile_rows(height, g, count, layout)):
            y0, y1 = trow * TILE, min(trow * TILE + TILE, height)
            out[y0:y1] = gathered[g, lrow * TILE: lrow * TILE + (y1 - y0)]
    return out


class FrameGatherer:
    """Gathers band tensors to rank 0 and stitches the frame.  `dist` is torch.distributed
    (already initialised: backend nccl == RCCL on ROCm, or gloo on CPU) or None for 1 rank."""

    def __init__(self, dist, rank: int, world: int, width: int, height: int, device, engine=None,
                 via_host: bool = False, layout: int = 0):
        import torch

        self.layout = layout        # 0 = interleaved tile rows, 1 = contiguous bands (GSR_OPT_SHARD_LAYOUT)
        self.via_host = via_host    # functional-test mode: collective on host copies (backend without GPU support)
        self.dist, self.rank, self.world = dist, rank, world
        self.width, self.height = width, height
        self.engine = engine
        self.rows = band_rows(height, world) if world > 1 else height
        self.band = torch.zeros((self.rows, width, 4), dtype=torch.float32, device=device)
        self.gathered = self.final = None
        if world > 1 and rank == 0:
            self.gathered = torch.zeros((world, self.rows, width, 4), dtype=torch.float32, device=device)
            self.final = torch.zeros((height, width, 4), dtype=torch.float32, device=device)

    def gather_and_stitch(self):
        """returns the full frame tensor on rank 0 (None elsewhere); 1 rank: the band itself"""
        if self.world == 1:
            return self.band
        if self.via_host and self.band.is_cuda:
            import torch
            hb = self.band.cpu()
            hg = [torch.empty_like(hb) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(hb, hg, dst=0)
            if self.rank == 0:
                self.gathered.copy_(torch.stack(hg))
        else:
            self.dist.gather(self.band, list(self.gathered.unbind(0)) if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        if self.band.is_cuda:
            # device-side de-interleave on the engine's stream (same stream as the gather's consumer)
            self.engine.stitch_bands(self.gathered.data_ptr(), self.world, self.width, self.height,
                                     self.final.data_ptr())
        else:
            import torch
            self.final.copy_(torch.from_numpy(stitch_bands_host(self.gathered.numpy(), self.height, self.layout)))
        return self.final
