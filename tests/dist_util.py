"""Helpers for the shard/tile layout (include/pt_render.h: tile g -> shard g % n, local index g // n)."""
import numpy as np


def unshard_reference(gathered: np.ndarray, width: int, height: int, shard_count: int) -> np.ndarray:
    """numpy statement of pt_unshard_tiles: [shards][tiles_per_shard][64][3] -> [H][W][3]."""
    tiles_x = (width + 7) // 8
    fb = np.zeros((height, width, 3), dtype=np.float32)
    ys, xs = np.mgrid[0:height, 0:width]
    g = (ys // 8) * tiles_x + (xs // 8)
    fb[ys, xs] = gathered[g % shard_count, g // shard_count, (ys % 8) * 8 + (xs % 8)]
    return fb
