"""Fixed 2-D sin-cos position tables (reference: Models/mae/util/pos_embed.py:20-67).

Computed in float64 on the host, exactly once per model; half of the channels encode the patch's
column (w) coordinate, the other half its row (h), each as [sin | cos] over dim/4 frequencies.
"""
from __future__ import annotations

import numpy as np


def _axis_table(dim: int, coords: np.ndarray) -> np.ndarray:
    assert dim % 2 == 0
    freq = np.power(10000.0, -np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    phase = coords.astype(np.float64).reshape(-1, 1) * freq.reshape(1, -1)
    return np.concatenate([np.sin(phase), np.cos(phase)], axis=1)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    """[grid*grid (+1), embed_dim]; row-major over (h, w) patches; cls row (if any) is zero."""
    assert embed_dim % 2 == 0
    ww, hh = np.meshgrid(np.arange(grid_size, dtype=np.float32),
                         np.arange(grid_size, dtype=np.float32))  # w varies fastest
    table = np.concatenate([_axis_table(embed_dim // 2, ww), _axis_table(embed_dim // 2, hh)], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros((1, embed_dim)), table], axis=0)
    return table
