"""2-D tile decomposition: the arithmetic of the reference's ``distribute_mpi``
(reference pom/parallel_mpi.f:34-122), as a pure function of (rank, global size, tile size).

Tiles carry a 1-cell ghost rim, so neighbouring tiles share 2 columns/rows: local index ``i`` of
tile column ``px`` is global ``i + px*(im_local-2)`` (parallel_mpi.f:82).  East-/north-most tiles are
trimmed when the division is inexact (parallel_mpi.f:83-87,98-102).  Neighbour ``-1`` marks a
physical edge (parallel_mpi.f:111-119).
"""
from __future__ import annotations

from dataclasses import dataclass


def _ceil_div(a: int, b: int) -> int:
    return -(-a // b)


@dataclass(frozen=True)
class Tile:
    rank: int
    nproc_x: int
    nproc_y: int
    px: int
    py: int
    im_local: int
    jm_local: int
    im: int          # active extent in x (<= im_local)
    jm: int
    i_off: int       # global i = local i + i_off   (1-based on both sides)
    j_off: int
    n_west: int
    n_east: int
    n_south: int
    n_north: int
    # diagonal neighbours (the reference reaches them through its two exchange phases; the single-phase
    # device exchange sends them their corner cell directly)
    n_sw: int = -1
    n_se: int = -1
    n_nw: int = -1
    n_ne: int = -1


def tile_grid(im_global: int, jm_global: int, im_local: int, jm_local: int) -> tuple[int, int]:
    """Number of tiles in x and y (parallel_mpi.f:54-65)."""
    return _ceil_div(im_global - 2, im_local - 2), _ceil_div(jm_global - 2, jm_local - 2)


def local_size(im_global: int, jm_global: int, nproc_x: int, nproc_y: int) -> tuple[int, int]:
    """Smallest (im_local, jm_local) that covers the global grid with nproc_x x nproc_y tiles."""
    return _ceil_div(im_global - 2, nproc_x) + 2, _ceil_div(jm_global - 2, nproc_y) + 2


def make_tile(rank: int, im_global: int, jm_global: int, im_local: int, jm_local: int,
              n_proc: int | None = None) -> Tile:
    nproc_x, nproc_y = tile_grid(im_global, jm_global, im_local, jm_local)
    if n_proc is not None and nproc_x * nproc_y > n_proc:
        # parallel_mpi.f:68-75 -- "im_local or jm_local is too low"
        raise ValueError("im_local or jm_local is too low for n_proc")
    if not 0 <= rank < nproc_x * nproc_y:
        raise ValueError("rank outside the tile grid")
    px, py = rank % nproc_x, rank // nproc_x
    i_off, j_off = px * (im_local - 2), py * (jm_local - 2)
    im = min(im_local, im_global - i_off)
    jm = min(jm_local, jm_global - j_off)
    n_east = rank + 1 if px + 1 < nproc_x else -1
    n_west = rank - 1 if px > 0 else -1
    n_north = rank + nproc_x if py + 1 < nproc_y else -1
    n_south = rank - nproc_x if py > 0 else -1
    diag = lambda dx, dy: rank + dx + dy * nproc_x if 0 <= px + dx < nproc_x and 0 <= py + dy < nproc_y else -1
    return Tile(rank, nproc_x, nproc_y, px, py, im_local, jm_local, im, jm, i_off, j_off,
                n_west, n_east, n_south, n_north, diag(-1, -1), diag(1, -1), diag(-1, 1), diag(1, 1))


def choose_tile_grid(n: int, im_global: int, jm_global: int) -> tuple[int, int]:
    """Pick nproc_x x nproc_y = n.  The reference leaves the split to the user (im_local, jm_local in pom.h); here: whole rows
    (1 x n) when the tiles keep at least 128 rows, else the most square tiles (fewest halo cells).

    Measured (tools/grid_sweep.sh, one tile of 2048x1536x50 on one MI355X, ms per step without transfers): 8 tiles 1x8 6.97,
    2x4 7.14, 4x2 7.31, 8x1 7.60; 4 tiles 1x4 12.27, 2x2 12.57, 4x1 12.67; 2 tiles 1x2 21.28, 2x1 22.77 -- full-width rows keep
    every wavefront of the 3-D kernels full (a 514-column row ends in a wavefront with 18 of 62 columns), a tile then has two
    neighbours instead of eight and its edge lines are contiguous in memory."""
    iml, jml = local_size(im_global, jm_global, 1, n)
    if n > 1 and jml >= 128 and tile_grid(im_global, jm_global, iml, jml) == (1, n):
        return 1, n
    best = None
    for nx in range(1, n + 1):
        if n % nx:
            continue
        ny = n // nx
        iml, jml = local_size(im_global, jm_global, nx, ny)
        if tile_grid(im_global, jm_global, iml, jml) != (nx, ny):
            continue
        cost = iml * ny + jml * nx  # total interior edge length ~ halo volume
        if iml % 2:
            # the two-columns-per-lane kernels (16-byte loads) need an even leading dimension; an odd one falls back to
            # one-column kernels (measured on a 2-tile split of 2048x1536x50: 29.6 ms per step against 28.7)
            cost = cost * 5 // 4
        if best is None or cost < best[0]:
            best = (cost, nx, ny)
    if best is None:
        raise ValueError(f"cannot split {im_global}x{jm_global} into {n} tiles")
    return best[1], best[2]
