"""HEALPix face topology as gather tables (host side of dlwp_healpix_pad_f32 / dlwp_conv3x3_hpx_f32).

The reference pads every face with slices of its neighbours by running ~20 tensor ops per face per call
(utils/healpix.py:165-368).  The topology is static, so here it is evaluated ONCE, on cell indices instead of
values, into a table the kernels gather through: entry (a, b) per padded cell, a/b = face*H*W + pixel,
b = -1 for a plain copy, else the cell is the mean of both (the two corners an equatorial face has no
neighbour for, healpix.py:316-368).

Face order and orientation follow the reference (healpix.py:209-226): faces 0-3 north, 4-7 equator,
8-11 south; the neighbour of face f across each of its 8 borders is listed in `_NEIGHBOURS`, with the
quarter-turns that border crossing applies.
"""
import functools

import numpy as np
import torch

# per face group: border -> (neighbour face as a function of k = f % 4, quarter-turns counter-clockwise)
# borders: T top, B bottom, L left, R right and the four corners.  None = synthesised corner.
_N = lambda off: (lambda k: (k + off) % 4)
_E = lambda off: (lambda k: 4 + (k + off) % 4)
_S = lambda off: (lambda k: 8 + (k + off) % 4)
_NEIGHBOURS = (
    # north (healpix.py:209-212, pn :229-258)
    {"T": (_N(1), 1), "TL": (_N(2), 2), "L": (_N(3), -1), "BL": (_N(3), 0), "B": (_E(0), 0), "BR": (_S(0), 0),
     "R": (_E(1), 0), "TR": (_N(1), 0)},
    # equator (healpix.py:215-218, pe :260-283)
    {"T": (_N(0), 0), "TL": None, "L": (_N(3), 0), "BL": (_E(3), 0), "B": (_S(3), 0), "BR": None,
     "R": (_S(0), 0), "TR": (_E(1), 0)},
    # south (healpix.py:221-224, ps :285-314)
    {"T": (_E(1), 0), "TL": (_N(0), 0), "L": (_E(0), 0), "BL": (_S(3), 0), "B": (_S(3), 1), "BR": (_S(2), 2),
     "R": (_S(1), -1), "TR": (_S(1), 0)},
)


def _cells(face: int, h: int, w: int) -> np.ndarray:
    """[h, w, 2] descriptors (a, b) of one face's own cells."""
    a = face * h * w + np.arange(h * w, dtype=np.int64).reshape(h, w)
    return np.stack([a, np.full_like(a, -1)], axis=-1)


def _corner_from_two(first: np.ndarray, second: np.ndarray, p: int, top_left: bool) -> np.ndarray:
    """The p x p corner no face covers: off-diagonal cells continue the two adjacent faces, the diagonal is
    their mean (healpix.py:316-343 for top-left with (t, l), :345-368 for bottom-right with (b, r))."""
    out = np.full((p, p, 2), -1, dtype=np.int64)
    for i in range(p):
        if top_left:
            d = p - 1 - i                       # diagonal cell, counted from the inner corner outwards
            out[d, d] = (first[-i - 1, 0, 0], second[0, -i - 1, 0])
            if i:
                out[d, p - i:, 0] = first[-i - 1, :i, 0]
                out[p - i:, d, 0] = second[:i, -i - 1, 0]
        else:
            out[i, i] = (first[i, -1, 0], second[-1, i, 0])
            if i:
                out[:i, i, 0] = second[-i:, i, 0]
                out[i, :i, 0] = first[i, -i:, 0]
    return out


@functools.lru_cache(maxsize=32)
def pad_table(h: int, w: int, p: int) -> torch.Tensor:
    """int32 [12, (h+2p)*(w+2p), 2] gather table for HEALPixPadding(p) (CPU tensor; cache per shape)."""
    if h != w:
        raise ValueError("HEALPix faces are square")
    if not 0 < p <= h:
        raise ValueError(f"padding {p} does not fit a {h}x{w} face")
    table = np.empty((12, h + 2 * p, w + 2 * p, 2), dtype=np.int64)
    for f in range(12):
        rules, k = _NEIGHBOURS[f // 4], f % 4

        def nb(border):
            face_of, turns = rules[border]
            return np.rot90(_cells(face_of(k), h, w), turns, axes=(0, 1))

        pad = table[f]
        pad[p:-p, p:-p] = _cells(f, h, w)
        pad[:p, p:-p] = nb("T")[-p:, :]
        pad[-p:, p:-p] = nb("B")[:p, :]
        pad[p:-p, :p] = nb("L")[:, -p:]
        pad[p:-p, -p:] = nb("R")[:, :p]
        pad[-p:, :p] = nb("BL")[:p, -p:]
        pad[:p, -p:] = nb("TR")[-p:, :p]
        pad[:p, :p] = nb("TL")[-p:, -p:] if rules["TL"] else _corner_from_two(nb("T"), nb("L"), p, True)
        pad[-p:, -p:] = nb("BR")[:p, :p] if rules["BR"] else _corner_from_two(nb("B"), nb("R"), p, False)
    return torch.from_numpy(table.reshape(12, -1, 2).astype(np.int32))


_device_tables = {}


def device_table(h: int, w: int, p: int, device) -> torch.Tensor:
    key = (h, w, p, str(device))
    if key not in _device_tables:
        _device_tables[key] = pad_table(h, w, p).to(device).contiguous()
    return _device_tables[key]
