"""Test-only views into the three opaque state chunks the C ABI hands back
(frg_*_layout in include/frosting_rasterizer.h).  Not used by the render path."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _view(buf, off, dtype, n):
    esz = torch.empty(0, dtype=dtype).element_size()
    return buf[int(off):int(off) + n * esz].view(dtype)


class State:
    def __init__(self, P, W, H, R, geom, binning, img):
        L = _lib.lib()
        off = (C.c_longlong * 8)()
        L.frg_geometry_layout_n.argtypes = [C.c_int, C.c_void_p, C.c_int]
        self.P, self.W, self.H, self.R = P, W, H, R
        T = ((W + 15) // 16) * ((H + 15) // 16)
        N = W * H
        L.frg_geometry_layout_n(P, off, 6)
        stride = int(off[5]) // 4                      # floats between consecutive Gaussians (48-byte records)
        rec = _view(geom, off[0], torch.float32, stride * P).view(P, stride)
        self.xydr = rec[:, 0:4]
        self.conic_opacity = rec[:, 4:8]
        rgbc = rec[:, 8:12]
        self.rgb = rgbc[:, :3]
        self.flag_words = rgbc[:, 3].view(torch.int32)     # a VIEW (strided) of the records' flag words: byte 0 clamp flags, byte 1 the backward blend's "reached" mark
        self.clamp_bits = self.flag_words.contiguous() & 7
        self.tiles_touched = _view(geom, off[3], torch.int32, P)
        self.point_offsets = _view(geom, off[4], torch.int32, P)
        L.frg_image_layout(W, H, off)
        self.final_T = _view(img, off[0], torch.float32, N).view(H, W)
        self.n_contrib = _view(img, off[1], torch.int32, N).view(H, W)
        self.ranges = _view(img, off[2], torch.int32, 2 * T).view(T, 2)
        self.tile_count = _view(img, off[3], torch.int32, T)
        if R > 0:
            L.frg_binning_layout(R, 0, off)
            self.point_list = _view(binning, off[0], torch.int32, R)
        else:
            self.point_list = torch.empty(0, dtype=torch.int32, device=geom.device)

    @property
    def means2D(self):
        return self.xydr[:, :2]

    @property
    def depths(self):
        return self.xydr[:, 2]

    def sort_keys(self):
        """Reconstruct the reference's sorted 64-bit keys (tile << 32 | depth bits,
        rasterizer_impl.cu:102-104) from ranges + point_list + depth."""
        T = self.ranges.shape[0]
        counts = (self.ranges[:, 1] - self.ranges[:, 0]).to(torch.int64)
        tile_of = torch.repeat_interleave(torch.arange(T, device=counts.device, dtype=torch.int64), counts)
        dbits = self.depths.contiguous().view(torch.int32)[self.point_list.long()].to(torch.int64) & 0xFFFFFFFF
        return (tile_of << 32) | dbits
