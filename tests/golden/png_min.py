"""Minimal PNG reader (8/16-bit gray / RGB / RGBA, non-interlaced) — cv2 is not installed and PIL cannot return 16-bit RGB.
Returns an array [H, W, channels] in PNG channel order (R, G, B, A); note cv2.imread returns B, G, R, A."""
import struct
import zlib

import numpy as np


def read_png(path: str) -> np.ndarray:
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    W, H, depth, ctype, _, _, interlace = hdr
    assert interlace == 0 and depth in (8, 16) and ctype in (0, 2, 6)
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    bpp = ch * depth // 8
    stride = W * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(H, stride + 1)
    out = np.zeros((H, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(H):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(stride, dtype=np.int32)
            if f == 1:      # Sub: every byte-lane is a running sum mod 256
                cur = (np.cumsum(line.reshape(-1, bpp), axis=0) & 255).reshape(-1)
            else:           # Average / Paeth: sequential over pixels, vectorised over the bpp lanes
                ln, pv, cu = line.reshape(-1, bpp), prev.reshape(-1, bpp), cur.reshape(-1, bpp)
                left = np.zeros(bpp, dtype=np.int32)
                upleft = np.zeros(bpp, dtype=np.int32)
                for x in range(W):
                    up = pv[x]
                    if f == 3:
                        pred = (left + up) >> 1
                    else:
                        p = left + up - upleft
                        pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
                        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
                    left = (ln[x] + pred) & 255
                    cu[x] = left
                    upleft = up
                cur = cu.reshape(-1)
        out[y] = cur.astype(np.uint8)
        prev = cur
    if depth == 16:
        return out.reshape(H, W, ch, 2).astype(np.uint16).dot(np.array([256, 1], dtype=np.uint16)).astype(np.uint16)
    return out.reshape(H, W, ch)
