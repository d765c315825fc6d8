#!/usr/bin/env python3
"""Writes the two TGA fixtures next to this script (data for tests/test_assets_e2e.py):
  checker24.tga      8x8, type 2 (uncompressed), 24 bpp, bottom-left origin (descriptor 0x00)
  checker32_rle.tga  8x8, type 10 (RLE), 32 bpp, top-left origin (descriptor 0x28), runs and raw packets mixed
Pixel (x, y) of the DECODED top-down image is the same in both: R = 16 + 28x, G = 240 - 24y, B = 64 + 96((x+y)&1),
A = 255 - 8x - 4y (32 bpp only).  cube.obj in this directory is hand-written."""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
W = H = 8


def px(x, y, alpha):
    r, g, b = 16 + 28 * x, 240 - 24 * y, 64 + 96 * ((x + y) & 1)
    return bytes([b, g, r]) + (bytes([255 - 8 * x - 4 * y]) if alpha else b"")   # file order is B,G,R(,A)


def main():
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, W, H, 24, 0x00)
    rows = [b"".join(px(x, y, False) for x in range(W)) for y in range(H)]
    with open(os.path.join(HERE, "checker24.tga"), "wb") as fh:
        fh.write(hdr + b"".join(reversed(rows)))                                  # bottom row first
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, W, H, 32, 0x28)
    body = b""
    flat = [px(x, y, True) for y in range(H) for x in range(W)]
    i = 0
    while i < len(flat):                                                          # alternate raw packets of 3 and 5 pixels
        n = min(3 if (i // 3) % 2 == 0 else 5, len(flat) - i)
        body += bytes([n - 1]) + b"".join(flat[i:i + n])
        i += n
    with open(os.path.join(HERE, "checker32_rle.tga"), "wb") as fh:
        fh.write(hdr + body)
    # a second RLE file whose first packet is a run (all pixels of row 0 equal), appended rows raw
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, W, H, 24, 0x20)
    body = bytes([0x80 | (W - 1)]) + bytes([10, 200, 90])                          # run of 8 x (B,G,R) = (10,200,90)
    for y in range(1, H):
        body += bytes([W - 1]) + b"".join(px(x, y, False) for x in range(W))
    with open(os.path.join(HERE, "run24_rle.tga"), "wb") as fh:
        fh.write(hdr + body)


if __name__ == "__main__":
    main()
