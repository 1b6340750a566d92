"""Test-side PIZ *encoder* for OpenEXR blocks (compression 4), written from the published format (ImfPizCompressor / ImfHuf / ImfWav of
OpenEXR): value bitmap -> forward table, two-dimensional wavelet (14-bit and modulo-2^16 forms), canonical Huffman codes with a
run-length symbol.  The C++ reader (csrc/host/image_io.cpp, namespace piz) holds the decode side only; nothing here shares code with it.
No PIZ file written by OpenEXR itself exists in this image (no OpenEXR library, no network), so the pair is checked against each other.
"""
import heapq
import struct

import numpy as np

BITMAP_SIZE = 8192


def _s16(x):
    x &= 0xffff
    return x - 0x10000 if x & 0x8000 else x


def wenc14(a, b):
    a, b = _s16(a), _s16(b)
    return ((a + b) >> 1) & 0xffff, (a - b) & 0xffff


def wenc16(a, b):
    ao = (a + 0x8000) & 0xffff
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 0x8000) & 0xffff
    return m & 0xffff, d & 0xffff


def wav2_encode(buf, start, nx, ox, ny, oy, mx):
    enc = wenc14 if mx < (1 << 14) else wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py, ey = start, start + oy * (ny - p2)
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = enc(buf[px], buf[p01])
                i10, i11 = enc(buf[p10], buf[p11])
                buf[px], buf[p10] = enc(i00, i10)
                buf[p01], buf[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                buf[px], buf[p10] = enc(buf[px], buf[p10])
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                buf[px], buf[p01] = enc(buf[px], buf[p01])
                px += ox2
        p, p2 = p2, p2 << 1


class _Bits:
    def __init__(self):
        self.out, self.c, self.lc, self.count = bytearray(), 0, 0, 0

    def put(self, nbits, value):
        self.c = (self.c << nbits) | (value & ((1 << nbits) - 1))
        self.lc += nbits
        self.count += nbits
        while self.lc >= 8:
            self.lc -= 8
            self.out.append((self.c >> self.lc) & 0xff)
        self.c &= (1 << self.lc) - 1

    def flush(self):
        if self.lc:
            self.out.append((self.c << (8 - self.lc)) & 0xff)
            self.c, self.lc = 0, 0
        return bytes(self.out)


def huf_compress(symbols, use_runs=True):
    """hufCompress: 20-byte header, packed code lengths, data bits.  `symbols`: sequence of u16."""
    symbols = [int(s) for s in symbols]
    if not symbols:
        return b""
    freq = {}
    for s in symbols:
        freq[s] = freq.get(s, 0) + 1
    im, iM = min(freq), max(freq) + 1     # the run-length symbol sits behind the largest one, with frequency 1
    rlc = iM
    freq[rlc] = 1
    # code lengths of a Huffman tree
    heap = [(f, s, (s,)) for s, f in freq.items()]
    heapq.heapify(heap)
    length = {s: 0 for s in freq}
    if len(heap) == 1:
        length[heap[0][1]] = 1
    while len(heap) > 1:
        fa, ka, sa = heapq.heappop(heap)
        fb, kb, sb = heapq.heappop(heap)
        for s in sa + sb:
            length[s] += 1
        heapq.heappush(heap, (fa + fb, min(ka, kb), sa + sb))
    assert max(length.values()) <= 58
    # canonical codes: hufCanonicalCodeTable
    n = [0] * 59
    for l in length.values():
        n[l] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    code = {}
    for s in sorted(length):
        l = length[s]
        if l > 0:
            code[s] = n[l]
            n[l] += 1
    # hufPackEncTable
    tb = _Bits()
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            zerun = 1
            while s < iM and zerun < 255 + 6:
                if length.get(s + 1, 0) > 0:
                    break
                s += 1
                zerun += 1
            if zerun >= 2:
                if zerun >= 6:
                    tb.put(6, 63); tb.put(8, zerun - 6)
                else:
                    tb.put(6, 59 + zerun - 2)
                s += 1
                continue
        tb.put(6, l)
        s += 1
    table = tb.flush()
    # hufEncode
    db = _Bits()

    def send(sym, run):
        ls, lr = length[sym], length[rlc]
        if use_runs and ls + lr + 8 < ls * run:
            db.put(ls, code[sym]); db.put(lr, code[rlc]); db.put(8, run)
        else:
            for _ in range(run + 1):
                db.put(ls, code[sym])
    s, cs = symbols[0], 0
    for t in symbols[1:]:
        if s == t and cs < 255:
            cs += 1
        else:
            send(s, cs)
            cs = 0
        s = t
    send(s, cs)
    nbits = db.count
    data = db.flush()
    return struct.pack("<5I", im, iM, len(table), nbits, 0) + table + data


def compress_block(planes):
    """`planes`: per channel a (rows, cols, size) u16 array, size = 1 (half) or 2 (the two halves of a 32-bit sample, low half first).
    Returns the PIZ block payload."""
    buf, layout = [], []
    for pl in planes:
        ny, nx, size = pl.shape
        layout.append((len(buf), nx, ny, size))
        buf.extend(int(v) for v in pl.reshape(-1))
    bitmap = bytearray(BITMAP_SIZE)
    for v in buf:
        bitmap[v >> 3] |= 1 << (v & 7)
    bitmap[0] &= 0xfe
    nz = [i for i in range(BITMAP_SIZE) if bitmap[i]]
    min_nz, max_nz = (nz[0], nz[-1]) if nz else (BITMAP_SIZE - 1, 0)
    lut, k = [0] * 65536, 0
    for i in range(65536):
        if i == 0 or (bitmap[i >> 3] & (1 << (i & 7))):
            lut[i] = k
            k += 1
    max_value = k - 1
    buf = [lut[v] for v in buf]
    for start, nx, ny, size in layout:
        for j in range(size):
            wav2_encode(buf, start + j, nx, size, ny, nx * size, max_value)
    huf = huf_compress(buf)
    out = struct.pack("<HH", min_nz, max_nz)
    if min_nz <= max_nz:
        out += bytes(bitmap[min_nz:max_nz + 1])
    return out + struct.pack("<i", len(huf)) + huf


def block_from_rows(rows_by_channel):
    """rows_by_channel: per channel (in file order) a 2-D numpy array of float16 or float32 / uint32 samples of the block."""
    planes = []
    for a in rows_by_channel:
        a = np.ascontiguousarray(a)
        size = a.dtype.itemsize // 2
        planes.append(a.view(np.uint16).reshape(a.shape[0], a.shape[1], size))
    return compress_block(planes)
