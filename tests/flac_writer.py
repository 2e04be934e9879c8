"""A small FLAC ENCODER for the tests of ns_flac_decode, written from the format specification
(xiph.org/flac/format.html) independently of the decoder's code: the caller picks, per frame, the block size, the
channel assignment and, per subframe, the subframe type (constant / verbatim / fixed order 0-4 / LPC with given
coefficients), the Rice method, partition order and parameters (or escape partitions) and the wasted bits - so every
branch of the decoder can be driven.  STREAMINFO carries the MD5 of the PCM, which is what the product path checks."""
import hashlib

import numpy as np


class BitWriter(object):
    def __init__(self):
        self.bits = []

    def put(self, value, n):
        value = int(value)
        for i in range(n - 1, -1, -1):
            self.bits.append((value >> i) & 1)

    def put_signed(self, value, n):
        self.put(int(value) & ((1 << n) - 1), n)

    def unary(self, q):
        self.bits.extend([0] * int(q))
        self.bits.append(1)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    if v < 0x80:
        return [v]
    out, n = [], 0
    while True:
        n += 1
        lead_bits = 6 - n
        if v < (1 << (6 * n + lead_bits)):
            break
    lead = (0xFF << (7 - n)) & 0xFF
    out.append(lead | (v >> (6 * n)))
    for i in range(n - 1, -1, -1):
        out.append(0x80 | ((v >> (6 * i)) & 0x3F))
    return out


def _zigzag(r):
    return (r << 1) if r >= 0 else ((-r) << 1) - 1


def _write_residual(w, res, blocksize, order, method, porder, params):
    """params: one Rice parameter per partition, or ('esc', nbits)."""
    w.put(method, 2)
    w.put(porder, 4)
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    i = 0
    for pt in range(1 << porder):
        count = (blocksize >> porder) - (order if pt == 0 else 0)
        k = params[pt]
        if isinstance(k, tuple):
            nb = k[1]
            w.put(esc, pbits)
            w.put(nb, 5)
            for r in res[i:i + count]:
                w.put_signed(r, nb)
        else:
            w.put(k, pbits)
            for r in res[i:i + count]:
                u = _zigzag(int(r))
                w.unary(u >> k)
                w.put(u & ((1 << k) - 1), k)
        i += count
    assert i == len(res)


def _subframe(w, x, bps, spec):
    """x: int list of one channel's block at `bps` bits.  spec: dict(type=..., wasted=, order=, coefs=, shift=, prec=,
    method=, porder=, params=)."""
    blocksize = len(x)
    wasted = spec.get("wasted", 0)
    kind = spec["type"]
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + spec["order"]
    elif kind == "lpc":
        code = 32 + spec["order"] - 1
    w.put(0, 1)
    w.put(code, 6)
    if wasted:
        w.put(1, 1)
        w.unary(wasted - 1)
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
        bps -= wasted
    else:
        w.put(0, 1)
    if kind == "constant":
        assert all(v == x[0] for v in x)
        w.put_signed(x[0], bps)
        return
    if kind == "verbatim":
        for v in x:
            w.put_signed(v, bps)
        return
    order = spec["order"]
    for v in x[:order]:
        w.put_signed(v, bps)
    if kind == "fixed":
        coefs, shift = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order], 0
    else:
        coefs, shift, prec = spec["coefs"], spec["shift"], spec["prec"]
        w.put(prec - 1, 4)
        w.put_signed(shift, 5)
        for c in coefs:
            w.put_signed(c, prec)
    res = []
    for i in range(order, blocksize):
        pred = sum(c * x[i - 1 - j] for j, c in enumerate(coefs)) >> shift
        res.append(x[i] - pred)
    params = spec.get("params")
    porder = spec.get("porder", 0)
    if params is None:
        params = []
        i = 0
        for pt in range(1 << porder):
            count = (blocksize >> porder) - (order if pt == 0 else 0)
            part = res[i:i + count]
            mean = (sum(_zigzag(r) for r in part) / max(1, len(part))) if part else 0
            params.append(max(0, min(14, int(np.log2(mean + 1)))))
            i += count
    _write_residual(w, res, blocksize, order, spec.get("method", 0), porder, params)


_BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13,
                16384: 14, 32768: 15}
_BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}


def encode(pcm, bps, rate, frames, total_in_header=True, md5=True, variable=False, padding_block=0):
    """pcm int array [n, channels]; frames: list of dict(size=, assignment='independent'|'left_side'|'side_right'|
    'mid_side', subframes=[spec per channel], bps_from_streaminfo=bool).  Returns the file's bytes."""
    pcm = np.asarray(pcm, dtype=np.int64)
    n, C = pcm.shape
    out = bytearray(b"fLaC")
    sizes = [f["size"] for f in frames]
    assert sum(sizes) == n
    si = BitWriter()
    si.put(min(sizes[:-1] or sizes), 16)
    si.put(max(sizes), 16)
    si.put(0, 24)
    si.put(0, 24)
    si.put(rate, 20)
    si.put(C - 1, 3)
    si.put(bps - 1, 5)
    si.put(n if total_in_header else 0, 36)
    nb = (bps + 7) // 8
    raw = pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nb].tobytes()
    digest = hashlib.md5(raw).digest() if md5 else bytes(16)
    body = si.bytes() + digest
    last = 0 if padding_block else 0x80
    out += bytes([last | 0, 0, 0, len(body)]) + body
    if padding_block:
        out += bytes([0x80 | 1, 0, 0, padding_block]) + bytes(padding_block)
    pos = 0
    for fi, f in enumerate(frames):
        bs = f["size"]
        blk = pcm[pos:pos + bs]
        w = BitWriter()
        w.put(0b11111111111110, 14)
        w.put(0, 1)
        w.put(1 if variable else 0, 1)
        bcode = _BLOCK_CODES.get(bs)
        if bcode is None:
            bcode = 6 if bs <= 256 else 7
        w.put(bcode, 4)
        rcode = {8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}.get(rate, 0)
        if f.get("rate_field"):
            rcode = f["rate_field"]
        w.put(rcode, 4)
        asg = f.get("assignment", "independent")
        w.put({"independent": C - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[asg], 4)
        w.put(0 if f.get("bps_from_streaminfo") else _BPS_CODES[bps], 3)
        w.put(0, 1)
        for b in utf8_number(pos if variable else fi + f.get("number_offset", 0)):
            w.put(b, 8)
        if bcode == 6:
            w.put(bs - 1, 8)
        elif bcode == 7:
            w.put(bs - 1, 16)
        if rcode == 12:
            w.put(rate // 1000, 8)
        elif rcode == 13:
            w.put(rate, 16)
        elif rcode == 14:
            w.put(rate // 10, 16)
        hdr = w.bytes()
        w.put(crc8(hdr), 8)
        chans = [blk[:, c].tolist() for c in range(C)]
        widths = [bps] * C
        if asg == "left_side":
            chans = [chans[0], [a - b for a, b in zip(chans[0], chans[1])]]
            widths = [bps, bps + 1]
        elif asg == "side_right":
            chans = [[a - b for a, b in zip(chans[0], chans[1])], chans[1]]
            widths = [bps + 1, bps]
        elif asg == "mid_side":
            chans = [[(a + b) >> 1 for a, b in zip(chans[0], chans[1])], [a - b for a, b in zip(chans[0], chans[1])]]
            widths = [bps, bps + 1]
        for c in range(C):
            _subframe(w, chans[c], widths[c], f["subframes"][c])
        w.align()
        body = w.bytes()
        out += body + bytes([crc16(body) >> 8, crc16(body) & 0xFF])
        pos += bs
    return bytes(out)
