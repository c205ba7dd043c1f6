"""Test infrastructure: a small FLAC ENCODER (pure Python / numpy) used to make fixtures for the in-tree decoder
(csrc/flac.cpp) inside the tests themselves -- no FLAC tool or library exists in the container. It writes valid
streams of the shapes the decoder claims: STREAMINFO (+ a PADDING block), fixed blocksize with a short last frame,
CONSTANT / VERBATIM / FIXED(0-4) / LPC subframes, Rice / Rice2 partitions incl. escaped ones, wasted bits, the four
stereo modes, CRC-8 / CRC-16."""
import numpy as np


class BitWriter:
    def __init__(self):
        self.acc, self.n = 0, 0

    def put(self, v, k):
        if k:
            self.acc = (self.acc << k) | (int(v) & ((1 << k) - 1))
            self.n += k

    def put_signed(self, v, k):
        self.put(int(v) & ((1 << k) - 1), k)

    def unary(self, q):
        self.put(1, int(q) + 1)  # q zeros then a one

    def align(self):
        if self.n & 7:
            self.put(0, 8 - (self.n & 7))

    def bytes(self):
        assert self.n % 8 == 0
        return self.acc.to_bytes(self.n // 8, "big") if self.n else b""


def crc8(d):
    c = 0
    for x in d:
        c ^= x
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(d):
    c = 0
    for x in d:
        c ^= x << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def _utf8_number(v):
    if v < 0x80:
        return bytes([v])
    out, n = [], 0
    while v >= (0x40 >> n):
        out.append(0x80 | (v & 0x3f))
        v >>= 6
        n += 1
    lead = (0xff << (7 - n)) & 0xff
    return bytes([lead | v] + out[::-1])


def _write_residual(bw, res, blocksize, order, porder, rice2=False, escape_first=False):
    bw.put(1 if rice2 else 0, 2)
    bw.put(porder, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    idx = 0
    for pi in range(1 << porder):
        cnt = (blocksize >> porder) - (order if pi == 0 else 0)
        part = res[idx:idx + cnt]
        idx += cnt
        u = np.where(part >= 0, 2 * part, -2 * part - 1).astype(np.int64)
        if escape_first and pi == 0:
            nb = max(1, int(np.max(np.abs(part))).bit_length() + 1) if cnt else 0
            bw.put(esc, pbits)
            bw.put(nb, 5)
            for v in part:
                bw.put_signed(v, nb)
            continue
        mean = float(u.mean()) if cnt else 0.0
        k = min(max(int(np.floor(np.log2(mean + 1))), 0), esc - 1)
        bw.put(k, pbits)
        for v in u:
            v = int(v)
            bw.unary(v >> k)
            bw.put(v & ((1 << k) - 1), k)
    assert idx == len(res)


def _lpc(x, order, prec=12):
    xf = x.astype(np.float64)
    r = np.array([np.dot(xf[:len(xf) - k], xf[k:]) for k in range(order + 1)])
    if r[0] == 0:
        return None
    a = np.zeros(order + 1)
    a[0] = 1.0
    err = r[0]
    for i in range(1, order + 1):
        acc = r[i] + np.dot(a[1:i], r[i - 1:0:-1])
        kref = -acc / err
        a[1:i + 1] = a[1:i + 1] + kref * np.concatenate([a[1:i][::-1], [1.0]])
        err *= (1 - kref * kref)
        if err <= 0:
            return None
    coef = -a[1:]
    mx = np.max(np.abs(coef))
    if mx == 0:
        return None
    shift = int(min(max(prec - 1 - int(np.ceil(np.log2(mx + 1e-12))) - 1, 0), 15))
    q = np.clip(np.round(coef * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(np.int64)
    return q, shift, prec


def _write_subframe(bw, x, bps, kind, porder=0, rice2=False, escape_first=False, lpc_order=8):
    """x: int64 samples of one channel of one frame. kind: 'constant' | 'verbatim' | ('fixed', order) | 'lpc'."""
    n = len(x)
    wasted = 0
    if np.any(x != 0):
        while np.all(x % (1 << (wasted + 1)) == 0) and wasted < bps - 1:
            wasted += 1
    if wasted:
        x = x >> wasted
        bps -= wasted
    if kind == "constant":
        assert np.all(x == x[0])
        code = 0
    elif kind == "verbatim":
        code = 1
    elif kind == "lpc":
        lp = _lpc(x, lpc_order)
        if lp is None:
            kind, code = "verbatim", 1
        else:
            code = 32 | (lpc_order - 1)
    else:
        code = 8 | kind[1]
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        bw.put_signed(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            bw.put_signed(v, bps)
    elif kind == "lpc":
        q, shift, prec = lp
        order = lpc_order
        for v in x[:order]:
            bw.put_signed(v, bps)
        bw.put(prec - 1, 4)
        bw.put_signed(shift, 5)
        for c in q:
            bw.put_signed(c, prec)
        pred = np.zeros(n, dtype=np.int64)
        for j in range(order):
            pred[order:] += q[j] * x[order - 1 - j:n - 1 - j]
        res = x[order:] - (pred[order:] >> shift)
        _write_residual(bw, res, n, order, porder, rice2, escape_first)
    else:
        order = kind[1]
        for v in x[:order]:
            bw.put_signed(v, bps)
        res = x.copy()
        for _ in range(order):
            res = np.diff(res)
        res = x[order:] - (x[order:] - res) if order == 0 else res
        _write_residual(bw, res, n, order, porder, rice2, escape_first)


def encode(pcm, sample_rate=16000, bps=16, blocksize=1024, kinds=("lpc",), stereo="independent", porder=2, rice2=False,
           escape_first=False, with_total=True, padding_block=True):
    """pcm: int array [n] or [channels][n] (values must fit `bps`). kinds: subframe kind per frame, cycled.
    stereo (2 channels): 'independent' | 'left_side' | 'right_side' | 'mid_side'."""
    pcm = np.asarray(pcm, dtype=np.int64)
    if pcm.ndim == 1:
        pcm = pcm[None]
    nch, total = pcm.shape
    frames = []
    fno = 0
    for start in range(0, total, blocksize):
        blk = pcm[:, start:start + blocksize]
        n = blk.shape[1]
        kind = kinds[fno % len(kinds)]
        hdr = BitWriter()
        hdr.put(0b11111111111110, 14)
        hdr.put(0, 1)
        hdr.put(0, 1)  # fixed blocksize stream: the coded number is the frame number
        bs_codes = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
        bs_code = bs_codes.get(n, 6 if n <= 256 else 7)
        hdr.put(bs_code, 4)
        hdr.put({8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10}.get(sample_rate, 0), 4)
        ch_code = nch - 1
        if nch == 2 and stereo != "independent":
            ch_code = {"left_side": 8, "right_side": 9, "mid_side": 10}[stereo]
        hdr.put(ch_code, 4)
        hdr.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bps, 0), 3)
        hdr.put(0, 1)
        hb = hdr.bytes() + _utf8_number(fno)
        if bs_code == 6:
            hb += bytes([n - 1])
        elif bs_code == 7:
            hb += (n - 1).to_bytes(2, "big")
        hb += bytes([crc8(hb)])
        body = BitWriter()
        chans = [blk[c] for c in range(nch)]
        widths = [bps] * nch
        if ch_code == 8:
            chans, widths = [blk[0], blk[0] - blk[1]], [bps, bps + 1]
        elif ch_code == 9:
            chans, widths = [blk[0] - blk[1], blk[1]], [bps + 1, bps]
        elif ch_code == 10:
            chans, widths = [(blk[0] + blk[1]) >> 1, blk[0] - blk[1]], [bps, bps + 1]
        for x, w in zip(chans, widths):
            k = kind
            po = porder
            if k == "constant" and not np.all(x == x[0]):
                k = ("fixed", 2)
            order = 0 if k in ("constant", "verbatim") else (8 if k == "lpc" else k[1])
            while po > 0 and ((n >> po) << po != n or (n >> po) <= order):
                po -= 1
            if k == "lpc" and n <= 8:
                k = "verbatim"
            _write_subframe(body, x, w, k, po, rice2, escape_first)
        body.align()
        fb = hb + body.bytes()
        fb += crc16(fb).to_bytes(2, "big")
        frames.append(fb)
        fno += 1
    si = BitWriter()
    si.put(blocksize, 16)
    si.put(blocksize, 16)
    si.put(0, 24)
    si.put(0, 24)
    si.put(sample_rate, 20)
    si.put(nch - 1, 3)
    si.put(bps - 1, 5)
    si.put(total if with_total else 0, 36)
    si.put(0, 128)
    sib = si.bytes()
    out = b"fLaC" + bytes([0x00 if padding_block else 0x80]) + len(sib).to_bytes(3, "big") + sib
    if padding_block:
        out += bytes([0x81]) + (16).to_bytes(3, "big") + bytes(16)
    return out + b"".join(frames)
