"""TEST INFRASTRUCTURE - CPU restatement of the baseline JPEG decode behind the texture ingest (numpy + plain Python).

The reference reads a scan's texture with ``vtk.vtkJPEGReader`` (src/mvlm/utils/utils3d.py:28-34, :42-48, :457-462).
The decoder itself is not part of /root/reference: VTK bundles libjpeg-turbo (ThirdParty/jpeg/vtkjpeg, VTK 8.x / 9.x) and
reads with libjpeg's defaults - JDCT_ISLOW, fancy upsampling, JCS_RGB output.  This file restates that published algorithm
(ITU T.81 entropy decoding; libjpeg's jidctint.c / jdsample.c / jdcolor.c integer arithmetic) and is PINNED against
Pillow's decoder in this image (libjpeg-turbo, same defaults): tests/test_jpeg_cpu.py decodes every fixture under
tests/golden/jpeg/ with both and requires identical bytes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product decodes on the GPU
(mvlm_amd/csrc/jpeg.hip) or - for the kinds that kernel does not take (progressive, arithmetic, CMYK, 12 bit) - with libjpeg
through Pillow.

Scope: baseline / extended sequential Huffman (SOF0 / SOF1), 8 bit, one interleaved scan, 1 (grey) or 3 (YCbCr) components,
luma sampling 1x1 / 2x1 / 2x2 with 1x1 chroma, restart intervals.  ``parse`` raises Unsupported for anything else.
"""
from __future__ import annotations

import numpy as np

# zigzag position -> natural (row-major) position, T.81 figure A.6 / libjpeg jutils.c jpeg_natural_order
NATURAL_ORDER = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
    62, 63], dtype=np.int64)


class Unsupported(ValueError):
    """A JPEG this restatement (and the GPU decoder) does not take; the product hands such a file to libjpeg."""


def _u16(b: bytes, p: int) -> int:
    return (b[p] << 8) | b[p + 1]


def parse(data: bytes) -> dict:
    """Marker segments up to the first SOS + the entropy-coded segment behind it (T.81 annex B)."""
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        raise Unsupported("not a JPEG stream")
    p = 2
    quant: dict[int, np.ndarray] = {}
    huff: dict[tuple[int, int], tuple[np.ndarray, np.ndarray]] = {}
    frame = None
    restart_interval = 0
    jfif = False
    adobe_transform = None
    while True:
        while p < len(data) and data[p] != 0xFF:
            p += 1  # (garbage between segments is skipped like libjpeg's next_marker does)
        while p < len(data) and data[p] == 0xFF:
            p += 1
        if p >= len(data):
            raise Unsupported("no scan")
        m = data[p]
        p += 1
        if m == 0xD8 or 0xD0 <= m <= 0xD7 or m == 0x01:
            continue
        if m == 0xD9:
            raise Unsupported("no scan")
        seg_len = _u16(data, p)
        seg = data[p + 2:p + seg_len]
        if len(seg) != seg_len - 2:
            raise Unsupported("truncated segment")
        p += seg_len
        if m == 0xDB:  # DQT
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                q += 1
                if pq == 0:
                    tab = np.frombuffer(seg[q:q + 64], np.uint8).astype(np.int32)
                    q += 64
                else:
                    tab = np.frombuffer(seg[q:q + 128], ">u2").astype(np.int32)
                    q += 128
                if len(tab) != 64 or tq > 3:
                    raise Unsupported("bad DQT")
                if tab.max() > 255:
                    raise Unsupported("quantisation table with 16-bit entries")
                nat = np.zeros(64, np.int32)
                nat[NATURAL_ORDER] = tab  # the file holds zigzag order
                quant[tq] = nat
        elif m == 0xC4:  # DHT
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = np.frombuffer(seg[q + 1:q + 17], np.uint8).astype(np.int64)
                n = int(counts.sum())
                vals = np.frombuffer(seg[q + 17:q + 17 + n], np.uint8).astype(np.int64)
                if len(counts) != 16 or len(vals) != n or tc > 1 or th > 3 or n > 256:
                    raise Unsupported("bad DHT")
                huff[(tc, th)] = (counts, vals)
                q += 17 + n
        elif m in (0xC0, 0xC1):  # SOF0 / SOF1: sequential Huffman
            if seg[0] != 8:
                raise Unsupported("sample precision other than 8 bit")
            h, w, nc = _u16(seg, 1), _u16(seg, 3), seg[5]
            comps = [dict(id=seg[6 + 3 * i], h=seg[7 + 3 * i] >> 4, v=seg[7 + 3 * i] & 15, tq=seg[8 + 3 * i]) for i in range(nc)]
            frame = dict(height=h, width=w, comps=comps)
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported("progressive / lossless / arithmetic JPEG (SOF%d)" % (m - 0xC0))
        elif m == 0xDD:
            restart_interval = _u16(seg, 0)
        elif m == 0xE0 and seg[:5] == b"JFIF\0":
            jfif = True
        elif m == 0xEE and seg[:5] == b"Adobe" and len(seg) >= 12:
            adobe_transform = seg[11]
        elif m == 0xDA:  # SOS
            if frame is None:
                raise Unsupported("SOS before SOF")
            ns = seg[0]
            if ns != len(frame["comps"]):
                raise Unsupported("a scan that does not hold every component")
            for i in range(ns):
                cid, tabs = seg[1 + 2 * i], seg[2 + 2 * i]
                if cid != frame["comps"][i]["id"]:
                    raise Unsupported("scan components out of frame order")
                frame["comps"][i]["td"], frame["comps"][i]["ta"] = tabs >> 4, tabs & 15
            if seg[1 + 2 * ns] != 0 or seg[2 + 2 * ns] != 63 or seg[3 + 2 * ns] != 0:
                raise Unsupported("spectral selection / successive approximation in a sequential scan")
            break
    comps = frame["comps"]
    if frame["height"] == 0 or frame["width"] == 0:
        raise Unsupported("empty frame (DNL)")
    if len(comps) == 3:
        # jdapimin.c default_decompress_parms: JFIF says YCbCr; Adobe says by its transform flag; else by the component ids
        if jfif:
            ycc = True
        elif adobe_transform is not None:
            ycc = adobe_transform == 1
            if adobe_transform not in (0, 1):
                ycc = True
        else:
            ids = tuple(c["id"] for c in comps)
            ycc = ids != (82, 71, 66)  # 'R' 'G' 'B'
        if not ycc:
            raise Unsupported("three components that are not YCbCr")
        if (comps[0]["h"], comps[0]["v"]) not in ((1, 1), (2, 1), (2, 2)) or any((c["h"], c["v"]) != (1, 1) for c in comps[1:]):
            raise Unsupported("sampling factors other than 1x1 / 2x1 / 2x2 luma over 1x1 chroma")
    elif len(comps) == 1:
        comps[0]["h"] = comps[0]["v"] = 1  # a single-component scan is never interleaved (T.81 A.2.2)
    else:
        raise Unsupported("%d components" % len(comps))
    for c in comps:
        if c["tq"] not in quant or (0, c["td"]) not in huff or (1, c["ta"]) not in huff:
            raise Unsupported("a table the scan names is missing")
    # the entropy-coded segment: up to the first marker that is neither a stuffed zero nor RSTn
    start = p
    segments = []  # restart intervals, byte stuffing removed
    cur = bytearray()
    while True:
        q = data.find(b"\xff", p)
        if q < 0 or q + 1 >= len(data):
            cur += data[p:]
            p = len(data)
            break
        cur += data[p:q]
        nxt = data[q + 1]
        if nxt == 0:
            cur.append(0xFF)
            p = q + 2
        elif 0xD0 <= nxt <= 0xD7:
            segments.append(bytes(cur))
            cur = bytearray()
            p = q + 2
        elif nxt == 0xFF:
            p = q + 1  # fill byte
        else:
            p = q
            break
    segments.append(bytes(cur))
    return dict(frame=frame, quant=quant, huff=huff, restart_interval=restart_interval, segments=segments,
                scan_start=start, scan_end=p)


def _huff_lookup(counts: np.ndarray, vals: np.ndarray):
    """T.81 annex C / F.2.2.3: canonical codes -> {(length, code): value}."""
    table = {}
    code = 0
    k = 0
    for length in range(1, 17):
        for _ in range(int(counts[length - 1])):
            table[(length, code)] = int(vals[k])
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    def __init__(self, data: bytes):
        self.data = data
        self.pos = 0  # in bits

    def bit(self) -> int:
        byte = self.pos >> 3
        # past the end: libjpeg feeds zero bits (jdhuff.c jpeg_fill_bit_buffer "no_more_bytes"); a segment's own padding is 1s
        v = (self.data[byte] >> (7 - (self.pos & 7))) & 1 if byte < len(self.data) else 0
        self.pos += 1
        return v

    def bits(self, n: int) -> int:
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v

    def symbol(self, table: dict) -> int:
        code = 0
        for length in range(1, 17):
            code = (code << 1) | self.bit()
            hit = table.get((length, code))
            if hit is not None:
                return hit
        raise Unsupported("corrupt entropy-coded data")


def _extend(v: int, s: int) -> int:
    return v if s == 0 or v >= (1 << (s - 1)) else v - (1 << s) + 1  # T.81 F.2.2.1 EXTEND


def geometry(frame: dict) -> dict:
    comps = frame["comps"]
    hmax = max(c["h"] for c in comps)
    vmax = max(c["v"] for c in comps)
    mcus_x = -(-frame["width"] // (8 * hmax))
    mcus_y = -(-frame["height"] // (8 * vmax))
    return dict(hmax=hmax, vmax=vmax, mcus_x=mcus_x, mcus_y=mcus_y)


def decode_coefficients(info: dict) -> list[np.ndarray]:
    """Entropy decoding (T.81 F.2.2): per component [block rows, block columns, 64] quantised coefficients, natural order."""
    frame = info["frame"]
    comps = frame["comps"]
    g = geometry(frame)
    tables = {k: _huff_lookup(*v) for k, v in info["huff"].items()}
    out = [np.zeros((g["mcus_y"] * c["v"], g["mcus_x"] * c["h"], 64), np.int32) for c in comps]
    n_mcus = g["mcus_x"] * g["mcus_y"]
    ri = info["restart_interval"] or n_mcus
    mcu = 0
    for seg in info["segments"]:
        br = _Bits(seg)
        pred = [0] * len(comps)
        for _ in range(ri):
            if mcu >= n_mcus:
                break
            my, mx = divmod(mcu, g["mcus_x"])
            for ci, c in enumerate(comps):
                dc_t, ac_t = tables[(0, c["td"])], tables[(1, c["ta"])]
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        blk = out[ci][my * c["v"] + by, mx * c["h"] + bx]
                        s = br.symbol(dc_t)
                        pred[ci] += _extend(br.bits(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = br.symbol(ac_t)
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break  # EOB
                                k += 16
                                continue
                            k += r
                            if k > 63:
                                raise Unsupported("corrupt entropy-coded data")
                            blk[NATURAL_ORDER[k]] = _extend(br.bits(s), s)
                            k += 1
            mcu += 1
    return out


# jidctint.c (libjpeg 6b / libjpeg-turbo): CONST_BITS 13, PASS1_BITS 2, constants FIX(x) = round(x * 2^13)
_C = dict(c0_298=2446, c0_390=3196, c0_541=4433, c0_765=6270, c0_899=7373, c1_175=9633, c1_501=12299, c1_847=15137,
          c1_961=16069, c2_053=16819, c2_562=20995, c3_072=25172)


def _idct_1d(x, shift):
    """One pass of jpeg_idct_islow over the LAST axis of x (int64 [..., 8]); DESCALE by ``shift`` bits."""
    c = _C
    x0, x1, x2, x3, x4, x5, x6, x7 = (x[..., i] for i in range(8))
    z1 = (x2 + x6) * c["c0_541"]
    tmp2 = z1 - x6 * c["c1_847"]
    tmp3 = z1 + x2 * c["c0_765"]
    tmp0 = (x0 + x4) << 13
    tmp1 = (x0 - x4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = x7, x5, x3, x1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * c["c1_175"]
    t0 = t0 * c["c0_298"]
    t1 = t1 * c["c2_053"]
    t2 = t2 * c["c3_072"]
    t3 = t3 * c["c1_501"]
    z1 = -z1 * c["c0_899"]
    z2 = -z2 * c["c2_562"]
    z3 = -z3 * c["c1_961"] + z5
    z4 = -z4 * c["c0_390"] + z5
    t0 = t0 + z1 + z3
    t1 = t1 + z2 + z4
    t2 = t2 + z2 + z3
    t3 = t3 + z1 + z4
    half = 1 << (shift - 1)
    outs = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
    return np.stack([(o + half) >> shift for o in outs], axis=-1)


def idct_islow(coef: np.ndarray, quant: np.ndarray) -> np.ndarray:
    """[..., 64] quantised coefficients -> [..., 8, 8] samples (jidctint.c jpeg_idct_islow: columns, then rows, +128, clamp)."""
    x = (coef.astype(np.int64) * quant.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(x, -1, -2), 13 - 2)      # pass 1 runs down the columns
    ws = np.swapaxes(ws, -1, -2)
    px = _idct_1d(ws, 13 + 2 + 3)                       # pass 2 along the rows
    return np.clip(px + 128, 0, 255).astype(np.uint8)


def _planes(info: dict, coefs: list[np.ndarray]) -> list[np.ndarray]:
    frame = info["frame"]
    g = geometry(frame)
    planes = []
    for c, co in zip(frame["comps"], coefs):
        px = idct_islow(co, info["quant"][c["tq"]])  # [by, bx, 8, 8]
        plane = px.transpose(0, 2, 1, 3).reshape(co.shape[0] * 8, co.shape[1] * 8)
        # the samples that exist: jdmaster.c downsampled_width / _height = ceil(image * factor / max factor)
        dw = -(-frame["width"] * c["h"] // g["hmax"])
        dh = -(-frame["height"] * c["v"] // g["vmax"])
        planes.append(plane[:dh, :dw])
    return planes


def _h2v1_fancy(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample: 3/4 nearer + 1/4 farther, alternating rounding; the edge columns copied."""
    x = p.astype(np.int32)
    left = np.concatenate([x[:, :1], x[:, :-1]], axis=1)
    right = np.concatenate([x[:, 1:], x[:, -1:]], axis=1)
    out = np.empty((x.shape[0], 2 * x.shape[1]), np.int32)
    out[:, 0::2] = (3 * x + left + 1) >> 2
    out[:, 1::2] = (3 * x + right + 2) >> 2
    out[:, 0] = x[:, 0]
    out[:, -1] = x[:, -1]
    return out.astype(np.uint8)


def _h2v2_fancy(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: the triangle filter in both directions; the rows above the first / below the last
    are those rows themselves (jdmainct.c context rows)."""
    x = p.astype(np.int32)
    above = np.concatenate([x[:1], x[:-1]], axis=0)
    below = np.concatenate([x[1:], x[-1:]], axis=0)
    out = np.empty((2 * x.shape[0], 2 * x.shape[1]), np.int32)
    for v, other in ((0, above), (1, below)):
        col = 3 * x + other  # "thiscolsum"
        last = np.concatenate([col[:, :1], col[:, :-1]], axis=1)
        nxt = np.concatenate([col[:, 1:], col[:, -1:]], axis=1)
        even = (3 * col + last + 8) >> 4
        odd = (3 * col + nxt + 7) >> 4
        even[:, 0] = (4 * col[:, 0] + 8) >> 4
        odd[:, -1] = (4 * col[:, -1] + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out.astype(np.uint8)


def _ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    """jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: 16-bit fixed point tables, arithmetic right shifts."""
    def fix(v):
        return int(v * 65536 + 0.5)

    x = np.arange(256, dtype=np.int64) - 128
    cr_r = (fix(1.40200) * x + 32768) >> 16
    cb_b = (fix(1.77200) * x + 32768) >> 16
    cr_g = -fix(0.71414) * x
    cb_g = -fix(0.34414) * x + 32768
    yy = y.astype(np.int64)
    r = yy + cr_r[cr]
    g = yy + ((cb_g[cb] + cr_g[cr]) >> 16)
    b = yy + cb_b[cb]
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode(data: bytes) -> np.ndarray:
    """JPEG bytes -> [H, W, 3] uint8 RGB as libjpeg hands it to vtkJPEGReader / Pillow's ``convert("RGB")``."""
    info = parse(data)
    frame = info["frame"]
    h, w = frame["height"], frame["width"]
    planes = _planes(info, decode_coefficients(info))
    if len(planes) == 1:
        return np.repeat(planes[0][:h, :w, None], 3, axis=2)
    y, cb, cr = planes
    c0 = frame["comps"][0]
    if (c0["h"], c0["v"]) == (2, 1):
        # jdsample.c jinit_upsampler: the fancy filters only when downsampled_width > 2, else replication
        up = _h2v1_fancy if cb.shape[1] > 2 else (lambda p: np.repeat(p, 2, axis=1))
        cb, cr = up(cb), up(cr)
    elif (c0["h"], c0["v"]) == (2, 2):
        up = _h2v2_fancy if cb.shape[1] > 2 else (lambda p: np.repeat(np.repeat(p, 2, axis=0), 2, axis=1))
        cb, cr = up(cb), up(cr)
    return _ycc_to_rgb(y[:h, :w], cb[:h, :w], cr[:h, :w])
