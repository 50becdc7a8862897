"""rt_image_decode_png and rt_image_decode, the library's stand-ins for (image-texture url)'s ImageIO/read
(texture.clj:68-81) on PNG files.

PNG is lossless, so the expected texels are the samples that went into the file: files Pillow writes are compared
with the arrays they were made from and with Pillow's own reading of them; files written here (every colour type
and bit depth, Adam7 interlacing, each row filter, stored / fixed / dynamic deflate blocks), which Pillow cannot all
write, with their samples, and with Pillow's reading where Pillow converts the type the same way. Host code: runs
without a GPU."""
import io
import os
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from raytracing_amd import abi, core, lib
from raytracing_amd.texture import ASSET_DIR, image_texture

ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _pack_rows(samples, depth):
    """(h, w, channels) integer samples -> h rows of packed bytes (16 bits big-endian, sub-byte depths from the high bits)."""
    h, w, c = samples.shape
    if depth == 16:
        return [samples[y].astype(">u2").tobytes() for y in range(h)]
    if depth == 8:
        return [samples[y].astype(np.uint8).tobytes() for y in range(h)]
    rows = []
    for y in range(h):
        bits = np.zeros(((w * depth + 7) // 8) * 8, np.uint8)
        for b in range(depth):
            bits[np.arange(w) * depth + b] = (samples[y, :, 0] >> (depth - 1 - b)) & 1
        rows.append(np.packbits(bits).tobytes())
    return rows


def _filter_row(kind, cur, prev, bpp):
    out = bytearray(len(cur))
    for i in range(len(cur)):
        a = cur[i - bpp] if i >= bpp else 0
        b = prev[i] if prev is not None else 0
        c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
        if kind == 0:
            pred = 0
        elif kind == 1:
            pred = a
        elif kind == 2:
            pred = b
        elif kind == 3:
            pred = (a + b) // 2
        else:
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        out[i] = (cur[i] - pred) & 0xFF
    return bytes(out)


def make_png(samples, ctype, depth, interlace=0, filters=(0, 1, 2, 3, 4), level=6, palette=None, extra=(), split=None):
    """A PNG file of the given samples ((h, w, channels) integers below 2^depth), written from the specification."""
    h, w, c = samples.shape
    assert c == CHANNELS[ctype]
    bpp = max(1, c * depth // 8)
    raw = bytearray()
    k = 0
    for (x0, y0, dx, dy) in (ADAM7 if interlace else [(0, 0, 1, 1)]):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for row in _pack_rows(sub, depth):
            f = filters[k % len(filters)]
            k += 1
            raw.append(f)
            raw += _filter_row(f, row, prev, bpp)
            prev = row
    z = zlib.compress(bytes(raw), level)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    for kind, body in extra:
        out += _chunk(kind, body)
    if palette is not None:
        out += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    pieces = [z] if not split else [z[i:i + split] for i in range(0, len(z), split)]
    for piece in pieces:
        out += _chunk(b"IDAT", piece)
    return out + _chunk(b"IEND", b"")


def expected_rgb(samples, ctype, depth, palette=None):
    s = samples.astype(np.int64)
    if ctype == 3:
        return np.asarray(palette, np.uint8)[s[:, :, 0]]
    if depth == 16:
        s = (s * 255 + 32767) // 65535
    elif depth < 8:
        s = s * (255 // ((1 << depth) - 1))
    if CHANNELS[ctype] <= 2:
        return np.repeat(s[:, :, :1], 3, axis=2).astype(np.uint8)
    return s[:, :, :3].astype(np.uint8)


def _samples(w, h, c, depth, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    top = (1 << depth) - 1
    base = np.stack([(x * 7 + y * 3 + 11 * k) % (top + 1) for k in range(c)], axis=2)       # ramps: the filters and LZ77 have something to find
    noisy = rng.integers(0, top + 1, (h, w, c))
    return np.where(rng.random((h, w, 1)) < 0.8, base, noisy).astype(np.int64)


ALL_TYPES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [0, 1])
@pytest.mark.parametrize("ctype,depth", ALL_TYPES)
def test_every_colour_type_and_bit_depth(ctype, depth, interlace):
    for (w, h) in [(1, 1), (3, 2), (8, 8), (13, 9), (33, 5)]:      # Adam7 passes that are empty, ragged last bytes
        s = _samples(w, h, CHANNELS[ctype], depth, seed=ctype * 100 + depth + w)
        pal = None
        if ctype == 3:
            pal = np.random.default_rng(depth).integers(0, 256, (1 << depth, 3))
        data = make_png(s, ctype, depth, interlace, palette=pal)
        got = lib.decode_png(data)
        assert got.shape == (h, w, 3)
        assert np.array_equal(got, expected_rgb(s, ctype, depth, pal)), (ctype, depth, interlace, w, h)
        if depth == 8 or ctype in (0, 3):       # where Pillow's conversion to RGB is the same map: it reads the file the same way
            if not (ctype == 0 and depth == 16):
                assert np.array_equal(got, np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))), (ctype, depth, interlace, w, h)
        assert np.array_equal(lib.decode_image(data), got)


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("level", [0, 1, 9])
def test_each_row_filter_and_each_kind_of_deflate_block(filt, level):
    """Level 0 writes stored blocks, a few bytes at level 1 a fixed-code block, level 9 dynamic codes; 70 000 bytes at
    level 0 need more than one stored block."""
    for (w, h, ctype, depth) in [(5, 4, 2, 8), (160, 150, 2, 8), (61, 40, 6, 16), (90, 33, 0, 4)]:
        s = _samples(w, h, CHANNELS[ctype], depth, seed=filt * 10 + level)
        data = make_png(s, ctype, depth, 0, filters=(filt,), level=level, split=997 if level == 9 else None)
        assert np.array_equal(lib.decode_png(data), expected_rgb(s, ctype, depth))


def test_codes_longer_than_the_lookup_table():
    """Sample values with a geometric distribution: the rare ones get Huffman codes of 11 to 15 bits, which the decoder
    finds bit by bit from the code-length counts instead of by its ten-bit table."""
    rng = np.random.default_rng(77)
    s = np.minimum(rng.geometric(0.35, (300, 400, 1)) - 1, 255).astype(np.int64)
    data = make_png(s, 0, 8, filters=(0,), level=9)
    assert np.array_equal(lib.decode_png(data), expected_rgb(s, 0, 8))
    assert np.array_equal(lib.decode_png(data), np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


@pytest.mark.parametrize("mode,kw", [("RGB", dict()), ("RGB", dict(optimize=True)), ("RGB", dict(compress_level=0)),
                                     ("RGBA", dict()), ("L", dict()), ("LA", dict()), ("P", dict()), ("1", dict()),
                                     ("RGB", dict(compress_level=1))])
def test_files_pillow_writes_decode_to_what_went_in(mode, kw):
    rng = np.random.default_rng(len(mode) + len(kw))
    w, h = 157, 83
    y, x = np.mgrid[0:h, 0:w]
    rgb = np.stack([(x * 255 // (w - 1)), (y * 255 // (h - 1)), (x + y) % 256], axis=2).astype(np.uint8)
    rgb[rng.random((h, w)) < 0.1] = (255, 0, 128)
    img = Image.fromarray(rgb)
    if mode == "P":
        img = img.quantize(64)
    elif mode != "RGB":
        img = img.convert(mode)
    buf = io.BytesIO()
    img.save(buf, "PNG", **kw)
    got = lib.decode_png(buf.getvalue())
    assert np.array_equal(got, np.asarray(img.convert("RGB")))
    if mode == "RGB":
        assert np.array_equal(got, rgb)     # the bytes the reference's raster read gives for such a file (texture.clj:73-77)


def test_a_texture_from_a_png_file_is_the_texture_from_its_texels(tmp_path):
    with np.load(os.path.join(ASSET_DIR, "stones_rgb8.npz")) as z:
        rgb = z["rgb8"]
    path = str(tmp_path / "stones.png")
    Image.fromarray(rgb).save(path, "PNG")
    tex = image_texture(path)
    assert (tex.height, tex.width) == rgb.shape[:2] and np.array_equal(tex.data, rgb)
    # ... and by content, not by name: the reference's map2.png is a JPEG (ImageIO/read sniffs, texture.clj:72)
    jpeg = open(os.path.join(ASSET_DIR, "map2.png"), "rb").read()
    assert np.array_equal(lib.decode_image(jpeg), lib.decode_jpeg(jpeg))
    odd = str(tmp_path / "really_a_png.jpg")
    Image.fromarray(rgb[:40, :50]).save(odd, "PNG")
    assert np.array_equal(image_texture(odd).data, rgb[:40, :50])
    with pytest.raises(lib.RtError) as e:
        lib.decode_image(b"II*\0" + bytes(100))        # a TIFF by its first bytes
    assert e.value.code == abi.RT_ERR_UNSUPPORTED and "not a JPEG, PNG, GIF or BMP" in str(e.value)


def test_the_frame_writer_round_trips(tmp_path):
    rng = np.random.default_rng(3)
    argb = (rng.integers(0, 1 << 24, (37, 53)).astype(np.uint32)) | np.uint32(0xFF000000)
    path = str(tmp_path / "frame.png")
    core.write_png(path, argb)
    got = lib.decode_png(open(path, "rb").read())
    top_first = argb[::-1]
    want = np.stack([(top_first >> 16) & 255, (top_first >> 8) & 255, top_first & 255], axis=-1).astype(np.uint8)
    assert np.array_equal(got, want)
    assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), want)


def test_ancillary_chunks_are_skipped_and_unknown_critical_ones_refused():
    s = _samples(9, 7, 3, 8, 1)
    extra = [(b"gAMA", struct.pack(">I", 45455)), (b"tEXt", b"Comment\0hello"), (b"tRNS", bytes(6))]
    assert np.array_equal(lib.decode_png(make_png(s, 2, 8, extra=extra)), expected_rgb(s, 2, 8))
    with pytest.raises(lib.RtError) as e:
        lib.decode_png(make_png(s, 2, 8, extra=[(b"FOOB", b"x")]))
    assert e.value.code == abi.RT_ERR_ARG and "critical chunk" in str(e.value)


def _patched(data, kind, edit):
    """The file with the body of its first chunk of that kind edited and the chunk's CRC made right again."""
    at = data.index(kind) - 4
    n = struct.unpack(">I", data[at:at + 4])[0]
    body = bytearray(data[at + 8:at + 8 + n])
    edit(body)
    return data[:at] + _chunk(kind, bytes(body)) + data[at + 12 + n:]


def test_what_is_damaged_is_refused_with_a_reason():
    s = _samples(40, 30, 3, 8, 2)
    good = make_png(s, 2, 8, level=9)
    assert np.array_equal(lib.decode_png(good), expected_rgb(s, 2, 8))

    def refused(data, code, word):
        with pytest.raises(lib.RtError) as e:
            lib.decode_png(data)
        assert e.value.code == code and word in str(e.value), str(e.value)

    refused(b"", abi.RT_ERR_ARG, "not a PNG")
    refused(open(os.path.join(ASSET_DIR, "earthmap.jpg"), "rb").read()[:4000], abi.RT_ERR_ARG, "not a PNG")
    refused(good[:8], abi.RT_ERR_ARG, "ends inside a chunk")
    refused(good[:len(good) // 2], abi.RT_ERR_ARG, "ends inside a chunk")
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x10
    refused(bytes(flipped), abi.RT_ERR_ARG, "checksum")                     # the chunk's CRC
    refused(_patched(good, b"IDAT", lambda b: b.__setitem__(len(b) - 1, b[-1] ^ 1)), abi.RT_ERR_ARG, "checksum of the image data")  # Adler-32
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(8, 3)), abi.RT_ERR_ARG, "bit depth")
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(9, 5)), abi.RT_ERR_ARG, "colour type")
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(12, 2)), abi.RT_ERR_ARG, "interlace")
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(slice(0, 4), bytes(4))), abi.RT_ERR_ARG, "without size")
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(slice(4, 8), struct.pack(">I", 31))), abi.RT_ERR_ARG, "less image data")   # one row too many
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(slice(4, 8), struct.pack(">I", 29))), abi.RT_ERR_ARG, "more image data")
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(slice(0, 8), struct.pack(">II", 1 << 20, 1 << 20))), abi.RT_ERR_UNSUPPORTED, "2^28")
    refused(_patched(good, b"IHDR", lambda b: b.__setitem__(slice(0, 8), struct.pack(">II", 16000, 16000))), abi.RT_ERR_ARG, "less image data")  # refused before 768 MB are allocated
    # a row filter that does not exist
    raw = bytearray(zlib.decompress(good[good.index(b"IDAT") + 4: good.index(b"IEND") - 8]))
    raw[(1 + 40 * 3) * 5] = 9
    broken = good[:good.index(b"IDAT") - 4] + _chunk(b"IDAT", zlib.compress(bytes(raw))) + _chunk(b"IEND", b"")
    refused(broken, abi.RT_ERR_ARG, "row filter")
    # a palette image without its palette, and one whose pixels point past it
    idx = _samples(6, 6, 1, 8, 4) % 200
    refused(make_png(idx, 3, 8), abi.RT_ERR_ARG, "without its palette")
    refused(make_png(idx, 3, 8, palette=np.zeros((int(idx.max()), 3))), abi.RT_ERR_ARG, "palette entry")
    # a buffer that is too small is refused, nothing written
    import ctypes as C
    L = lib.load()
    w, h = C.c_int32(), C.c_int32()
    small = np.full(100, 7, np.uint8)
    assert L.rt_image_decode_png(good, len(good), C.byref(w), C.byref(h), small.ctypes.data, small.nbytes) == abi.RT_ERR_ARG
    assert (w.value, h.value) == (40, 30) and (small == 7).all()


def test_deflate_streams_that_break_the_rules_are_refused():
    """Hand-made zlib streams behind a valid header: a reserved block type, a stored block whose lengths disagree, a
    distance before the start of the data, code lengths that are not a prefix code."""
    head = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 1, 8, 0, 0, 0, 0))       # 2 x 1 grey: three raw bytes

    def png(z):
        return head + _chunk(b"IDAT", z) + _chunk(b"IEND", b"")

    def bits(*fields):
        """(value, width) fields packed from the low bit of each byte up, as deflate packs everything but Huffman codes;
        (code, -width) is a Huffman code, which goes in from its high bit."""
        seq = []
        for v, n in fields:
            seq += [(v >> i) & 1 for i in (range(n) if n > 0 else range(-n - 1, -1, -1))]
        seq += [0] * (-len(seq) % 8)
        return bytes(sum(b << i for i, b in enumerate(seq[k:k + 8])) for k in range(0, len(seq), 8))

    raw = bytes([0, 10, 200])
    ok = b"\x78\x01" + b"\x01\x03\x00\xfc\xff" + raw + struct.pack(">I", zlib.adler32(raw))
    assert lib.decode_png(png(ok)).tolist() == [[[10, 10, 10], [200, 200, 200]]]
    # the same three bytes from a fixed-code block: literals 0 and 10 (8-bit codes 0x30 + value), 200 (9-bit code 0x190 + 56), end (7 zeros)
    fixed = b"\x78\x01" + bits((1, 1), (1, 2), (0x30, -8), (0x3A, -8), (0x190 + 56, -9), (0, -7)) + struct.pack(">I", zlib.adler32(raw))
    assert lib.decode_png(png(fixed)).tolist() == [[[10, 10, 10], [200, 200, 200]]]
    for z, word in [(b"\x78\x01" + b"\x07" + bytes(8), "reserved type"),
                    (b"\x78\x01" + b"\x01\x03\x00\xfc\xfe" + raw + bytes(4), "length check"),
                    # fixed block: literal 0, then length 3 (code 257: 0000001) at distance 2 (code 1) with one byte out
                    (b"\x78\x01" + bits((1, 1), (1, 2), (0x30, -8), (1, -7), (1, -5)) + bytes(8), "before the start"),
                    # dynamic block: 257 + 1 codes announced, nineteen code-length codes, all of length one
                    (b"\x78\x01" + bits((1, 1), (2, 2), (0, 5), (0, 5), (15, 4), *[(1, 3)] * 19) + bytes(8), "prefix code"),
                    # dynamic block whose code-length code is two codes of length one; 'repeat the previous length' comes first
                    (b"\x78\x01" + bits((1, 1), (2, 2), (0, 5), (0, 5), (0, 4), (1, 3), (0, 3), (0, 3), (1, 3), (1, -1)) + bytes(8), "repeat with no"),
                    (b"\x78\x21" + bytes(12), "not a zlib stream"), (b"\x79\x00" + bytes(12), "not a zlib stream"),
                    (b"\x78\x20" + bytes(12), "preset dictionary")]:
        with pytest.raises(lib.RtError) as e:
            lib.decode_png(png(z))
        assert e.value.code == abi.RT_ERR_ARG and word in str(e.value), (word, str(e.value))


def _mutations(data, rng, n):
    for _ in range(n):
        d = bytearray(data)
        mode = int(rng.integers(0, 5))
        if mode == 0:
            d = d[: int(rng.integers(0, len(d) + 1))]
        elif mode == 1:
            for _ in range(int(rng.integers(1, 9))):
                d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        elif mode == 2:     # damage inside a chunk with the chunk's CRC made right again: past the first gate
            kind = [b"IHDR", b"IDAT", b"IDAT", b"IDAT"][int(rng.integers(0, 4))]

            def edit(b):
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            d = bytearray(_patched(bytes(d), kind, edit))
        elif mode == 3:
            def cut(b):
                q = int(rng.integers(0, len(b)))
                del b[q: q + int(rng.integers(1, 33))]
            d = bytearray(_patched(bytes(d), b"IDAT", cut))
        else:
            q = int(rng.integers(0, len(d)))
            del d[q: q + int(rng.integers(1, 65))]
        yield bytes(d)


@pytest.mark.parametrize("ctype,depth,interlace,level", [(2, 8, 0, 9), (6, 16, 1, 6), (3, 4, 1, 9), (0, 1, 0, 1), (2, 8, 0, 0)])
def test_mutated_files_never_take_the_library_down(ctype, depth, interlace, level):
    """Every mutation either decodes to a picture of the size its header names or is refused with a code and a
    message (tools/fuzz/png_fuzz.cpp runs such mutations under AddressSanitizer)."""
    s = _samples(48, 40, CHANNELS[ctype], depth, 9)
    pal = np.random.default_rng(1).integers(0, 256, (1 << depth, 3)) if ctype == 3 else None
    data = make_png(s, ctype, depth, interlace, level=level, palette=pal)
    rng = np.random.default_rng(len(data))
    decoded = refused = 0
    for bad in _mutations(data, rng, 400):
        try:
            got = lib.decode_png(bad)
            assert got.ndim == 3 and got.shape[2] == 3 and got.shape[0] > 0 and got.shape[1] > 0
            decoded += 1
        except lib.RtError as e:
            assert e.code in (abi.RT_ERR_ARG, abi.RT_ERR_UNSUPPORTED, abi.RT_ERR_OOM) and str(e)
            refused += 1
    assert decoded + refused == 400 and refused > 0
