"""rt_image_decode_jpeg, the library's stand-in for (image-texture url)'s ImageIO/read (texture.clj:68-81).

Expected outputs: the reference's three image files as Pillow (libjpeg-turbo, the IJG decoder's defaults)
decoded them when the assets were converted (raytracing_amd/assets/*_rgb8.npz, tools/make_assets.py), and, for
files written here, what Pillow makes of the same bytes. javax.imageio is that same IJG decoder; a difference of
one LSB per texel would be within what two JPEG decoders may differ by — the decoder restates the IJG
reconstruction step for step and the tests find no difference at all. Host code: runs without a GPU."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from raytracing_amd import abi, lib
from raytracing_amd.texture import ASSET_DIR, image_texture, resource

FILES = [("earthmap.jpg", (512, 1024)), ("stones.jpg", (513, 512)), ("map2.png", (360, 700))]


@pytest.mark.parametrize("name,shape", FILES)
def test_reference_assets_decode_to_the_converted_arrays(name, shape):
    """earthmap.jpg: baseline, 4:4:4; stones.jpg: progressive with successive approximation, 4:2:0, odd height;
    map2.png: a baseline JPEG whose width is not a multiple of eight."""
    data = open(os.path.join(ASSET_DIR, name), "rb").read()
    got = lib.decode_jpeg(data)
    with np.load(os.path.join(ASSET_DIR, os.path.splitext(name)[0] + "_rgb8.npz")) as z:
        want = z["rgb8"]
    assert got.shape == want.shape == shape + (3,)
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1, f"{name}: a texel differs by {diff.max()}"
    assert diff.max() == 0, f"{name}: {int((diff.max(axis=2) > 0).sum())} texels differ by one step"
    tex = image_texture(resource(name))       # the path the scene constructors take
    assert np.array_equal(tex.data, want) and (tex.height, tex.width) == shape


def _picture(w, h, seed):
    """Smooth colour ramps with edges and some noise: every coefficient band gets used."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), (x * 255.0 / max(w - 1, 1)), (y * 255.0 / max(h - 1, 1))], axis=2)
    img[(x // 8 + y // 8) % 5 == 0] = (250, 10, 40)
    img += rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # (w, h, mode, save keywords)
    (64, 48, "RGB", dict(quality=90, subsampling=0)),
    (65, 47, "RGB", dict(quality=75, subsampling=1)),                       # 4:2:2, ragged edges
    (67, 45, "RGB", dict(quality=75, subsampling=2)),                       # 4:2:0, ragged edges
    (67, 45, "RGB", dict(quality=35, subsampling=2, progressive=True)),     # refinement scans at low quality
    (33, 70, "RGB", dict(quality=95, subsampling=1, progressive=True)),
    (40, 40, "RGB", dict(quality=100, subsampling=0, progressive=True)),
    (31, 17, "L", dict(quality=80)),
    (31, 17, "L", dict(quality=60, progressive=True)),
    (1, 1, "RGB", dict(quality=90, subsampling=2)),
    (2, 3, "RGB", dict(quality=90, subsampling=2, progressive=True)),
    (200, 120, "RGB", dict(quality=85, subsampling=2, restart_marker_blocks=3)),
    (200, 120, "RGB", dict(quality=85, subsampling=0, restart_marker_rows=1, progressive=True)),
    (96, 64, "RGB", dict(quality=50, subsampling=2, optimize=True)),        # its own Huffman tables
]


@pytest.mark.parametrize("w,h,mode,kw", CASES)
def test_files_written_here_decode_as_pillow_decodes_them(w, h, mode, kw):
    img = _picture(w, h, seed=w * 1000 + h)
    pil = Image.fromarray(img if mode == "RGB" else img[:, :, 0], mode)
    buf = io.BytesIO()
    pil.save(buf, "JPEG", **kw)
    data = buf.getvalue()
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    got = lib.decode_jpeg(data)
    assert got.shape == (h, w, 3)
    assert np.array_equal(got, want), f"max difference {np.abs(got.astype(int) - want.astype(int)).max()}"


def test_what_is_not_read_is_refused_with_a_reason():
    data = open(os.path.join(ASSET_DIR, "earthmap.jpg"), "rb").read()
    for bad in (b"", b"\x89PNG\r\n\x1a\n" + bytes(64), data[:2], data[:300]):
        with pytest.raises(lib.RtError) as e:
            lib.decode_jpeg(bad)
        assert e.value.code == abi.RT_ERR_ARG and "JPEG" in str(e.value)
    buf = io.BytesIO()
    Image.fromarray(_picture(16, 16, 1)).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(lib.RtError) as e:
        lib.decode_jpeg(buf.getvalue())
    assert e.value.code == abi.RT_ERR_UNSUPPORTED and "four-component" in str(e.value)
    # a file cut off in the middle of its picture still gives the rows that came through (as the IJG decoder does)
    part = lib.decode_jpeg(data[: len(data) // 2])
    with np.load(os.path.join(ASSET_DIR, "earthmap_rgb8.npz")) as z:
        assert np.array_equal(part[:100], z["rgb8"][:100])
    # a buffer that is too small is refused, nothing written
    import ctypes as C
    L = lib.load()
    w, h = C.c_int32(), C.c_int32()
    small = np.full(100, 7, np.uint8)
    assert L.rt_image_decode_jpeg(data, len(data), C.byref(w), C.byref(h), small.ctypes.data, small.nbytes) == abi.RT_ERR_ARG
    assert (w.value, h.value) == (1024, 512) and (small == 7).all()


def _mutations(data, rng, n):
    """Truncations, overwritten bytes (anywhere and in the headers), planted markers, pieces cut out."""
    for _ in range(n):
        d = bytearray(data)
        mode = int(rng.integers(0, 5))
        if mode == 0:
            d = d[: int(rng.integers(0, len(d) + 1))]
        elif mode == 1:
            for _ in range(int(rng.integers(1, 9))):
                d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        elif mode == 2:
            for _ in range(int(rng.integers(1, 5))):
                d[int(rng.integers(0, min(len(d), 700)))] = int(rng.integers(0, 256))
        elif mode == 3:
            q = int(rng.integers(0, len(d) - 1))
            d[q], d[q + 1] = 0xFF, int(rng.integers(0xC0, 0x100))
        else:
            q = int(rng.integers(0, len(d)))
            del d[q: q + int(rng.integers(1, 65))]
        yield bytes(d)


def test_a_huffman_table_that_is_not_a_prefix_code_is_refused():
    """Three codes of length one cannot exist: such a table used to overrun the decoder's nine-bit lookup table."""
    buf = io.BytesIO()
    Image.fromarray(_picture(32, 32, 5)).save(buf, "JPEG", quality=80)
    data = bytearray(buf.getvalue())
    at = data.index(b"\xff\xc4")          # the first DHT segment: marker, length, class | id, sixteen counts
    data[at + 5] = 3
    with pytest.raises(lib.RtError) as e:
        lib.decode_jpeg(bytes(data))
    assert e.value.code == abi.RT_ERR_ARG and "prefix code" in str(e.value)


@pytest.mark.parametrize("kw", [dict(quality=85, subsampling=2), dict(quality=40, subsampling=1, progressive=True),
                                dict(quality=85, subsampling=0, restart_marker_blocks=2)])
def test_mutated_files_never_take_the_library_down(kw):
    """Every mutation either decodes to a picture of the size its header names or is refused with a code and a
    message (tools/fuzz/jpeg_fuzz.cpp runs the same mutations under AddressSanitizer)."""
    buf = io.BytesIO()
    Image.fromarray(_picture(48, 40, 9)).save(buf, "JPEG", **kw)
    rng = np.random.default_rng(len(buf.getvalue()))
    decoded = refused = 0
    for bad in _mutations(buf.getvalue(), rng, 400):
        try:
            got = lib.decode_jpeg(bad)
            assert got.ndim == 3 and got.shape[2] == 3 and got.shape[0] > 0 and got.shape[1] > 0
            decoded += 1
        except lib.RtError as e:
            assert e.code in (abi.RT_ERR_ARG, abi.RT_ERR_UNSUPPORTED, abi.RT_ERR_OOM) and str(e)
            refused += 1
    assert decoded + refused == 400 and refused > 0
