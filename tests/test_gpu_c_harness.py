"""The C ABI driven from plain C (examples/cornell.c, built by `make -C raytracing_amd/csrc`): a
process with no Python and no torch in it builds the cornell-box description by hand, renders it and
must produce the very pixels the Python host layer gets for the same request."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import abi
from util import scene_bundle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "cornell")


def test_plain_c_program_renders_the_same_pixels(gpu_ctx, tmp_path):
    assert os.path.exists(EXE), "examples/cornell is built by make -C raytracing_amd/csrc"
    W, H, spp = 72, 40, 24
    ppm, raw = str(tmp_path / "c.ppm"), str(tmp_path / "c.bin")
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)  # the program finds librtmi.so through its rpath
    r = subprocess.run([EXE, str(W), str(H), str(spp), ppm, raw], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    c_argb = np.fromfile(raw, dtype=np.uint32).reshape(H, W)
    cfg, flat, _ = scene_bundle("cornell-box")
    from raytracing_amd.core import build_camera
    cam = build_camera(cfg, W / H)
    sc = gpu_ctx.scene(flat)
    _, argb, st = sc.render(cam, abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001))
    sc.close()
    np.testing.assert_array_equal(c_argb, np.asarray(argb, dtype=np.uint32).reshape(H, W))
    with open(ppm, "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == f"{W} {H}\n".encode() and f.readline() == b"255\n"
        body = np.frombuffer(f.read(), dtype=np.uint8).reshape(H, W, 3)
    top = c_argb[H - 1]  # the PPM's first row is the image's top row = y = H-1
    np.testing.assert_array_equal(body[0, :, 0], (top >> 16) & 255)
    assert f"{W * H * spp} samples" in r.stderr


def test_plain_c_program_writes_and_loads_a_scene_file(gpu_ctx, tmp_path):
    """The same program with a scene-file argument: it writes its hand-built description with
    rt_scene_write and creates the world from that file. Its pixels must not change, and the file
    must load through the Python reader into a scene that renders them too."""
    from raytracing_amd import scenefile
    from raytracing_amd.core import build_camera
    W, H, spp = 40, 40, 8
    raw0, raw1, scn = str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), str(tmp_path / "c.rtscn")
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
    for args in ([raw0], [raw1, scn]):
        r = subprocess.run([EXE, str(W), str(H), str(spp), str(tmp_path / "x.ppm")] + args, capture_output=True,
                           text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr
    a, b = np.fromfile(raw0, dtype=np.uint32), np.fromfile(raw1, dtype=np.uint32)
    np.testing.assert_array_equal(a, b)
    flat = scenefile.load(scn)
    assert flat.desc.n_prims == 18 and flat.desc.n_mats == 4
    cfg, _, _ = scene_bundle("cornell-box")
    sc = gpu_ctx.scene(flat)
    _, argb, _ = sc.render(build_camera(cfg, W / H), abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001))
    sc.close()
    np.testing.assert_array_equal(a, np.asarray(argb, dtype=np.uint32).ravel())


@pytest.mark.parametrize("form", ["jpeg", "png", "gif-as-jpg-name"])
def test_plain_c_program_reads_a_picture_file_and_renders_the_earth(gpu_ctx, tmp_path, form):
    """examples/earth.c: (image-texture url) from plain C. The program hands the library the FILE (rt_image_decode picks
    the decoder by content), wraps the texels round the reference's earth and renders it; the pixels must be the ones the
    Python host layer gets from the same texels."""
    from PIL import Image

    from raytracing_amd import lib
    from raytracing_amd.core import build_camera
    from raytracing_amd.flatten import flatten_world
    from raytracing_amd.hittables import sphere
    from raytracing_amd.materials import lambertian
    from raytracing_amd.scenes import scene_config
    from raytracing_amd.texture import ASSET_DIR, image_texture
    exe = os.path.join(ROOT, "examples", "earth")
    assert os.path.exists(exe), "examples/earth is built by make -C raytracing_amd/csrc"
    picture = os.path.join(ASSET_DIR, "earthmap.jpg")
    if form != "jpeg":
        with np.load(os.path.join(ASSET_DIR, "earthmap_rgb8.npz")) as z:
            img = Image.fromarray(z["rgb8"][::4, ::4].copy())
        if form == "png":
            picture = str(tmp_path / "earth.png")
            img.save(picture, "PNG")
        else:       # a palette picture under a name that lies about its format
            picture = str(tmp_path / "earth.jpg")
            img.quantize(256).save(picture, "GIF")
    W, H, spp = 60, 40, 12
    raw = str(tmp_path / "e.bin")
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
    r = subprocess.run([exe, picture, str(W), str(H), str(spp), str(tmp_path / "e.ppm"), raw], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    c_argb = np.fromfile(raw, dtype=np.uint32).reshape(H, W)
    texels = lib.decode_image(open(picture, "rb").read())
    assert f"{texels.shape[1]} x {texels.shape[0]} texels" in r.stderr
    cfg = scene_config("earth")
    sc = gpu_ctx.scene(flatten_world([sphere([0, 0, 0], 2, lambertian(image_texture(texels)))]))
    _, argb, _ = sc.render(build_camera(cfg, W / H), abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001))
    sc.close()
    np.testing.assert_array_equal(c_argb, np.asarray(argb, dtype=np.uint32).reshape(H, W))
    assert len(np.unique(c_argb)) > 50     # a picture, not a flat frame
