"""The earth (scenes.clj:38-42) rendered from the bytes of earthmap.jpg as the library decodes them: the world the
scene constructor builds now holds the library's texels, and the frame equals the one rendered from the array Pillow
made of the same file when the assets were converted."""
import os

import numpy as np
import pytest

from raytracing_amd import abi
from raytracing_amd.core import build_camera, setup
from raytracing_amd.flatten import flatten_world
from raytracing_amd.hittables import sphere
from raytracing_amd.materials import lambertian
from raytracing_amd.texture import ASSET_DIR, image_texture, resource

pytestmark = pytest.mark.gpu


def test_earth_from_decoded_bytes(gpu_ctx):
    cfg, flat = setup("earth", 1)
    cam = build_camera(cfg, 1.5)
    p = abi.make_params(96, 64, 16, 50, cfg["background"], seed=5)
    sc = gpu_ctx.scene(flat)
    rgb, argb, st = sc.render(cam, p)
    sc.close()
    # the same world over the converted array
    with np.load(os.path.join(ASSET_DIR, "earthmap_rgb8.npz")) as z:
        world = [sphere([0, 0, 0], 2, lambertian(image_texture(z["rgb8"])))]
    sc = gpu_ctx.scene(flatten_world(world))
    rgb2, argb2, st2 = sc.render(cam, p)
    sc.close()
    assert np.array_equal(rgb, rgb2) and np.array_equal(argb, argb2) and st["rays"] == st2["rays"]
    assert rgb.max() > 0
    # and the progressive, chroma-subsampled file as a texture
    world = [sphere([0, 0, 0], 2, lambertian(image_texture(resource("stones.jpg"))))]
    sc = gpu_ctx.scene(flatten_world(world))
    rgb3, _, _ = sc.render(cam, p)
    sc.close()
    assert rgb3.max() > 0 and not np.array_equal(rgb3, rgb)


def test_earth_from_a_png_file_and_the_frame_written_as_one(gpu_ctx, tmp_path):
    """The same texels through the other format (image-texture url) can be handed (texture.clj:72): the earth map
    re-written as a PNG file (lossless, so the texels are the decoded JPEG's) gives the frame bit for bit; and the frame
    written by core.main as a PNG decodes to the frame."""
    from PIL import Image

    from raytracing_amd import core, lib
    cfg, flat = setup("earth", 1)
    cam = build_camera(cfg, 1.5)
    p = abi.make_params(96, 64, 16, 50, cfg["background"], seed=5)
    sc = gpu_ctx.scene(flat)
    rgb, argb, st = sc.render(cam, p)
    sc.close()
    with np.load(os.path.join(ASSET_DIR, "earthmap_rgb8.npz")) as z:
        Image.fromarray(z["rgb8"]).save(str(tmp_path / "earthmap.png"), "PNG")
    world = [sphere([0, 0, 0], 2, lambertian(image_texture(str(tmp_path / "earthmap.png"))))]
    sc = gpu_ctx.scene(flatten_world(world))
    rgb2, argb2, st2 = sc.render(cam, p)
    sc.close()
    assert np.array_equal(rgb, rgb2) and np.array_equal(argb, argb2) and st["rays"] == st2["rays"]
    out = str(tmp_path / "frame.png")
    frame, _ = core.main(samples=4, scene="earth", image_width=64, aspect_ratio=1.0, output=out)
    top_first = frame[::-1]
    want = np.stack([(top_first >> 16) & 255, (top_first >> 8) & 255, top_first & 255], axis=-1).astype(np.uint8)
    assert np.array_equal(lib.decode_png(open(out, "rb").read()), want)
