"""The oracle against the reference's own golden render: example_output.png is the 1000x1000 image
the reference's binary produced for its default scene (`cargo run --release example_output.ppm 1000
1000`).  Every pixel of the oracle's render must quantise (Canvas::clamp, canvas.rs:50-60) to the
PNG's 8-bit value.  This is what pins the CPU restatement to the reference."""
import hashlib
import os

import numpy as np
import pytest

from raytracer_amd import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_output.png")


def test_default_scene_matches_reference_png(orc):
    Image = pytest.importorskip("PIL.Image")
    ref = np.asarray(Image.open(GOLDEN).convert("RGB"))
    assert ref.shape == (1000, 1000, 3)
    assert hashlib.sha256(ref.tobytes()).hexdigest() == \
        "49c652e0c3fd325cc0cd9793b6b0f4a3176a2e2d7c367106ebd6bddd6263e4d0"
    img = orc.render(scenes.build_world(orc), scenes.default_camera(orc, 1000, 1000))
    q = np.clip(np.ceil(255.0 * img), 0, 255).astype(np.uint8)
    diff = np.any(q != ref, axis=2)
    assert not diff.any(), "%d pixels differ from the reference render" % int(diff.sum())
    ppm = orc.to_ppm(img)
    assert ppm.startswith(b"P3\n1000 1000\n255\n")
    assert max(len(line) for line in ppm.split(b"\n")) <= 69


def test_faithful_cost_mode_is_bit_identical(orc):
    """The cost-faithful mode (inverse per use, secondary rays per light) only changes the cost."""
    world, cam = scenes.build_world(orc), scenes.default_camera(orc, 64, 48)
    fast = orc.render(world, cam, threads=4)
    orc.set_faithful(True)
    try:
        slow = orc.render(world, cam, threads=4)
    finally:
        orc.set_faithful(False)
    assert np.array_equal(fast, slow)
