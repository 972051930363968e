"""The GPU path against the reference's own golden render: `python -m raytracer_amd out.ppm 1000 1000`
is the reference's `cargo run --release example_output.ppm 1000 1000`; the PPM it writes must decode
to the pixels of the reference's example_output.png (at most 0.01 % of pixels off by one step of
1/255, the tolerance for the device's pow/cos; measured: none)."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_output.png")


def test_command_line_reproduces_the_reference_image(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    from raytracer_amd import __main__ as cli
    ref = np.asarray(Image.open(GOLDEN).convert("RGB"))
    assert hashlib.sha256(ref.tobytes()).hexdigest() == \
        "49c652e0c3fd325cc0cd9793b6b0f4a3176a2e2d7c367106ebd6bddd6263e4d0"
    out = tmp_path / "example_output.ppm"
    assert cli.main(["raytracer_amd", str(out), "1000", "1000"]) == 0
    tokens = out.read_bytes().split()
    assert tokens[:4] == [b"P3", b"1000", b"1000", b"255"]
    got = np.array(tokens[4:], dtype=np.int64).reshape(1000, 1000, 3)
    diff = np.abs(got - ref.astype(np.int64))
    assert diff.max() <= 1
    assert np.any(diff != 0, axis=2).mean() <= 1e-4
    assert cli.main(["raytracer_amd", str(out)]) == 1   # not enough arguments (src/main.rs:8-11)
