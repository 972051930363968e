"""`python -m raytracer_amd <file.ppm> <width> <height>`: the reference binary's command line
(src/main.rs:5-17, src/lib.rs:39-75) with the render running on the GPU."""
import sys

from . import api, scenes


def main(argv):
    if len(argv) < 4:
        sys.stderr.write("Problem parsing arguments: not enough arguments\n")  # src/main.rs:8-11
        return 1
    path, width, height = argv[1], int(argv[2]), int(argv[3])
    rt = api.load()
    # render and Canvas::to_ppm both run on the device; only the text comes back
    text = rt.render_ppm(scenes.build_world(rt), scenes.default_camera(rt, width, height))
    with open(path, "wb") as f:
        f.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
