"""How fast can the GPU start workgroups of the render kernel?  Renders frames in which every tile is
trivial (the camera looks away from the scene: every primary ray misses everything but still walks
the top-level records) and prints the kernel time per frame and per workgroup."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402


def main():
    host = api.load()
    M = host.Matrix
    for W, H in ((1920, 1080), (3840, 2160), (7680, 4320)):
        world = scenes.build_world(host)
        cam = host.Camera(H, W, 0.785, M.view_transform(host.point(-6, 6, -10), host.point(-60, 60, -100),
                                                        host.vector(0, 1, 0)))
        for _ in range(3):
            img, st = host.render(world, cam, stats=True)
        tiles = ((W + 7) // 8) * ((H + 7) // 8)
        print("%dx%d: %d workgroups, kernel %.3f ms, %.1f ns per workgroup, mean colour %.3f, rays %d"
              % (W, H, tiles, st.kernel_ms, st.kernel_ms * 1e6 / tiles, float(img.mean()), st.total_rays()))
        host.reset()


if __name__ == "__main__":
    main()
