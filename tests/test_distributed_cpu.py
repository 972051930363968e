"""The multi-GPU frame assembly (raytracer_amd/distributed.py) on CPU: two, three and eight gloo ranks each
produce their cyclic row tiles in compact order (the oracle stands in for the GPU renderer here) and
rank 0 must reassemble exactly the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, W, H, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from oracle import binding
        from raytracer_amd import api, scenes
        from raytracer_amd.distributed import FrameGather
        host = api.load()   # row-partition arithmetic comes from the shipped library (no GPU needed)
        orc = binding.load()
        world, cam = scenes.build_world(orc), scenes.default_camera(orc, W, H)
        rows = host.global_rows(H, rank, world_size)
        local = np.zeros((len(rows), W, 3))
        for k in range(0, len(rows), 8):          # one oracle call per owned tile
            r0 = int(rows[k])
            r1 = int(rows[min(k + 7, len(rows) - 1)]) + 1
            local[k:k + (r1 - r0)] = orc.render(world, cam, threads=2, rows=(r0, r1))[r0:r1]
        gather = FrameGather(host, W, H, rank, world_size, torch.device("cpu"))
        frame = gather(torch.from_numpy(local))
        if rank == 0:
            full = orc.render(world, cam, threads=2)
            np.save(out_path, np.stack([frame.numpy(), full]))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world_size,size", [(2, (40, 37)), (3, (24, 50)), (8, (16, 70))])
def test_gather_reassembles_frame(tmp_path, world_size, size):
    W, H = size
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(world_size, _free_port(), W, H, out), nprocs=world_size, join=True)
    frame, full = np.load(out)
    assert np.array_equal(frame, full)


def _worker_u8(rank, world_size, port, W, H, out_path, mode="p2p"):
    """Byte frames through OverlappedGather (its CPU form): three frames in a row, each different, so
    that a buffer mix-up between consecutive frames would show."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from raytracer_amd import api
        from raytracer_amd.distributed import FrameGather, OverlappedGather
        host = api.load()
        rows = host.global_rows(H, rank, world_size)
        gather = FrameGather(host, W, H, rank, world_size, torch.device("cpu"), dtype=torch.uint8, mode=mode)
        bufs = [torch.empty((len(rows), W, 3), dtype=torch.uint8) for _ in range(2)]
        pipe = OverlappedGather(gather, bufs)
        frames = []
        for k in range(3):
            def render(buf, slot, k=k):
                # pixel (x, y, c) of frame k is a fixed function of its frame coordinates
                yy = torch.from_numpy(rows).view(-1, 1, 1)
                xx = torch.arange(W).view(1, -1, 1)
                cc = torch.arange(3).view(1, 1, -1)
                buf.copy_(((yy * 7 + xx * 3 + cc * 11 + k * 5) % 256).to(torch.uint8))
            frame = pipe.step(render)
            pipe.wait()
            if rank == 0:
                frames.append(frame.numpy().copy())
            else:
                assert frame is None
        if rank == 0:
            np.save(out_path, np.stack(frames))
    finally:
        dist.destroy_process_group()


# (8 ranks: the size of the machine the scaling run uses; 8-row tiles of a 67-row frame leave rank 0 with
# two tiles, ranks 1..7 with one and the last of them short; 13 rows leave six ranks with nothing)
@pytest.mark.parametrize("mode", ["p2p", "collective"])
@pytest.mark.parametrize("world_size,size", [(2, (16, 37)), (3, (12, 50)), (8, (12, 67)), (8, (9, 13))])
def test_overlapped_byte_gather(tmp_path, world_size, size, mode):
    W, H = size
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker_u8, args=(world_size, _free_port(), W, H, out, mode), nprocs=world_size, join=True)
    frames = np.load(out)
    yy, xx, cc = np.meshgrid(np.arange(H), np.arange(W), np.arange(3), indexing="ij")
    for k in range(3):
        assert np.array_equal(frames[k], ((yy * 7 + xx * 3 + cc * 11 + k * 5) % 256).astype(np.uint8))


@pytest.mark.parametrize("world_size", [2, 3, 8])
def test_spawned_ranks_assemble_a_frame(world_size):
    """bench.py --gpus N without a launcher starts its own ranks with raytracer_amd.launch.spawn_ranks:
    fresh interpreters that find RANK / WORLD_SIZE / MASTER_* in their environment.  Here the ranks
    are CPU processes on gloo."""
    import hashlib
    import json
    from raytracer_amd.launch import spawn_ranks
    W, H = 20, 45
    status, out = spawn_ranks([os.path.join(ROOT, "tests", "rank_worker.py"), str(W), str(H)], world_size, timeout=240)
    assert status == 0
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    yy, xx, cc = np.meshgrid(np.arange(H), np.arange(W), np.arange(3), indexing="ij")
    want = ((yy * 7 + xx * 3 + cc * 11) % 256).astype(np.uint8)
    assert got == {"n": world_size, "sha": hashlib.sha256(want.tobytes()).hexdigest()}


def test_spawned_ranks_report_a_failing_rank():
    from raytracer_amd.launch import spawn_ranks
    status, out = spawn_ranks([os.path.join(ROOT, "tests", "rank_worker.py"), "8", "8", "1"], 2, timeout=120)
    assert status != 0
