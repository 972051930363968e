"""A rank started by raytracer_amd.launch.spawn_ranks (tests/test_distributed_cpu.py): reads its
rank from the environment like bench.py's workers do, assembles a synthetic byte frame from the
ranks' cyclic tiles over gloo and lets rank 0 print one JSON line."""
import hashlib
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    W, H = int(sys.argv[1]), int(sys.argv[2])
    fail_rank = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    rank, world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    if rank == fail_rank:
        sys.exit(7)
    dist.init_process_group("gloo")
    from raytracer_amd import api
    from raytracer_amd.distributed import FrameGather
    host = api.load()
    rows = torch.from_numpy(host.global_rows(H, rank, world_size))
    local = ((rows.view(-1, 1, 1) * 7 + torch.arange(W).view(1, -1, 1) * 3 + torch.arange(3).view(1, 1, -1) * 11) % 256).to(torch.uint8)
    frame = FrameGather(host, W, H, rank, world_size, torch.device("cpu"), dtype=torch.uint8)(local)
    if rank == 0:
        print("not json: something a library printed")
        print(json.dumps({"n": world_size, "sha": hashlib.sha256(frame.numpy().tobytes()).hexdigest()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
