"""Wall time per call of the entry points that hand the frame to HOST memory (bench.py's e2e block alone).
usage (GPU box): python tools/e2e_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from raytracer_amd import api  # noqa: E402

host = api.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
out = bench.e2e_block(host, torch, dev, 0)
print(json.dumps(out))
