"""Multi-GPU assembly of a frame rendered in cyclic row tiles (one process per GPU).

Tile t (8 rows) belongs to rank t % world_size, which mirrors the reference's cyclic pixel
partition (src/lib.rs:94-111); rendering needs no communication.  The only exchange step is the one
the reference does through its channel (src/lib.rs:102-104, 121-126): every rank sends its compact
tile buffer to rank 0, which places the rows in frame order.  With the nccl backend that is a grouped
send/recv over RCCL; each peer uses its own xGMI link to rank 0.  The same code runs on CPU tensors
with gloo, which is how it is tested without GPUs.
"""
import numpy as np
import torch
import torch.distributed as dist


class FrameGather:
    def __init__(self, host, width, height, rank, world_size, device, dst=0):
        self.rank, self.world_size, self.dst = rank, world_size, dst
        self.width, self.height = width, height
        self.rows = [host.local_rows(height, r, world_size) for r in range(world_size)]
        self.frame = None
        self.parts = None
        self.index = None
        if rank == dst:
            self.frame = torch.empty((height, width, 3), dtype=torch.float64, device=device)
            self.parts = [None if r == dst else torch.empty((self.rows[r], width, 3), dtype=torch.float64, device=device)
                          for r in range(world_size)]
            self.index = [torch.from_numpy(np.asarray(host.global_rows(height, r, world_size), dtype=np.int64)).to(device)
                          for r in range(world_size)]

    def __call__(self, local):
        """local: this rank's (rows, W, 3) float64 tensor.  Returns the full frame on rank dst."""
        assert tuple(local.shape) == (self.rows[self.rank], self.width, 3)
        if self.world_size > 1 and local.is_cuda and dist.get_backend() == "gloo":
            return self._staged(local)  # rehearsal only: gloo moves host memory
        if self.world_size > 1:
            if self.rank == self.dst:
                ops = [dist.P2POp(dist.irecv, self.parts[r], r) for r in range(self.world_size)
                       if r != self.dst and self.rows[r] > 0]
            else:
                ops = [dist.P2POp(dist.isend, local, self.dst)] if self.rows[self.rank] > 0 else []
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        return self._place(local)

    def _staged(self, local):
        if self.rank == self.dst:
            for r in range(self.world_size):
                if r != self.dst and self.rows[r] > 0:
                    tmp = torch.empty(self.parts[r].shape, dtype=torch.float64)
                    dist.recv(tmp, r)
                    self.parts[r].copy_(tmp)
        elif self.rows[self.rank] > 0:
            dist.send(local.cpu(), self.dst)
        return self._place(local)

    def _place(self, local):
        if self.rank != self.dst:
            return None
        for r in range(self.world_size):
            part = local if r == self.dst else self.parts[r]
            if self.rows[r] > 0:
                self.frame.index_copy_(0, self.index[r], part)
        return self.frame
