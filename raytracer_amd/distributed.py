"""Multi-GPU assembly of a frame rendered in cyclic row tiles (one process per GPU).

Tile t (8 rows) belongs to rank t % world_size, which mirrors the reference's cyclic pixel
partition (src/lib.rs:94-111); rendering needs no communication.  The only exchange step is the one
the reference does through its channel (src/lib.rs:102-104, 121-126): every rank sends its compact
tile buffer to rank 0, which places the rows in frame order.  With the nccl backend that is a grouped
send/recv over RCCL; each peer uses its own xGMI link to rank 0.  The same code runs on CPU tensors
with gloo, which is how it is tested without GPUs.

The buffers hold either the f64 colours (what `render` returns, 24 bytes per pixel) or the clamped
bytes the PPM prints (3 bytes per pixel); rank 0's xGMI ingest is the shared resource, so the byte
form moves an eighth of the data.  OverlappedGather runs the exchange of frame k on a side stream
while frame k + 1 renders.
"""
import numpy as np
import torch
import torch.distributed as dist


class FrameGather:
    def __init__(self, host, width, height, rank, world_size, device, dst=0, dtype=torch.float64, mode="p2p",
                 frames=2):
        """mode "p2p": grouped send/recv, each rank sends exactly its rows.  mode "collective": one
        dist.gather of buffers padded to the largest rank's row count (a plain collective, for
        stacks where grouped point-to-point is not available).  frames: how many assembled frames
        rank dst keeps; call k writes frame k % frames, so a consumer may still read frame k while
        frame k + 1 is being assembled."""
        self.rank, self.world_size, self.dst = rank, world_size, dst
        self.dtype = dtype
        self.mode = mode
        self.padded = None
        self.width, self.height = width, height
        self.rows = [host.local_rows(height, r, world_size) for r in range(world_size)]
        self.frame = None   # the frame the most recent call assembled
        self.frames = None
        self.calls = 0
        self.parts = None
        self.index = None
        self.recv_ops = None
        if rank == dst:
            self.frames = [torch.empty((height, width, 3), dtype=dtype, device=device) for _ in range(max(1, frames))]
            self.frame = self.frames[0]
            # the peers' rows arrive in slices of one staging buffer, so that a single index_copy_
            # puts them all in frame order (one launch per frame instead of one per peer)
            others = [r for r in range(world_size) if r != dst]
            self.staging = torch.empty((sum(self.rows[r] for r in others), width, 3), dtype=dtype, device=device)
            self.parts = [None] * world_size
            at = 0
            for r in others:
                self.parts[r] = self.staging[at: at + self.rows[r]]
                at += self.rows[r]
            self.index = [torch.from_numpy(np.asarray(host.global_rows(height, r, world_size), dtype=np.int64)).to(device)
                          for r in range(world_size)]
            self.index_others = torch.cat([self.index[r] for r in others]) if others else None

    def __call__(self, local):
        """local: this rank's (rows, W, 3) tensor.  Returns the full frame on rank dst."""
        assert tuple(local.shape) == (self.rows[self.rank], self.width, 3) and local.dtype == self.dtype
        if self.frames is not None:
            self.frame = self.frames[self.calls % len(self.frames)]
        self.calls += 1
        if self.world_size > 1 and local.is_cuda and dist.get_backend() == "gloo":
            return self._staged(local)  # rehearsal only: gloo moves host memory
        if self.world_size > 1 and self.mode == "collective":
            return self._collective(local)
        if self.world_size > 1:
            if self.rank == self.dst:
                if self.recv_ops is None:  # the same slices of the staging buffer every frame
                    self.recv_ops = [dist.P2POp(dist.irecv, self.parts[r], r) for r in range(self.world_size)
                                     if r != self.dst and self.rows[r] > 0]
                ops = self.recv_ops
            else:
                ops = [dist.P2POp(dist.isend, local, self.dst)] if self.rows[self.rank] > 0 else []
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        return self._place(local)

    def _wide(self, t):
        """Rows of clamped bytes seen as rows of 8-byte words, so that index_copy_ moves whole words:
        placing the 3,780 peer rows of a 7680x4320 byte frame takes 0.03 ms that way and 0.135 ms byte
        by byte (MI355X, tools/ubench/place_rows.py), on the GPU that is rendering the next frame."""
        if t.dtype == torch.uint8 and (self.width * 3) % 8 == 0 and t.is_contiguous() and t.storage_offset() % 8 == 0:
            return t.view(t.shape[0], self.width * 3).view(torch.int64)
        return t

    def _collective(self, local):
        most = max(self.rows)
        if self.padded is None:
            self.padded = torch.zeros((most, self.width, 3), dtype=self.dtype, device=local.device)
            self.slots = ([torch.empty_like(self.padded) for _ in range(self.world_size)]
                          if self.rank == self.dst else None)
        self.padded[: self.rows[self.rank]].copy_(local)
        dist.gather(self.padded, self.slots, dst=self.dst)
        if self.rank != self.dst:
            return None
        for r in range(self.world_size):
            if self.rows[r] > 0:
                src, dst = self._wide(self.slots[r][: self.rows[r]]), self._wide(self.frame)
                if src.dtype != dst.dtype:
                    src, dst = self.slots[r][: self.rows[r]], self.frame
                dst.index_copy_(0, self.index[r], src)
        return self.frame

    def _staged(self, local):
        if self.rank == self.dst:
            for r in range(self.world_size):
                if r != self.dst and self.rows[r] > 0:
                    tmp = torch.empty(self.parts[r].shape, dtype=self.dtype)
                    dist.recv(tmp, r)
                    self.parts[r].copy_(tmp)
        elif self.rows[self.rank] > 0:
            dist.send(local.cpu(), self.dst)
        return self._place(local)

    def _place(self, local):
        if self.rank != self.dst:
            return None
        frame, own, others = self._wide(self.frame), self._wide(local), self._wide(self.staging)
        if not (frame.dtype == own.dtype == others.dtype):
            frame, own, others = self.frame, local, self.staging
        if self.rows[self.dst] > 0:
            frame.index_copy_(0, self.index[self.dst], own)
        if self.index_others is not None and self.index_others.numel() > 0:
            frame.index_copy_(0, self.index_others, others)
        return self.frame


class OverlappedGather:
    """Double-buffered frames: the exchange of frame k runs on a side stream while frame k + 1 renders
    into the other buffer.  With `render_streams` (one per buffer) consecutive frames also render on
    different streams, so the wavefronts of frame k + 1 fill the GPU while the last long wavefronts
    of frame k finish; each buffer then needs its own scene handle, because launches on one scene
    must be stream-ordered.  On CPU tensors (tests) it degrades to render then gather.  step()
    returns the assembled frame (rank dst) and, on CUDA, sets `self.frame_done` to the event after
    which it is complete; the gather alternates between its frame buffers, so the next step() does
    not overwrite the frame just returned.  `wait()` waits for everything."""

    def __init__(self, gather, buffers, render_streams=None):
        assert len(buffers) >= 2
        self.gather, self.buffers, self.k = gather, buffers, 0
        self.cuda = buffers[0].is_cuda
        self.render_streams = render_streams
        if self.cuda:
            assert render_streams is None or len(render_streams) == len(buffers)
            self.side = torch.cuda.Stream(device=buffers[0].device)
            self.done = [None] * len(buffers)
        self.frame_done = None

    def step(self, render):
        """render(buffer, slot) enqueues the render of the next frame into `buffer` on the current
        stream; `slot` is the buffer's index."""
        b = self.k % len(self.buffers)
        self.k += 1
        buf = self.buffers[b]
        if not self.cuda:
            render(buf, b)
            return self.gather(buf) if self.gather is not None else buf
        stream = self.render_streams[b] if self.render_streams else torch.cuda.current_stream(buf.device)
        with torch.cuda.stream(stream):
            if self.done[b] is not None:
                stream.wait_event(self.done[b])  # the exchange that last read this buffer
            render(buf, b)
            ready = torch.cuda.Event()
            ready.record(stream)
        if self.gather is None:
            self.done[b] = ready
            return buf
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            frame = self.gather(buf)
            self.done[b] = torch.cuda.Event()
            self.done[b].record(self.side)
            self.frame_done = self.done[b]
        return frame

    def wait(self):
        if self.cuda:
            torch.cuda.synchronize(self.buffers[0].device)
