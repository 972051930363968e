"""Starts one worker process per rank on this machine, the way `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N` does: each worker is a fresh interpreter that finds RANK, LOCAL_RANK,
WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR and MASTER_PORT in its environment.

The parent never touches a GPU and never replaces itself with another program: it only starts the
children, relays rank 0's standard output and reports the worst exit status."""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(argv, n_ranks, env_extra=None, timeout=None):
    """Runs `sys.executable argv...` once per rank.  Returns (exit status, rank 0's stdout as text).
    The other ranks' stdout goes to this process's stderr.  When a rank fails or the time runs out the
    remaining ranks are stopped (each by its own process id)."""
    port = free_port()
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ)
        env.update({
            "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(n_ranks),
            "LOCAL_WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
        })
        # this image's host driver only supports dmabuf IPC handles, which RCCL needs between the ranks
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr,
                                      text=True if rank == 0 else None))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # rank 0's pipe is drained while the ranks run, whatever it prints
    deadline = time.time() + timeout if timeout else None
    status = 0
    pending = set(range(n_ranks))
    try:
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and status == 0:
                    status = rc
            if status != 0:
                break
            if deadline and time.time() > deadline:
                status = 124
                break
            time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()  # only reached when a rank failed or the time ran out
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        reader.join(timeout=30)
    return status, "".join(c for c in chunks if c)
