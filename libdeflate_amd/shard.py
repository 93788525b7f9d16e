"""One process per GPU, no collective on the data path.

Streams are independent, so a job over N GPUs is N shards of streams; the only communication is
the rendezvous around the timed region (barrier, MAX of the elapsed time, agreement on the
verification flag).  bench.py uses this with the "nccl" (= RCCL) backend and CUDA tensors; the CPU
tests run the same code with "gloo" and world_size 2.
"""
import os
import socket
import subprocess
import sys
import time


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend, device=None):
    """Join the job.  Returns the torch.distributed module, or None for a single process."""
    rank, _, world = env_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    import datetime
    # (a rank that never arrives must not hold the others for the default half hour: launch_ranks stops its siblings
    # at once, this bounds the same case under torch.distributed.run)
    kw = {"timeout": datetime.timedelta(seconds=float(os.environ.get("LDH_DIST_TIMEOUT", "600")))}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_of(total_streams, rank, world):
    """Streams [lo, hi) of a strong-scaling job handled by `rank` (contiguous, sizes differ by <= 1)."""
    q, r = divmod(total_streams, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _reduce(dist, torch, value, dtype, op, device):
    t = torch.tensor([value], dtype=dtype, device=device)
    dist.all_reduce(t, op=op)
    return t.item()


def timed_steps(step, steps, warmup, sync, dist=None, torch=None, device="cpu"):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + device sync on both sides.
    Returns the elapsed seconds, MAX over ranks."""
    def fence():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt_local = time.perf_counter() - t0
    fence()
    if dist is None:
        return dt_local
    return float(_reduce(dist, torch, dt_local, torch.float64, dist.ReduceOp.MAX, device))


def all_true(ok, dist=None, torch=None, device="cpu"):
    if dist is None:
        return bool(ok)
    return bool(_reduce(dist, torch, 1 if ok else 0, torch.int32, dist.ReduceOp.MIN, device))


def total(value, dist=None, torch=None, device="cpu"):
    if dist is None:
        return value
    return _reduce(dist, torch, value, torch.float64, dist.ReduceOp.SUM, device)


def _free_port():
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(argv, world, env=None, timeout=None, grace=5.0):
    """Start `world` fresh processes of `argv` (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would) and wait for them.  Returns the first non-zero exit code, else 0.
    The children are polled together: the moment one of them exits non-zero (no device, out of memory, a crash) the
    others -- which would sit in init_process_group or a barrier until the distributed timeout -- are terminated
    (SIGTERM, SIGKILL after `grace` seconds) and that rank's code is returned; `timeout` seconds (None: no limit) for
    the whole job end it the same way with code 124.  The caller must not have touched the GPU: the children are new
    processes (never a re-exec of this one), and a parent that holds no GPU state just waits."""
    port = _free_port()
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                  "MASTER_PORT": str(port)})
        procs.append(subprocess.Popen(list(argv), env=e))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + grace
        for p in procs:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t0 = time.monotonic()
    rc = 0
    try:
        while True:
            running = 0
            for r, p in enumerate(procs):
                code = p.poll()
                if code is None:
                    running += 1
                elif code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write("rank process %d exited with %d: stopping the other ranks\n" % (r, code))
            if rc or not running:
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc = 124
                sys.stderr.write("ranks still running after %.0f s: stopping them\n" % timeout)
                break
            time.sleep(0.05)
    finally:
        stop_all()
    return rc
