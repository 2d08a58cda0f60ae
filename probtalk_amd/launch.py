"""One process per GPU without torchrun: start N ranks of a command, watch them, never leave one behind.

The reference's multi-device site is single-process `nn.DataParallel` (nets/inpainting/predictor.py:100-105); its
replacement here is one process per GPU over RCCL.  `bench.py --gpus N` (no WORLD_SIZE in the environment) uses
`spawn_ranks` to become the parent of N fresh children; the parent never touches the GPU (a process that has initialised
HIP must not be replaced or forked), the children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _kill(procs, grace=5.0):
    """End exactly the children we started (by pid): SIGTERM, then SIGKILL after `grace` seconds."""
    for p in procs:
        if p.poll() is None:
            try:
                p.send_signal(signal.SIGTERM)
            except OSError:
                pass
    t_end = time.monotonic() + grace
    for p in procs:
        while p.poll() is None and time.monotonic() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def spawn_ranks(cmd, nproc, limit_s, extra_env=None, poll_s=0.1, log=sys.stderr):
    """Run `cmd` (argv list) as `nproc` ranks on 127.0.0.1 and wait for them.

    Returns 0 when every rank exited 0.  The first rank that exits non-zero ends the job: the remaining ranks (which
    would otherwise sit in a collective waiting for it) are killed and its code is returned.  After `limit_s` seconds of
    wall clock every rank is killed and 124 is returned (the code `timeout` uses)."""
    port = free_port()
    procs = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(cmd, env=env))
    t_end = time.monotonic() + limit_s
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"[launch] rank {r} exited with code {c}: stopping the other ranks", file=log, flush=True)
                _kill(procs)
                return c if c > 0 else 128 - c
            if all(c == 0 for c in codes):
                return 0
            if time.monotonic() > t_end:
                print(f"[launch] {nproc} ranks still running after {limit_s:.0f} s: killing them", file=log, flush=True)
                _kill(procs)
                return 124
            time.sleep(poll_s)
    except BaseException:
        _kill(procs)
        raise
