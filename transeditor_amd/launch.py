"""One-process-per-GPU launcher for the reference's UNCHANGED training script.

    python -m transeditor_amd.launch --nproc 8 train_spatial_query.py --batch 16 --size 256 ...

Why it exists: `train_spatial_query.py:399` only accepts `--local_rank` and pins the device with it
(`:426`), but torch >= 2.0's `torch.distributed.run` passes `--local-rank` (dash) or just the LOCAL_RANK variable, so the
unchanged script either fails to parse or puts every rank on GPU 0.  This launcher starts `nproc` copies with
`--local_rank=<i>` appended and the `env://` rendezvous variables the script's `init_process_group(backend='nccl',
init_method='env://')` (`:427`) reads: MASTER_ADDR (127.0.0.1: the container hostname may not resolve), MASTER_PORT, RANK,
WORLD_SIZE, LOCAL_RANK; HSA_ENABLE_IPC_MODE_LEGACY=0 for RCCL over dmabuf IPC.  If a rank dies the others are terminated.
"""
import argparse
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--nproc', type=int, required=True, help='processes = GPUs on this node')
    ap.add_argument('--master-port', type=int, default=0)
    ap.add_argument('--no-local-rank-arg', action='store_true', help='do not append --local_rank=<i> (scripts that read LOCAL_RANK)')
    ap.add_argument('script')
    ap.add_argument('script_args', nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    port = a.master_port or free_port()
    procs = []
    for r in range(a.nproc):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(a.nproc),
                   LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-u', a.script, *a.script_args]
        if not a.no_local_rank_arg:
            cmd.append(f'--local_rank={r}')
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in alive:                   # a rank failed: stop exactly the processes started here
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.1)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        rc = 130
    return rc


if __name__ == '__main__':
    sys.exit(main())
