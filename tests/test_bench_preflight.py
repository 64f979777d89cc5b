"""bench.py's communication preflight: the first multi-rank RCCL run happens on the
driver's node, so a failing or hanging set-up must leave a JSON record (`comm.error`) instead of a silent time-out.  Run here
on CPU with gloo: a healthy world of 2, and a world of 2 in which one rank never shows up."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import json, os, sys, types
sys.path.insert(0, os.environ['TE_ROOT'])
import torch
import bench
args = types.SimpleNamespace(steps=4, warmup=1)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
info, dog = bench.comm_preflight(args, 'gloo', world, rank, rank, torch.device('cpu'))
dog.disarm()
import torch.distributed as dist
if rank == 0:
    print(json.dumps(info))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(rank, world, port, **kw):
    return dict(os.environ, TE_ROOT=ROOT, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                MASTER_PORT=str(port), **kw)


def test_preflight_world2_reports_ranks_and_library():
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, '-c', _SCRIPT], env=_env(r, 2, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    info = json.loads(outs[0][0].strip().splitlines()[-1])
    assert info['world_size'] == 2 and [r['rank'] for r in info['ranks']] == [0, 1]
    assert 'rccl_version' in info and info['preflight_allreduce_s'] >= 0 and info['backend'] == 'gloo'


def test_preflight_hang_leaves_a_json_record_and_exit_code_3():
    port = _free_port()
    # rank 1 never starts: rank 0 waits in the rendezvous; the watchdog (3 x TE_BENCH_COMM_TIMEOUT) must end it with a record
    p = subprocess.run([sys.executable, '-c', _SCRIPT], env=_env(0, 2, port, TE_BENCH_COMM_TIMEOUT='2'), capture_output=True, text=True,
                       timeout=240)
    assert p.returncode == 3, (p.returncode, p.stderr[-1500:])
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec['value'] is None and rec['n_gpus'] == 2 and 'did not complete within' in rec['comm']['error']
    assert 'watchdog' in p.stderr
