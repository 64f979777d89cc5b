"""The --local_rank launcher (train_spatial_query.py:399,426 accept only --local_rank, torch >= 2.0 passes --local-rank)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = '''
import argparse, os, torch, torch.distributed as dist
p = argparse.ArgumentParser()
p.add_argument('--local_rank', type=int, default=0)      # exactly the reference's declaration (train_spatial_query.py:399)
p.add_argument('--tag', type=str)
a = p.parse_args()
assert int(os.environ['LOCAL_RANK']) == a.local_rank == int(os.environ['RANK'])
dist.init_process_group(backend='gloo', init_method='env://')      # the reference uses nccl + env:// (:427)
t = torch.tensor([float(a.local_rank + 1)])
dist.all_reduce(t)
assert t.item() == 3.0 and dist.get_world_size() == 2 and os.environ['MASTER_ADDR'] == '127.0.0.1'
open(os.path.join(os.environ['OUT_DIR'], f'rank{a.local_rank}.{a.tag}'), 'w').write('ok')
dist.destroy_process_group()
'''


def test_launcher_passes_local_rank_and_rendezvous(tmp_path):
    script = tmp_path / 'fake_train.py'
    script.write_text(SCRIPT)
    env = dict(os.environ, OUT_DIR=str(tmp_path), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'transeditor_amd.launch', '--nproc', '2', str(script), '--tag', 'x'], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / 'rank0.x').exists() and (tmp_path / 'rank1.x').exists()


def test_launcher_propagates_failure(tmp_path):
    script = tmp_path / 'bad.py'
    script.write_text('import sys, time\nif "--local_rank=1" in sys.argv: sys.exit(7)\ntime.sleep(60)\n')
    r = subprocess.run([sys.executable, '-m', 'transeditor_amd.launch', '--nproc', '2', str(script)], cwd=ROOT,
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode == 7
