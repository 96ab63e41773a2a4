"""First execution of the N > 1 path on a GPU: RCCL process-group initialisation and the flat gradient all-reduce
(cliora_amd/parallel.py; the reference's DDP over NCCL, cliora/net/trainer.py:528-532, 572-574), at world size 1.

`bench.py --force-dist` initialises the process group with backend nccl (= RCCL on ROCm) and runs FlatGradAllReduce on the GPU
gradients every step even with one rank.  It runs as a CHILD process: the group's state stays out of the test session."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_force_dist_runs_rccl_at_world_size_one():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '3', '--warmup', '1', '--no-extras',
           '--no-cpu-baseline', '--no-kernel-events']
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert lines, r.stdout[-2000:]
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0
    ge = d['config'].get('gradient_exchange', '')
    assert ge.startswith('RCCL') and '; 0 of ' in ge, d['config']        # every gradient written in place by the chart backward


def test_bench_c3_force_dist_runs_the_cliora_step_through_rccl():
    """BASELINE configs[3] is `bench.py --gpus 8 --workload c3`; its per-rank step (CLIORA training step: Embed, ImageEncoder, chart with
    regions, three losses, backward, ONE flat all-reduce of chart + head + ImageEncoder gradients, clip + Adam; trainer.py:437-501, 572-574)
    runs here at world size 1 through RCCL, as a child process: the CLIORA parameter set goes through the flat buffer on the GPU."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29534', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--workload', 'c3', '--force-dist', '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert lines, r.stdout[-2000:]
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0
    assert 'configs[2]' in d['config']['workload'] and d['config']['global_batch'] == 64
    ge = d['config'].get('gradient_exchange', '')
    assert ge.startswith('RCCL') and '; 0 of ' in ge, d['config']        # chart, head and ImageEncoder gradients all written in place


def test_bench_via_launcher_takes_the_n_gpu_route_on_one_gpu():
    """`bench.py --gpus N` (N > 1) starts torch.distributed.run as a child process before anything touches the GPU (bench.py: launch_ranks);
    no 8-GPU node runs in a round, so the SAME route is taken here with one rank: --via-launcher sends --gpus 1 through launch_ranks(), the
    rank initialises RCCL from the launcher's environment, and the line must show that the process group saw WORLD_SIZE ranks
    (config.ranks_seen = an all-reduce of ones) and every rank's own ms per step (batch_iterator.py:134-136, trainer.py:572-574)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}      # no launcher around this one
    env.update(MASTER_PORT='29537', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--via-launcher', '--steps', '3', '--warmup', '1', '--no-extras',
           '--no-cpu-baseline', '--no-kernel-events']
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert lines, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[-1])
    c = d['config']
    assert d['n_gpus'] == 1 and d['value'] > 0
    assert c['via_launcher'] is True and c['ranks_seen'] == c['world_size'] == 1, c
    assert c['per_rank_ms_per_step']['min'] > 0 and c['per_rank_ms_per_step']['max'] >= c['per_rank_ms_per_step']['min']
    assert c.get('gradient_exchange', '').startswith('RCCL'), c


@pytest.mark.parametrize('workload', ['c2', 'c3'])
def test_bench_two_ranks_on_one_gpu_through_the_launcher(workload):
    """World size 2 on real kernels without a second GPU: `bench.py --gpus 2 --share-device --backend gloo` -- launch_ranks() starts two ranks
    (torch.distributed.run), both on cuda:0, each with its own 64 sentences, the flat gradient buffer all-reduced through gloo (RCCL refuses
    two ranks on one device).  Not a scaling figure: the point is that the N > 1 route -- rank-seeded batches, FlatGradAllReduce on gradients
    the chart backward wrote in place, barrier + max-over-ranks timing, one JSON line from rank 0 -- runs on the GPU with N = 2
    (batch_iterator.py:134-136, trainer.py:572-574)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(MASTER_PORT='29539', HSA_ENABLE_IPC_MODE_LEGACY='0')
    # c3: the CLIORA training step of configs[3] (word branch on the caller lane, deferred table gradient, flat buffer over chart + heads)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--share-device', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
           '--workload', workload, '--no-extras', '--no-cpu-baseline', '--no-kernel-events']
    env['MASTER_PORT'] = '29539' if workload == 'c2' else '29540'
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])          # rank 0 only
    d = json.loads(lines[-1])
    c = d['config']
    assert d['n_gpus'] == 2 and c['global_batch'] == 128 and d['value'] > 0
    assert c['ranks_seen'] == c['world_size'] == 2 and c['via_launcher'] is True, c
    assert len(c['per_rank_ms_per_step']['all']) == 2 and 'share_device' in c
    assert c['gradient_exchange'].startswith('gloo') and '; 0 of ' in c['gradient_exchange'], c
