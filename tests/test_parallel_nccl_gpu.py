"""RCCL path of the population sharding on real hardware.  Needs >= 2 MI355X in one node
(`gpurun` boxes have one: the test is skipped there and waits for the day a node exists).
Runs bench.py exactly as the driver does -- torch.distributed.run, one rank per GPU,
backend nccl (= RCCL over xGMI) -- and checks the one JSON line of rank 0."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs in one node')
@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29611', os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '2', '--warmup', '1', '--backend', 'nccl']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['rccl_ranks'] == 2
    assert rec['config']['population'] == 18 and len(rec['config']['last_losses']) == 18
    assert rec['value'] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs in one node')
@pytest.mark.timeout(900)
def test_sharded_step_matches_single_gpu_losses():
    """2 ranks over RCCL vs 1 rank, same seeds: per-candidate losses after the same steps
    agree to the gradient-noise level and the candidates rank identically up to near-ties"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = {}
    for n in (1, 2):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
               '--master-addr', '127.0.0.1', '--master-port', str(29620 + n),
               os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1',
               '--no-cpu-baseline', '--no-fp32-leg', '--no-extra']
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
        assert r.returncode == 0, r.stderr[-4000:]
        out[n] = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    a, b = out[1]['config']['last_losses'], out[2]['config']['last_losses']
    assert max(abs(x - y) for x, y in zip(a, b)) < 2e-3
