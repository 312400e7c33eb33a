"""A candidate's bits must not depend on who shares its launch -- include/p2l.h promises it, CMA-ES ranks on
these numbers (/root/reference pix2latent/optimizer/base_cma_optimizer.py:117-140) and the gradients are
piecewise, so a one-ulp difference can part two trajectories after a few steps
(/root/reference pix2latent/optimizer/closure.py:27,58 is the semantics at stake: the reference's result does
not depend on how the population is cut into chunks beyond the 1/b_chunk gradient factor).

Real BigGAN-deep-256 shapes, the bench problem: 3 Adam steps + the forward-only re-score, run as
  * ONE process, one device pass of 18 candidates,
  * ONE process, the reference's chunks 9 + 9,
  * 2 and 4 ranks sharing this GPU over gloo (9+9 and 5+5+4+4 local candidates; HIP-graph replay for <= 6),
and the per-candidate losses of EVERY step, the re-scored losses and the final latents compared bit for bit.
Until round 5 the split-K factor of the 4^2 ... 16^2 layers followed the grid size (hence the batch) and
this was `< 1e-5, same order`."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, 'tools', 'shard_bits.py')


def _run(world, extra=(), **more_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **more_env)
    if world == 1:
        cmd = [sys.executable, TOOL] + list(extra)
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
               '--master-addr', '127.0.0.1', '--master-port', str(29710 + world), TOOL, '--backend', 'gloo'] + list(extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"world"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(3000)
def test_bits_do_not_depend_on_the_number_of_ranks():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    one = _run(1)
    assert len(one['rescore']) == 18 and len(one['steps']) == 3
    # (chunks 9+9: on two streams, one lane of workspaces each -- pix2latent_amd/lanes.py -- and on one stream)
    runs = {'chunks 9+9': _run(1, ['--chunks']), 'chunks 9+9, one stream': _run(1, ['--chunks'], P2L_STREAMS='1'),
            '2 ranks': _run(2), '4 ranks': _run(4),
            '4 ranks, eager': _run(4, ['--graph', '0'])}
    assert runs['4 ranks']['local_candidates'] == [5, 5, 4, 4]
    for name, rec in runs.items():
        for i, (a, b) in enumerate(zip(one['steps'], rec['steps'])):
            assert a == b, '%s: losses of step %d differ from the single pass of 18 in candidates %s' % (
                name, i, [j for j in range(18) if a[j] != b[j]])
        assert rec['rescore'] == one['rescore'], '%s: re-scored losses differ in candidates %s' % (
            name, [j for j in range(18) if rec['rescore'][j] != one['rescore'][j]])
        assert rec['argsort'] == one['argsort']
        assert rec['z_bits_sum'] == one['z_bits_sum'], '%s: final latents differ' % name
