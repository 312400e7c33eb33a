"""bench.py as the driver runs it: the single-GPU line, and the sharded launch with several
ranks SHARING this box's one GPU over gloo (the RCCL launch needs one GPU per rank:
tests/test_parallel_nccl_gpu.py).  Ranks of a 4- or 8-GPU job hold <= 6 candidates, which
switches the optimizers to HIP-graph replay by default -- the regime in which the per-launch
profiler of the bench once failed for every rank."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-fp32-leg', '--no-extra']


def _bench_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_single_gpu_line():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + QUICK, cwd=ROOT,
                       capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, r.stderr[-4000:]
    rec = _bench_line(r.stdout)
    assert rec['n_gpus'] == 1 and rec['steps'] == 2 and rec['value'] > 0
    assert rec['unit'] == 'evals/s' and rec['higher_is_better'] is True
    roof = rec['roofline']
    assert roof['bound'] == 'mfma' and 0 < roof['frac'] < 1 and roof['sampled_launches'] > 0
    assert roof['launches_per_step'] > 0 and roof['algorithmic']['tflops'] > roof['algorithmic']['executed_fp32_equiv_tflops'] > 0
    assert 'executed' in roof['frac_is']          # frac = what the matrix pipe executes / its dense peak
    assert roof['conv1x1']['bound'] == 'hbm' and 'telemetry' in rec
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    assert len(rec['config']['last_losses']) == 18
    # the top-level figures are the kernels' own (one pass of 18, nothing beside the launches):
    # they fit inside their step; what the launches get while the two lanes overlap sits under `concurrent`
    assert rec['config']['lanes'] == 2 and roof['exec_batch_size'] == 18 and roof['lanes'] == 1
    # the timed steps run as optimize() runs them: one HIP graph with two branches, replayed
    assert rec['config']['hip_graph_replay'] is True
    assert 0 < roof['time_share_of_step'] <= 1.0
    dom = roof['dominant_kernel']
    assert dom['name'].startswith('wino16s_conv_kernel') and 0 < dom['frac'] < 1 and dom['avg_launch_ms'] > 0
    assert dom['time_share_of_step'] <= roof['time_share_of_step']
    conc = roof['concurrent']
    assert conc['exec_batch_size'] == 9 and conc['lanes'] == 2 and conc['frac'] < roof['frac']
    # a traffic record belongs to the execution batch of its leg, or is absent with the reason
    for leg in (roof, conc):
        src = leg['traffic_source']
        assert leg['traffic'] is None or src['exec_batch_size'] == leg['exec_batch_size']


@pytest.mark.timeout(600)
def test_bench_eager_form_for_profilers():
    """`--pmc-run` (implies --eager: events inside the timed steps, no legs behind them) is what tools/gpu_round6.sh
    wraps in rocprofv3: warm-up + timed steps only, every dispatch at the named execution batch; the record is then
    built from the timed steps themselves"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    for eb, lanes in ((9, 2), (18, 1)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--pmc-run', '--steps', '2', '--warmup', '1',
                            '--exec-batch', str(eb)], cwd=ROOT, capture_output=True, text=True, timeout=550)
        assert r.returncode == 0, r.stderr[-4000:]
        rec = _bench_line(r.stdout)
        assert rec['config']['hip_graph_replay'] is False and rec['config']['exec_batch_size'] == eb
        assert rec['config']['lanes'] == lanes and 'cpu_baseline' not in rec and 'extra' not in rec['config']
        roof = rec['roofline']
        assert roof['exec_batch_size'] == eb and roof['sampled_launches'] > 0 and 'concurrent' not in roof
        src = roof['traffic_source']
        assert roof['traffic'] is None or src['exec_batch_size'] == eb


@pytest.mark.parametrize('world', [2, 4])
@pytest.mark.timeout(900)
def test_bench_sharded_ranks_share_one_gpu(world):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(29630 + world),
           os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--backend', 'gloo'] + QUICK
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-4000:]
    rec = _bench_line(r.stdout)
    assert rec['n_gpus'] == world and rec['value'] > 0 and rec['scaling'] == 'strong'
    assert rec['config']['population'] == 18 and len(rec['config']['last_losses']) == 18
    assert rec['roofline']['sampled_launches'] > 0          # the eager timed loop was profiled
