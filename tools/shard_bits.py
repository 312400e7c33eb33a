"""Do a candidate's bits depend on how many ranks share the population?  (VERDICT r4 weak #2.)

The bench problem (real BigGAN-deep-256 shapes, pop 18, weighted L1 + 10 LPIPS-VGG16) runs
`--steps` Adam steps and one forward-only re-score, either in one process (world 1: one device pass
of 18, or the reference's chunks 9 + 9 with --chunks) or block-sharded over WORLD_SIZE ranks
(torch.distributed.run; --backend gloo lets the ranks share one GPU).  Rank 0 prints ONE JSON line
with the losses of every step and of the re-score as float32 bit patterns: tests/test_shard_bits_gpu.py
compares the lines of 1, 2 and 4 ranks for EQUALITY.

    python tools/shard_bits.py
    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/shard_bits.py --backend gloo
"""
import argparse, contextlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--backend', default='nccl')
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--chunks', action='store_true', help='world 1: execute in the reference chunks of 9')
ap.add_argument('--graph', default=None, help="'0' / '1': force eager / HIP-graph execution of the steps")
args = ap.parse_args()
if args.graph is not None:
    os.environ['P2L_GRAPH'] = args.graph
import torch.distributed as dist
world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
local_rank = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
torch.cuda.set_device(local_rank)
dev = torch.device('cuda', local_rank)
if world > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)
    else:
        dist.init_process_group(args.backend)
import bench
torch.manual_seed(0)
opt, vm, _ = bench.build_problem(dev, exec_batch_size=None if args.chunks else 'all')
with contextlib.redirect_stdout(sys.stderr):
    opt.setup_cma(vm)
    variables = opt.cma_init(vm)


def bits(losses):
    return [int(v) for v in np.asarray([float(x) for x in losses], dtype=np.float32).view(np.uint32)]


steps = []
for i in range(args.steps):
    opt.step(variables, optimize=True, transform=(i == 0))
    steps.append(bits(opt.loss))                 # (sharded: the all-gather, on every rank)
_, losses, _ = opt.step(variables, optimize=False)
rescore = bits(losses)
opt.gather_population(variables)
z = torch.stack([t.detach().float().cpu() for t in variables.input.z.data])
if rank == 0:
    print(json.dumps({'world': world, 'backend': dist.get_backend() if world > 1 else None,
                      'local_candidates': [hi - lo for lo, hi in __import__('pix2latent_amd.parallel', fromlist=['partition']).partition(18, world)],
                      'chunks': bool(args.chunks), 'steps': steps, 'rescore': rescore,
                      'rescore_values': [float(x) for x in np.asarray(rescore, dtype=np.uint32).view(np.float32)],
                      'argsort': [int(i) for i in np.argsort(np.asarray(rescore, dtype=np.uint32).view(np.float32))],
                      'z_bits_sum': int(z.numpy().view(np.uint32).astype(np.uint64).sum())}))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
