"""self-attention of the BigGAN-256 generator, 18 candidates: fused kernels vs the
materialised GEMM + softmax sequence"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N
dev = 'cuda'; B = int(os.environ.get('P2L_POP', 18)); Nq, Nk = 4096, 1024
g = torch.Generator().manual_seed(0)
q = (0.3 * torch.randn(B, Nq, 64, generator=g)).to(dev); k = (0.3 * torch.randn(B, Nk, 64, generator=g)).to(dev)
v = torch.randn(B, Nk, 256, generator=g).to(dev); do = torch.randn(B, Nq, 256, generator=g).to(dev)

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def old_fwd():
    S = O.gemm(q, k, B, Nq, Nk, 64)
    P = O.softmax_fwd(S.view(B * Nq, Nk))
    return O.gemm(P.view(B, Nq, Nk), v, B, Nq, 256, Nk, b_kmajor=True)
out, lse = O.attn_fwd(q, k, v)
print('forward : materialised %.3f ms   fused %.3f ms' % (timed(old_fwd), timed(lambda: O.attn_fwd(q, k, v))))
P = O.softmax_fwd(O.gemm(q, k, B, Nq, Nk, 64).view(B * Nq, Nk)).view(B, Nq, Nk)
print('d values: materialised %.3f ms   fused %.3f ms' % (
    timed(lambda: O.gemm(P, do, B, Nk, 256, Nq, a_kmajor=True, b_kmajor=True)), timed(lambda: O.attn_bwd_dv(q, k, do, lse))))
fl = 2.0 * B * Nq * Nk * (64 + 256)
t = timed(lambda: O.attn_fwd(q, k, v), 20)
print('fused forward: %.1f TFLOP/s algorithmic' % (fl / t / 1e9))
dP = O.gemm(do, v, B, Nq, Nk, 256)
def old_qk():
    dPm = O.gemm(do, v, B, Nq, Nk, 256)
    dS = O.softmax_bwd(P.view(B * Nq, Nk), dPm.view(B * Nq, Nk)).view(B, Nq, Nk)
    dq = O.gemm(dS, k, B, Nq, 64, Nk, b_kmajor=True)
    dk = O.gemm(dS, q, B, Nk, 64, Nq, a_kmajor=True, b_kmajor=True)
    return dq, dk
print('d q, d k: materialised %.3f ms (+ S, softmax recompute)   fused %.3f ms' % (
    timed(old_qk), timed(lambda: O.attn_bwd_qk(q, k, v, out, do, lse))))
