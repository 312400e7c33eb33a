"""BigGAN-deep-256 generator on the native MI355X path.

Host-side mirror of `pix2latent.model.BigGAN` (reference
pix2latent/model/biggan.py:15-58): same constructor argument, same
`get_class_embedding(cls)` and `forward(z=, c=, truncation=1.0)` contract and
assertion messages, but the arithmetic is libp2l_hip (`p2l_biggan_fwd/bwd`)
instead of pytorch_pretrained_biggan + autograd.  Parameters are frozen: only
d/dz and d/dc are computed (the reference also computes weight gradients that
nobody reads).
"""
import ctypes as C
import math
import os
import warnings

import torch
import torch.nn as nn

from .. import _native as N
from .. import lanes
from ..utils import synthetic

BN_EPS = 1e-4
N_STATS = 51


def _bn_row(means, vars_, truncation):
    """statistics row of BigGANBatchNorm for a truncation value (51 rows, step
    0.02; interpolated between rows when truncation is not a multiple)."""
    step = 1.0 / (N_STATS - 1)
    coef, start = math.modf(truncation / step)
    start = int(start)
    if coef != 0.0:
        mean = means[start] * coef + means[start + 1] * (1 - coef)
        var = vars_[start] * coef + vars_[start + 1] * (1 - coef)
    else:
        mean, var = means[start], vars_[start]
    return mean, var


class _GeneratorView(object):
    """tiny stand-in for `biggan.generator` so that code reading
    `model.generator.gen_z` (reference pix2latent/edit/ganspace.py:36) works."""

    def __init__(self, weights):
        self.gen_z = nn.Linear(weights['generator.gen_z.weight'].shape[1],
                               weights['generator.gen_z.weight'].shape[0])
        with torch.no_grad():
            self.gen_z.weight.copy_(weights['generator.gen_z.weight'])
            self.gen_z.bias.copy_(weights['generator.gen_z.bias'])
        self.gen_z.requires_grad_(False)


class _BigGANFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, c, model):
        z = z.contiguous().float()
        c = c.contiguous().float()
        out = model._run_forward(z, c)
        ctx.model = model
        ctx.lane = lanes.current()
        ctx.ticket = model._ticket
        ctx.save_for_backward(z, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        z, c = ctx.saved_tensors
        with lanes.use(ctx.lane):        # (the saved activations live in the forward's lane)
            if model._ticket != ctx.ticket:
                # another forward reused the workspace: rebuild the saved activations
                model._run_forward(z, c)
                ctx.ticket = model._ticket
            dz, dc = model._run_backward(z.shape[0], dout)
        return dz, dc, None


class _Lane(object):
    """the device scratch of one execution lane (lanes.py): arena, image staging, forward ticket"""

    def __init__(self):
        self.ws, self.ws_bytes, self.ws_B = None, 0, -1
        self.img16 = self.dimg16 = None
        self.ticket, self.last_B = 0, 0


def _lane_attr(name):
    return property(lambda self: getattr(self._lane_state(), name),
                    lambda self, v: setattr(self._lane_state(), name, v))


class BigGAN(nn.Module):
    """
    Drop-in for pix2latent.model.BigGAN.

    Args:
        model_version: only 'biggan-deep-256' (reference biggan.py:23).
        weights: dict of CPU tensors keyed like the HF state_dict with spectral
            norm baked out.  None -> $P2L_BIGGAN_WEIGHTS (torch.load) if set,
            else seeded random-init weights of the same architecture (no network
            access exists to fetch the pretrained checkpoint).
        device: where the packed parameters live (a ROCm device).
    """

    def __init__(self, model_version='biggan-deep-256', weights=None, device='cuda',
                 seed=0, wfmt=None):
        super(BigGAN, self).__init__()
        if model_version != 'biggan-deep-256':
            raise ValueError('only biggan-deep-256 is implemented, got %s' % model_version)
        if weights is None:
            path = os.environ.get('P2L_BIGGAN_WEIGHTS')
            if path:
                from ..utils.checkpoint import load_biggan_state_dict
                weights = load_biggan_state_dict(torch.load(path, map_location='cpu'))
            else:
                warnings.warn('BigGAN: no pretrained weights available (no network); '
                              'using seeded random-init biggan-deep-256 weights')
                weights = synthetic.biggan_weights(seed)
        self.ch = synthetic.CH
        self.z_dim = synthetic.Z_DIM
        self.truncation = None
        self._dev = torch.device(device)
        if self._dev.type != 'cuda':
            raise N.NativeError('BigGAN needs a ROCm device: the generator only exists as '
                                'HIP kernels (no CPU fallback)')
        self._lib = N.lib()
        self._w = weights
        self.embeddings = nn.Linear(weights['embeddings.weight'].shape[1], self.z_dim, bias=False)
        with torch.no_grad():
            self.embeddings.weight.copy_(weights['embeddings.weight'])
        self.embeddings.requires_grad_(False)
        self.embeddings.to(self._dev)
        self.generator = _GeneratorView(weights)
        self._keep = []          # device tensors referenced by the C struct
        self._desc = N.P2LBigGAN()
        # arithmetic of the 3x3 convs: bf16x3 split (fp32-equivalent, default) or exact fp32
        self._wfmt = N.default_wfmt() if wfmt is None else wfmt
        # 1x1 convs: same arithmetic on the bf16 pipe (their buffers then carry the pre-split image)
        self._pw = N.default_pw() and self._wfmt != N.WFMT_F32
        # conv_to_rgb and its input gradient: three real channels on one side (csrc/p2l_thin.hip)
        self._thin = N.default_thin() and self._wfmt != N.WFMT_F32
        self._desc.wfmt = (self._wfmt | (N.WFMT_FLAG_PW if self._pw else 0) |
                           (N.WFMT_FLAG_THIN if self._thin else 0) |
                           (N.WFMT_FLAG_ATTN_GEMM if N.default_attn_gemm() else 0) |
                           (N.WFMT_FLAG_NO_AMAX if N.default_no_amax() else 0))
        self._lanes = {}         # lane -> _Lane (lanes.py: one per stream that runs chunks of a step)
        self.ws_generation = 0
        self._pack(weights)
        self._set_truncation(1.0)

    # ------------------------------------------------------------------ setup
    def _dev_t(self, t):
        t = t.detach().to(self._dev, torch.float32).contiguous()
        self._keep.append(t)
        return t

    def _pack_conv(self, w, taps, n_pad, k_pad, flip, thin=False):
        fmt = self._wfmt if taps == 9 else (N.WFMT_PW if self._pw else N.WFMT_F32)
        if thin and self._thin:
            fmt = N.WFMT_BF16X3T
        dst = N.pack_conv_weight(w.detach().to(self._dev, torch.float32), taps, n_pad, k_pad, flip, fmt)
        torch.cuda.current_stream().synchronize()
        self._keep.append(dst)
        return dst

    def _pack_subpix(self, w, n_pad, k_pad, flip):
        dst = N.pack_conv_weight(w.detach().to(self._dev, torch.float32), 9, n_pad, k_pad, flip,
                                 self._wfmt, subpix_mode=0)
        torch.cuda.current_stream().synchronize()
        self._keep.append(dst)
        return dst

    def _pack(self, W):
        d = self._desc
        table = synthetic.layer_table()
        blocks = [(i, s) for i, s in enumerate(table) if s[0] == 'block']
        attn = [(i, s) for i, s in enumerate(table) if s[0] == 'attn']
        d.n_blocks = len(blocks)
        d.attn_before = synthetic.ATTN_POS
        d.ch = self.ch
        d.z_dim = self.z_dim
        d.c_dim = self.z_dim
        d.genz_w = self._dev_t(W['generator.gen_z.weight'].t()).data_ptr()
        d.genz_b = self._dev_t(W['generator.gen_z.bias']).data_ptr()
        scale_cols, offset_cols = [], []
        self._bn_prefixes = []
        off = 0
        for bi, (li, spec) in enumerate(blocks):
            _, up, cin, cout = spec
            mid = cin // 4
            p = 'generator.layers.%d' % li
            g = d.blocks[bi]
            g.cin, g.cout, g.up = cin, cout, int(up)
            chans = [cin, mid, mid, mid]
            for k in range(4):
                g.cbn_off[k] = off
                off += chans[k]
                bp = '%s.bn_%d' % (p, k)
                self._bn_prefixes.append(bp)
                scale_cols.append(W[bp + '.scale.weight'].t())
                offset_cols.append(W[bp + '.offset.weight'].t())
            shapes = [(1, mid, cin), (9, mid, mid), (9, mid, mid), (1, cout, mid)]
            for k, (taps, o, i_) in enumerate(shapes):
                w = W['%s.conv_%d.weight' % (p, k)]
                assert tuple(w.shape[:2]) == (o, i_), (p, k, w.shape)
                g.w[k] = self._pack_conv(w, taps, o, i_, False).data_ptr()
                g.wt[k] = self._pack_conv(w, taps, i_, o, True).data_ptr()
                g.b[k] = self._dev_t(W['%s.conv_%d.bias' % (p, k)]).data_ptr()
            if up:
                # conv_1 reads a nearest-x2 upsampled input: sub-pixel forms (2.25x fewer FLOPs)
                w1 = W['%s.conv_1.weight' % p]
                g.w1_sp = self._pack_subpix(w1, mid, mid, False).data_ptr()
                g.wt1_sp = self._pack_subpix(w1, mid, mid, True).data_ptr()
        d.cbn_total = off
        d.cbn_w = self._dev_t(torch.cat(scale_cols + offset_cols, dim=1)).data_ptr()
        # attention
        (ai, aspec), = attn
        ap = 'generator.layers.%d' % ai
        C_ = aspec[1]
        d.attn_ch = C_
        names = ['snconv1x1_theta', 'snconv1x1_phi', 'snconv1x1_g', 'snconv1x1_o_conv']
        for k, nme in enumerate(names):
            w = W['%s.%s.weight' % (ap, nme)]
            o, i_ = w.shape[0], w.shape[1]
            d.att_w[k] = self._pack_conv(w, 1, o, i_, False).data_ptr()
            d.att_wt[k] = self._pack_conv(w, 1, i_, o, True).data_ptr()
        d.gamma = float(W[ap + '.gamma'].reshape(-1)[0])
        # tail
        wrgb = W['generator.conv_to_rgb.weight'][:3]
        d.rgb_w = self._pack_conv(wrgb, 9, 32, self.ch, False, thin=True).data_ptr()
        d.rgb_wt = self._pack_conv(wrgb, 9, self.ch, 16, True, thin=True).data_ptr()
        brgb = torch.zeros(32)
        brgb[:3] = W['generator.conv_to_rgb.bias'][:3]
        d.rgb_b = self._dev_t(brgb).data_ptr()

    def _set_truncation(self, truncation):
        if self.truncation == truncation:
            return
        W, d = self._w, self._desc
        means, rstds = [], []
        for bp in self._bn_prefixes:
            m, v = _bn_row(W[bp + '.running_means'], W[bp + '.running_vars'], truncation)
            means.append(m)
            rstds.append(1.0 / torch.sqrt(v + BN_EPS))
        self._cbn_mean = torch.cat(means).to(self._dev).contiguous()
        self._cbn_rstd = torch.cat(rstds).to(self._dev).contiguous()
        d.cbn_mean = self._cbn_mean.data_ptr()
        d.cbn_rstd = self._cbn_rstd.data_ptr()
        m, v = _bn_row(W['generator.bn.running_means'], W['generator.bn.running_vars'], truncation)
        s = W['generator.bn.weight'] / torch.sqrt(v + BN_EPS)
        t = W['generator.bn.bias'] - m * s
        self._tail_s = s.to(self._dev).contiguous()
        self._tail_t = t.to(self._dev).contiguous()
        d.tail_s = self._tail_s.data_ptr()
        d.tail_t = self._tail_t.data_ptr()
        self.truncation = truncation

    # -------------------------------------------------------------- execution
    lanes_ok = True              # per-lane workspaces: chunks of one step may run on several streams
    _ws, _ws_bytes, _ws_B = _lane_attr('ws'), _lane_attr('ws_bytes'), _lane_attr('ws_B')
    _img16, _dimg16 = _lane_attr('img16'), _lane_attr('dimg16')
    _ticket, _last_B = _lane_attr('ticket'), _lane_attr('last_B')

    def _lane_state(self):
        k = lanes.current()
        st = self._lanes.get(k)
        if st is None:
            st = self._lanes[k] = _Lane()
        return st

    def _workspace(self, B):
        # sized for the largest batch seen: a ragged last chunk must not re-allocate GBs
        # twice per step (the plan lays the arena out from the B of each call)
        if B > self._ws_B:
            nbytes = self._lib.p2l_biggan_ws_bytes(C.byref(self._desc), B)
            if nbytes == 0:
                raise N.NativeError('p2l_biggan_ws_bytes rejected batch %d' % B)
            self._ws = torch.empty(nbytes // 4, device=self._dev, dtype=torch.float32)
            self._ws_bytes = nbytes
            self._ws_B = B
            self.ws_generation += 1          # captured HIP graphs hold the old pointers
            self._img16 = torch.empty(B, 256, 256, 16, device=self._dev, dtype=torch.float32)
            self._dimg16 = torch.empty(B, 256, 256, 16, device=self._dev, dtype=torch.float32)
        return self._ws

    def _run_forward(self, z, c):
        B = z.shape[0]
        ws = self._workspace(B)
        self._last_B = B
        self._ticket += 1
        N.check(self._lib.p2l_biggan_fwd(C.byref(self._desc), N.ptr(z), N.ptr(c), B,
                                         N.ptr(ws), C.c_size_t(self._ws_bytes),
                                         N.ptr(self._img16), N.stream()), 'p2l_biggan_fwd')
        out = torch.empty(B, 3, 256, 256, device=self._dev, dtype=torch.float32)
        N.check(self._lib.p2l_nhwc16_to_nchw3(N.ptr(self._img16), N.ptr(out), B, 256, 256,
                                              N.stream()), 'p2l_nhwc16_to_nchw3')
        return out

    def _run_backward(self, B, dout):
        dout = dout.contiguous().float()
        N.check(self._lib.p2l_nchw3_to_nhwc16(N.ptr(dout), N.ptr(self._dimg16), B, 256, 256,
                                              N.stream()), 'p2l_nchw3_to_nhwc16')
        dz = torch.empty(B, self.z_dim, device=self._dev, dtype=torch.float32)
        dc = torch.empty(B, self.z_dim, device=self._dev, dtype=torch.float32)
        N.check(self._lib.p2l_biggan_bwd(C.byref(self._desc), B, N.ptr(self._ws),
                                         C.c_size_t(self._ws_bytes), N.ptr(self._img16),
                                         N.ptr(self._dimg16), N.ptr(dz), N.ptr(dc),
                                         N.stream()), 'p2l_biggan_bwd')
        return dz, dc

    def saved_activation(self, what, layer=0):
        """test hook: view of a saved NHWC activation inside the workspace."""
        off = C.c_size_t(0)
        shape = (C.c_int32 * 4)()
        N.check(self._lib.p2l_biggan_ws_lookup(C.byref(self._desc), self._last_B, what, layer,
                                               C.byref(off), shape), 'p2l_biggan_ws_lookup')
        n = shape[0] * shape[1] * shape[2] * shape[3]
        return self._ws[off.value:off.value + n].view(*list(shape))

    # ------------------------------------------------------------- public API
    def get_class_embedding(self, cls):
        with torch.no_grad():
            if type(cls) == int:
                c = torch.zeros(1, 1000, device=self._dev).float()
                c[:, cls] = 1
            elif len(cls.size()) == 2:
                c = cls.to(self._dev)
            else:
                raise ValueError
            return self.embeddings(c)

    def forward(self, z=None, c=None, truncation=1.0):
        assert 0 < truncation <= 1

        assert len(z.size()) == 2, 'expected z to be 2D'
        assert len(c.size()) == 2, 'expected c to be 2D'
        assert c.size(1) == 128, \
            'expected c to have dim (?, 128) but got {}'.format(c.size())
        self._set_truncation(truncation)
        return _BigGANFn.apply(z, c, self)

    def cuda(self, device=None):   # parameters already live on the ROCm device
        return self
