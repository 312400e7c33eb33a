"""StyleGAN2 generator on the native MI355X path.

Host-side mirror of `pix2latent.model.StyleGAN2` (reference
pix2latent/model/stylegan2.py:66-138): `StyleGAN2(model='cars'|'ffhq', search='z'|'w+')`,
`__call__(z, noises=None, truncation=1.0)`, `forward_z`, `forward_w`, `reshape_noise`,
attributes `noise_shape`, `mean_latent` (z search) / `latent_mean`, `latent_std` (w+
search), `im_res`.  The arithmetic is libp2l_hip (`p2l_sg2_mapping_*`,
`p2l_sg2_synthesis_*`) instead of the git-cloned rosinality model + its two CUDA
extensions.  Generator parameters are frozen: gradients go to the latents and, in w+
search, to the noise inputs.

Weights: `$P2L_STYLEGAN2_<MODEL>_WEIGHTS` (a torch-loadable rosinality checkpoint with a
'g_ema' entry) if set, else seeded random-init tensors of the same architecture (no
network access here).  As in the reference, z-search draws fresh per-layer noise on
every forward unless `noises` is given.
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

IM_DIM = {'cars': 512, 'ffhq': 1024}
LR_MLP = 0.01


class _Lane(object):
    """the device scratch of one execution lane (lanes.py)"""

    def __init__(self):
        self.ws, self.ws_bytes, self.ws_B = None, 0, -1
        self.img16 = self.dimg16 = None
        self.ticket = 0


def _lane_attr(name):
    return property(lambda self: getattr(self._lane_state(), name),
                    lambda self, v: setattr(self._lane_state(), name, v))


class _SynthFn(torch.autograd.Function):
    """latent [B, n_latent, 512] (+ layer-major noise) -> image [B,3,S,S]"""

    @staticmethod
    def forward(ctx, latent, noise_lm, model, want_dnoise):
        B = latent.shape[0]
        latent = latent.contiguous().float()
        noise_lm = noise_lm.contiguous().float()
        model._ensure_ws(B)
        lib, S = model._lib, model.im_res
        model._ticket += 1
        N.check(lib.p2l_sg2_synthesis_fwd(C.byref(model._desc), N.ptr(latent), N.ptr(noise_lm), B,
                                          N.ptr(model._ws), C.c_size_t(model._ws_bytes),
                                          N.ptr(model._img16), N.stream()), 'p2l_sg2_synthesis_fwd')
        out = torch.empty(B, 3, S, S, device=latent.device, dtype=torch.float32)
        N.check(lib.p2l_nhwc16_to_nchw3(N.ptr(model._img16), N.ptr(out), B, S, S, N.stream()),
                'p2l_nhwc16_to_nchw3')
        ctx.model, ctx.ticket, ctx.want_dnoise = model, model._ticket, want_dnoise
        ctx.lane = lanes.current()
        ctx.save_for_backward(latent, noise_lm)
        return out

    @staticmethod
    def backward(ctx, dout):
        with lanes.use(ctx.lane):        # (the saved activations live in the forward's lane)
            return _SynthFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        model = ctx.model
        latent, noise_lm = ctx.saved_tensors
        if model._ticket != ctx.ticket:
            raise N.NativeError('StyleGAN2 workspace was reused by a later forward before '
                                'backward(); use one model object per in-flight graph')
        B, S, lib = latent.shape[0], model.im_res, model._lib
        N.check(lib.p2l_nchw3_to_nhwc16(N.ptr(dout.contiguous().float()), N.ptr(model._dimg16), B, S,
                                        S, N.stream()), 'p2l_nchw3_to_nhwc16')
        dlatent = torch.empty_like(latent)
        dnoise = torch.empty_like(noise_lm) if ctx.want_dnoise else None
        N.check(lib.p2l_sg2_synthesis_bwd(C.byref(model._desc), N.ptr(latent), N.ptr(noise_lm), B,
                                          N.ptr(model._ws), C.c_size_t(model._ws_bytes),
                                          N.ptr(model._dimg16), N.ptr(dlatent), N.ptr(dnoise),
                                          N.stream()), 'p2l_sg2_synthesis_bwd')
        return dlatent, dnoise, None, None


class _NoiseLayoutFn(torch.autograd.Function):
    """noises [B, noise_total] (the reference's variable, model/stylegan2.py:128-138 slices it per layer) ->
    layer-major flat vector for the synthesis entry points; one launch each way (p2l_sg2_noise_relayout)
    where 17 slices + cat and their autograd transposes were 1.3 ms of a 25 ms FFHQ-1024 step"""

    @staticmethod
    def forward(ctx, noises, model):
        x = noises.contiguous().float()
        assert x.dim() == 2 and x.size(1) == model._desc.noise_total      # (reference stylegan2.py:137)
        out = torch.empty(x.numel(), device=x.device, dtype=torch.float32)
        N.check(model._lib.p2l_sg2_noise_relayout(C.byref(model._desc), N.ptr(x), N.ptr(out), x.size(0), 1,
                                                  N.stream()), 'p2l_sg2_noise_relayout')
        ctx.model, ctx.B = model, x.size(0)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        dx = torch.empty(ctx.B, ctx.model._desc.noise_total, device=g.device, dtype=torch.float32)
        N.check(ctx.model._lib.p2l_sg2_noise_relayout(C.byref(ctx.model._desc), N.ptr(g), N.ptr(dx), ctx.B, 0,
                                                      N.stream()), 'p2l_sg2_noise_relayout')
        return dx, None


class _MappingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, model):
        z = z.contiguous().float()
        B, D = z.shape
        w = torch.empty_like(z)
        acts = torch.empty(9, B, D, device=z.device, dtype=torch.float32)
        N.check(model._lib.p2l_sg2_mapping_fwd(C.byref(model._desc), N.ptr(z), N.ptr(w), N.ptr(acts),
                                               B, N.stream()), 'p2l_sg2_mapping_fwd')
        ctx.model = model
        ctx.save_for_backward(z, acts)
        model._last_acts = acts              # (test hook: oracle/replay.py reads the run's decisions)
        return w

    @staticmethod
    def backward(ctx, dw):
        z, acts = ctx.saved_tensors
        B, D = z.shape
        dz = torch.empty_like(z)
        scratch = torch.empty(2, B, D, device=z.device, dtype=torch.float32)
        N.check(ctx.model._lib.p2l_sg2_mapping_bwd(C.byref(ctx.model._desc), N.ptr(z), N.ptr(acts),
                                                   N.ptr(dw.contiguous().float()), N.ptr(dz),
                                                   N.ptr(scratch), B, N.stream()),
                'p2l_sg2_mapping_bwd')
        return dz, None


class StyleGAN2(nn.Module):
    def __init__(self, model='cars', search='z', weights=None, size=None, device='cuda', seed=0,
                 wfmt=None):
        super(StyleGAN2, self).__init__()
        self._dev = torch.device(device)
        if self._dev.type != 'cuda':
            raise N.NativeError('StyleGAN2 needs a ROCm device: the generator only exists as HIP '
                                'kernels (no CPU fallback)')
        self.im_res = size if size is not None else IM_DIM[model]
        if weights is None:
            path = os.environ.get('P2L_STYLEGAN2_%s_WEIGHTS' % model.upper())
            if path:
                ck = torch.load(path, map_location='cpu')
                weights = ck['g_ema'] if 'g_ema' in ck else ck
            else:
                warnings.warn('StyleGAN2: no checkpoint available (no network); using seeded '
                              'random-init weights of the %s architecture' % model)
                weights = synthetic.stylegan2_weights(self.im_res, seed)
        self._lib = N.lib()
        self._keep = []
        self._desc = N.P2LStyleGAN2()
        self._wfmt = N.default_wfmt() if wfmt is None else wfmt
        # (P2L_AMAX=0: every fp16 x 2 launch reduces the maxima of its input itself -- model descriptor flag)
        self._desc.wfmt = self._wfmt | (N.WFMT_FLAG_NO_AMAX if N.default_no_amax() else 0)
        self._lanes = {}         # lane -> _Lane: arena + image staging of one stream (lanes.py)
        self.ws_generation = 0
        self._pack(weights)
        self.search = search
        with torch.no_grad():
            n_mean_latent = 4096
            g = torch.Generator(device='cpu').manual_seed(seed + 1)
            zs = torch.randn(n_mean_latent, 512, generator=g).to(self._dev)
            latent_out = self.mapping(zs)
            if search == 'z':
                self.mean_latent = latent_out.mean(0, keepdim=True)
            elif search == 'w+':
                self.latent_mean = latent_out.mean(0)
                latent_std = (latent_out - self.latent_mean).pow(2).sum()
                self.latent_std = (latent_std / n_mean_latent) ** 0.5
        return

    # ------------------------------------------------------------------ setup
    def _t(self, t):
        t = t.detach().to(self._dev, torch.float32).contiguous()
        self._keep.append(t)
        return t.data_ptr()

    def _pack_w(self, w, taps, n_pad, k_pad, flip, subpix=False):
        dst = N.pack_conv_weight(w.detach().to(self._dev, torch.float32), taps, n_pad, k_pad, flip,
                                 self._wfmt, subpix_mode=1 if subpix else None)
        torch.cuda.current_stream().synchronize()
        self._keep.append(dst)
        return dst.data_ptr()

    def _pack(self, W):
        d = self._desc
        S = self.im_res
        log_size = int(math.log2(S))
        d.size, d.style_dim = S, 512
        d.n_latent = log_size * 2 - 2
        for i in range(8):
            wt = W['style.%d.weight' % (i + 1)].float() * (LR_MLP / math.sqrt(512))
            d.map_w[i] = self._t(wt.t())
            d.map_b[i] = self._t(W['style.%d.bias' % (i + 1)].float() * LR_MLP)
        d.const_input = self._t(W['input.input'][0].permute(1, 2, 0))
        names = ['conv1'] + ['convs.%d' % i for i in range(2 * (log_size - 2))]
        self.noise_shape = []
        noise_off = 0
        for l, nm in enumerate(names):
            c = d.conv[l]
            w = W[nm + '.conv.weight'][0].float()                 # [out, in, 3, 3]
            cout, cin = w.shape[0], w.shape[1]
            up = (l >= 1) and (l % 2 == 1)
            res = 4 if l == 0 else 2 ** ((l + 1) // 2 + 2)
            ws = w / math.sqrt(cin * 9)
            c.cin, c.cout, c.up, c.res = cin, cout, int(up), res
            c.w = self._pack_w(ws, 9, cout, cin, False, subpix=up)
            c.wt = self._pack_w(ws, 9, cin, cout, True, subpix=up)
            c.wsq = self._t((ws ** 2).sum((2, 3)).t())            # [cin][cout]
            c.mod_w = self._t((W[nm + '.conv.modulation.weight'].float() / math.sqrt(512)).t())
            c.mod_b = self._t(W[nm + '.conv.modulation.bias'].float())
            c.act_b = self._t(W[nm + '.activate.bias'].float())
            c.noise_w = float(W[nm + '.noise.weight'].reshape(-1)[0])
            c.latent_idx = l
            c.noise_off = noise_off
            noise_off += res * res
            self.noise_shape.append([1, 1, res, res])
        d.n_conv = len(names)
        d.noise_total = noise_off
        rgb_names = ['to_rgb1'] + ['to_rgbs.%d' % j for j in range(log_size - 2)]
        for j, nm in enumerate(rgb_names):
            r = d.rgb[j]
            w = W[nm + '.conv.weight'][0].float()                 # [3, cin, 1, 1]
            cin = w.shape[1]
            ws = w / math.sqrt(cin)
            r.cin, r.res = cin, 4 * (2 ** j)
            r.after_conv = 2 * j
            r.latent_idx = 2 * j + 1
            r.w = self._pack_w(ws, 1, 32, cin, False)
            r.wt = self._pack_w(ws, 1, cin, 16, True)
            r.mod_w = self._t((W[nm + '.conv.modulation.weight'].float() / math.sqrt(512)).t())
            r.mod_b = self._t(W[nm + '.conv.modulation.bias'].float())
            b32 = torch.zeros(32)
            b32[:3] = W[nm + '.bias'].reshape(-1)
            r.bias = self._t(b32)
        d.n_rgb = len(rgb_names)
        self._noise_sizes = [s[-1] * s[-2] for s in self.noise_shape]

    lanes_ok = True              # per-lane workspaces: chunks of one step may run on several streams
    _ws, _ws_bytes, _ws_B = _lane_attr('ws'), _lane_attr('ws_bytes'), _lane_attr('ws_B')
    _img16, _dimg16, _ticket = _lane_attr('img16'), _lane_attr('dimg16'), _lane_attr('ticket')

    def _lane_state(self):
        k = lanes.current()
        st = self._lanes.get(k)
        if st is None:
            st = self._lanes[k] = _Lane()
        return st

    def _ensure_ws(self, B):
        # sized for the largest batch seen (32 samples = chunks of 9,9,9,5 alternate B)
        if B > self._ws_B:
            nbytes = self._lib.p2l_sg2_ws_bytes(C.byref(self._desc), B)
            if nbytes == 0:
                raise N.NativeError('p2l_sg2_ws_bytes rejected batch %d' % B)
            self._ws = torch.empty(nbytes // 4, device=self._dev, dtype=torch.float32)
            self._ws_bytes = nbytes
            S = self.im_res
            self._img16 = torch.empty(B, S, S, 16, device=self._dev, dtype=torch.float32)
            self._dimg16 = torch.empty(B, S, S, 16, device=self._dev, dtype=torch.float32)
            self._ws_B = B
            self.ws_generation += 1          # captured HIP graphs hold the old pointers

    def saved_activation(self, layer, B):
        """test hook: view of the post-activation output of styled conv `layer` inside the workspace
        of the last synthesis forward on B candidates, NHWC"""
        off = C.c_size_t(0)
        shape = (C.c_int32 * 4)()
        N.check(self._lib.p2l_sg2_ws_lookup(C.byref(self._desc), B, layer, C.byref(off), shape),
                'p2l_sg2_ws_lookup')
        n = shape[0] * shape[1] * shape[2] * shape[3]
        return self._ws[off.value:off.value + n].view(*list(shape))

    # -------------------------------------------------------------- pieces
    def mapping(self, z):
        """PixelNorm + 8 EqualLinear/fused-lrelu layers: z [B,512] -> w [B,512]"""
        return _MappingFn.apply(z.to(self._dev), self)

    def _noise_layer_major(self, noises, B):
        """list of [B,1,h,w] (or None -> fresh normal noise, as NoiseInjection does)"""
        if noises is None:
            noises = [torch.randn(B, 1, s[-2], s[-1], device=self._dev) for s in self.noise_shape]
        return torch.cat([n.reshape(B, -1).reshape(-1) for n in noises])

    def synthesis(self, latent, noises=None, want_dnoise=False):
        B = latent.shape[0]
        if torch.is_tensor(noises) and noises.dim() == 1:
            noise_lm = noises
        else:
            noise_lm = self._noise_layer_major(noises, B)
        return _SynthFn.apply(latent, noise_lm, self, want_dnoise)

    # ------------------------------------------------------------- public API
    def __call__(self, z, noises=None, truncation=1.0):
        if self.search == 'w+':
            return self.forward_w(z, noises)
        return self.forward_z(z, noises=noises)

    def forward_z(self, z, truncation=1.0, noises=None):
        """generator([z], truncation=1.0).clamp_(-1, 1); `noises` (list of [B,1,h,w]) makes
        the per-layer noise explicit (the reference draws it fresh on every call)."""
        w = self.mapping(z)
        latent = w.unsqueeze(1).expand(-1, self._desc.n_latent, -1)
        return self.synthesis(latent, noises)

    def forward_w(self, z, noises, truncation=1.0):
        """z: w+ latents [B, n_latent, 512] (or [B,512]); noises: [B, sum(h*w)] flat."""
        B = z.shape[0]
        latent = z if z.dim() == 3 else z.unsqueeze(1).expand(-1, self._desc.n_latent, -1)
        noise_lm = _NoiseLayoutFn.apply(noises.to(self._dev), self)
        return _SynthFn.apply(latent, noise_lm, self, bool(noises.requires_grad))

    def reshape_noise(self, z):
        st_idx = 0
        noises = []
        for d in self.noise_shape:
            en_idx = st_idx + (d[-2] * d[-1])
            noises.append(z[:, st_idx:en_idx].reshape(-1, 1, d[-2], d[-1]))
            st_idx = en_idx

        assert z.size(1) == en_idx
        return noises

    def cuda(self, device=None):
        return self
