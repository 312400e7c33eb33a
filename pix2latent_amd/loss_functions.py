"""Loss functions of the pix2latent hot path on the native MI355X path.

Host-side mirror of the reference file pix2latent/loss_functions.py: same names,
constructor arguments and `__call__(output, target, weight=None, loss_mask=None)`
signatures.  The weighted L1 + LPIPS-VGG16 combination used by every inversion
example (`ProjectionLoss`, reference loss_functions.py:86-100) runs as
`p2l_projloss_fwd/bwd` in libp2l_hip:

  * target features / resized weight maps are cached per (target, weight,
    loss_mask) identity instead of being recomputed each step (reference
    loss_functions.py:142 runs the perceptual net on the target every call);
  * bilinear upsampling of the five LPIPS maps followed by the weighted spatial
    sum is evaluated as a sum at tap resolution against the adjoint-resized
    weight map (exact identity; only summation order differs).

The un-weighted / L2 variants are not on the hot path and use plain torch ops on
the tensor's own device.
"""
import ctypes as C
import os
import warnings

import torch
from torch import nn

from . import _native as N
from . import lanes
from .utils import synthetic

LPIPS_SHIFT = (-.030, -.088, -.188)
LPIPS_SCALE = (.458, .448, .450)


def l1_loss(out, target):
    """ computes loss = | x - y |"""
    return torch.abs(target - out)


def l2_loss(out, target):
    """ computes loss = (x - y)^2 """
    return ((target - out) ** 2)


def invertibility_loss(ims, target_transform, transform_params, mask=None):
    """ MSE(ims - T^{-1}(T(ims))) (reference loss_functions.py:30-38) """
    if ims.size(0) == 1:
        ims = ims.repeat(len(transform_params), 1, 1, 1)
    transformed = target_transform(ims, transform_params)
    inverted = target_transform(transformed, transform_params, invert=True)
    if mask is None:
        return torch.mean((ims - inverted) ** 2, [1, 2, 3])
    return masked_l2_loss(ims, inverted, mask)


def _masked(fn, out, target, mask):
    if mask.size(0) == 1:
        mask = mask.repeat(out.size(0), 1, 1, 1)
    if target.size(0) == 1:
        target = target.repeat(out.size(0), 1, 1, 1)
    loss = fn(out, target)
    return torch.sum(loss * mask, [1, 2, 3]) / torch.sum(mask, [1, 2, 3])


def masked_l1_loss(out, target, mask):
    return _masked(l1_loss, out, target, mask)


def masked_l2_loss(out, target, mask):
    return _masked(l2_loss, out, target, mask)


def weight_regularization(orig_model, curr_model, reg='l1', weight_dict=None):
    """ reference loss_functions.py:64-83 (not on the hot path) """
    w = 1.0
    reg_loss = 0.0
    orig_state_dict = orig_model.state_dict()
    for param_name, curr_param in curr_model.named_parameters():
        if 'bn' in param_name:
            continue
        orig_param = orig_state_dict[param_name]
        if reg == 'l1':
            l = torch.abs(curr_param - orig_param).mean()
        elif reg == 'l2':
            l = ((curr_param - orig_param) ** 2).mean()
        elif reg == 'inf':
            l = torch.max(torch.abs(curr_param - orig_param))
        if weight_dict is not None:
            w = weight_dict[param_name]
        reg_loss += w * l
    return reg_loss


# ---------------------------------------------------------------------------
# native engine shared by Reconstruction / Perceptual / Projection losses
# ---------------------------------------------------------------------------
class _VggLpipsParams(object):
    """packed VGG16 + LPIPS-lin parameters on the device (P2LVggLpips)."""
    prefix = 'vgg'

    def __init__(self, weights, device):
        self.lib = N.lib()
        self.dev = torch.device(device)
        self.keep = []
        self.desc = N.P2LVggLpips()
        self.wfmt = N.default_wfmt()       # all 13 convs are 3x3
        thin = N.default_thin() and self.wfmt != N.WFMT_F32    # first conv: 3 real input channels
        self.desc.wfmt = (self.wfmt | (N.WFMT_FLAG_THIN if thin else 0) |
                          (N.WFMT_FLAG_NO_AMAX if N.default_no_amax() else 0))
        inv_scale = torch.tensor([1.0 / s for s in LPIPS_SCALE])
        for i, (cin, cout) in enumerate(synthetic.VGG_CONVS):
            w = weights['vgg.conv%d.weight' % i].float()
            if i == 0:
                f0 = N.WFMT_BF16X3T if thin else None
                self.desc.w[i] = self._pack(w, 9, cout, 16, False, f0)
                # d scaled / d img = 1/scale per input channel: fold into the dgrad copy
                self.desc.wt[i] = self._pack(w * inv_scale.view(1, 3, 1, 1), 9, 32, cout, True, f0)
            else:
                self.desc.w[i] = self._pack(w, 9, cout, cin, False)
                self.desc.wt[i] = self._pack(w, 9, cin, cout, True)
            self.desc.b[i] = self._t(weights['vgg.conv%d.bias' % i])
        for k in range(5):
            self.desc.lin[k] = self._t(weights['lpips.lin%d.weight' % k].reshape(-1))
        s16, t16 = torch.zeros(16), torch.zeros(16)
        for c in range(3):
            s16[c] = 1.0 / LPIPS_SCALE[c]
            t16[c] = -LPIPS_SHIFT[c] / LPIPS_SCALE[c]
        self.desc.in_s = self._t(s16)
        self.desc.in_t = self._t(t16)

    def _t(self, t):
        t = t.detach().to(self.dev, torch.float32).contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def _pack(self, w, taps, n_pad, k_pad, flip, fmt=None):
        if fmt is None:
            fmt = getattr(self, 'wfmt', N.WFMT_F32) if taps == 9 else N.WFMT_F32
        dst = N.pack_conv_weight(w.detach().to(self.dev, torch.float32), taps, n_pad, k_pad, flip, fmt)
        torch.cuda.current_stream().synchronize()
        self.keep.append(dst)
        return dst.data_ptr()


class _AlexLpipsParams(object):
    """packed torchvision-AlexNet features + LPIPS-lin parameters (P2LAlexLpips)."""
    prefix = 'alex'

    def __init__(self, weights, device):
        self.lib = N.lib()
        self.dev = torch.device(device)
        self.keep = []
        self.desc = N.P2LAlexLpips()
        inv_scale = torch.tensor([1.0 / s for s in LPIPS_SCALE])
        for i, (cin, cout, k) in enumerate(synthetic.ALEX_CONVS):
            w = weights['alex.conv%d.weight' % i].float()
            if i == 0:
                self.desc.w[i] = self._pack(w, k * k, cout, 16, False)
                # direct input-gradient kernel: [k*k][3][cout], 1/scale folded in
                w3 = (w * inv_scale.view(1, 3, 1, 1)).permute(2, 3, 1, 0).reshape(k * k, 3, cout)
                self.desc.wt[i] = self._t(w3)
            else:
                self.desc.w[i] = self._pack(w, k * k, cout, cin, False)
                self.desc.wt[i] = self._pack(w, k * k, cin, cout, True)
            self.desc.b[i] = self._t(weights['alex.conv%d.bias' % i])
        for k in range(5):
            self.desc.lin[k] = self._t(weights['lpips.lin%d.weight' % k].reshape(-1))
        s16, t16 = torch.zeros(16), torch.zeros(16)
        for c in range(3):
            s16[c] = 1.0 / LPIPS_SCALE[c]
            t16[c] = -LPIPS_SHIFT[c] / LPIPS_SCALE[c]
        self.desc.in_s = self._t(s16)
        self.desc.in_t = self._t(t16)

    _t = _VggLpipsParams._t
    _pack = _VggLpipsParams._pack


class _SqueezeLpipsParams(object):
    """packed torchvision-SqueezeNet1.1 features + LPIPS-lin parameters (P2LSqueezeLpips, csrc/p2l_plan_squeeze.hip)."""
    prefix = 'squeeze'

    def __init__(self, weights, device):
        self.lib = N.lib()
        self.dev = torch.device(device)
        self.keep = []
        d = self.desc = N.P2LSqueezeLpips()
        inv_scale = torch.tensor([1.0 / s for s in LPIPS_SCALE])
        w = weights['squeeze.conv0.weight'].float()
        d.w0 = self._gpack(w, 9, 64, 16, False)
        d.b0 = self._t(weights['squeeze.conv0.bias'])
        d.wt0 = self._t((w * inv_scale.view(1, 3, 1, 1)).permute(2, 3, 1, 0).reshape(9, 3, 64))
        for i, (cin, sq, ex) in enumerate(synthetic.SQZ_FIRES):
            g = lambda part, what: weights['squeeze.fire%d.%s.%s' % (i, part, what)].float()
            # squeeze 1x1: output channels padded to one 64-wide tile (the launch stores `sq` of them)
            d.sq_w[i] = self._gpack(g('squeeze', 'weight'), 1, 64, cin, False)
            d.sq_b[i] = self._t(torch.cat([g('squeeze', 'bias').cpu(), torch.zeros(64 - sq)]))
            d.sq_wt[i] = self._gpack(g('squeeze', 'weight'), 1, cin, sq, True)
            d.e1_w[i] = self._gpack(g('expand1x1', 'weight'), 1, ex, sq, False)
            d.e1_b[i] = self._t(g('expand1x1', 'bias'))
            d.e1_wt[i] = self._gpack(g('expand1x1', 'weight'), 1, 64, ex, True)
            d.e3_w[i] = self._gpack(g('expand3x3', 'weight'), 9, ex, sq, False)
            d.e3_b[i] = self._t(g('expand3x3', 'bias'))
            d.e3_wt[i] = self._gpack(g('expand3x3', 'weight'), 9, 64, ex, True)
        for k in range(7):
            d.lin[k] = self._t(weights['lpips.lin%d.weight' % k].reshape(-1))
        s16, t16 = torch.zeros(16), torch.zeros(16)
        for c in range(3):
            s16[c] = 1.0 / LPIPS_SCALE[c]
            t16[c] = -LPIPS_SHIFT[c] / LPIPS_SCALE[c]
        d.in_s = self._t(s16)
        d.in_t = self._t(t16)

    _t = _VggLpipsParams._t

    def _gpack(self, w, taps, n_pad, k_pad, flip):
        dst = N.pack_gconv_weight(w.detach().to(self.dev, torch.float32), taps, n_pad, k_pad, flip)
        torch.cuda.current_stream().synchronize()
        self.keep.append(dst)
        return dst.data_ptr()


class _CacheSlot(object):
    """target-dependent state of one (target, weight, loss_mask) chunk."""

    def __init__(self, cache_floats, B, H, W, dev, taps=5):
        nft_off = (C.c_size_t * taps)()
        wt_off = (C.c_size_t * taps)()
        wsum_off = C.c_size_t(0)
        n = cache_floats(B, H, W, nft_off, wt_off, C.byref(wsum_off))
        self.buf = torch.empty(n, device=dev, dtype=torch.float32)
        self.desc = N.P2LLossCache() if taps == 5 else N.P2LLossCache7()
        base = self.buf.data_ptr()
        for k in range(taps):
            self.desc.nft[k] = base + 4 * nft_off[k]
            self.desc.wt[k] = base + 4 * wt_off[k]
        self.desc.wsum = base + 4 * wsum_off.value
        self.held = None
        self.B = B
        self.guard = _StreamGuard()
        self.last_use = {}           # stream -> event behind its latest forward / backward over this slot

    def used(self):
        """called behind every launch sequence that READS the slot (loss forward, loss backward)"""
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            st = torch.cuda.current_stream()
            self.last_use[st.cuda_stream] = st.record_event()

    def wait_for_readers(self):
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            st = torch.cuda.current_stream()
            for sid, ev in self.last_use.items():
                if sid != st.cuda_stream:
                    st.wait_event(ev)
        self.last_use = {}


class _StreamGuard(object):
    """Orders streams on a tensor that ONE stream filled and others read (lanes: the side streams of a step
    synchronise with the caller's stream, not with each other).  `filled()` records an event on the filling
    stream; `reader()` makes any OTHER stream wait for it once.  Inside a graph capture nothing is recorded:
    the memo entries and slots a captured step touches were made by the eager steps before it."""

    def __init__(self):
        self.event, self.seen = None, set()

    @staticmethod
    def _capturing():
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def filled(self):
        if not torch.cuda.is_available():
            return self
        st = torch.cuda.current_stream()
        self.event = None if self._capturing() else st.record_event()
        self.seen = {st.cuda_stream}
        return self

    def reader(self):
        if self.event is None or not torch.cuda.is_available():
            return
        st = torch.cuda.current_stream()
        if st.cuda_stream not in self.seen and not self._capturing():
            st.wait_event(self.event)
            self.seen.add(st.cuda_stream)


class _EngineLane(object):
    def __init__(self):
        self.ws, self.ws_bytes, self.shape = None, 0, None
        self.img16 = self.dimg16 = None
        self.fwd_ticket = 0


def _lane_attr(name):
    return property(lambda self: getattr(self._lane(), name),
                    lambda self, v: setattr(self._lane(), name, v))


class _LossEngine(object):
    """workspace + target caches for p2l_projloss_*; one per loss object.

    The cache is keyed on the identity (data_ptr, version, shape) of the target /
    weight / loss_mask tensors.  A population is evaluated in several chunks per
    step, each with its own slice of the per-sample targets, so several slots
    are kept (LRU): the VGG pass over the targets runs once per chunk and
    generation, not once per step as in the reference (loss_functions.py:142).
    """
    MAX_SLOTS = 8

    def __init__(self, vgg_params):
        self.lib = N.lib()
        self.vgg = vgg_params          # _VggLpipsParams or _AlexLpipsParams
        lib = self.lib
        self.prefix = getattr(vgg_params, 'prefix', 'vgg')     # None = L1-only engine
        if self.prefix == 'alex':
            self.f_ws, self.f_cache = lib.p2l_alexloss_ws_bytes, lib.p2l_alex_cache_floats
            self.f_prepare, self.f_fwd, self.f_bwd = (lib.p2l_alexloss_prepare, lib.p2l_alexloss_fwd,
                                                      lib.p2l_alexloss_bwd)
        elif self.prefix == 'squeeze':
            self.f_ws, self.f_cache = lib.p2l_sqzloss_ws_bytes, lib.p2l_sqz_cache_floats
            self.f_prepare, self.f_fwd, self.f_bwd = (lib.p2l_sqzloss_prepare, lib.p2l_sqzloss_fwd,
                                                      lib.p2l_sqzloss_bwd)
        else:
            self.f_ws, self.f_cache = lib.p2l_projloss_ws_bytes, lib.p2l_loss_cache_floats
            self.f_prepare, self.f_fwd, self.f_bwd = (lib.p2l_projloss_prepare, lib.p2l_projloss_fwd,
                                                      lib.p2l_projloss_bwd)
        self.res = None          # (H, W) the caches were made at
        self._lanes = {}         # lane -> _EngineLane: arena + image staging of one stream (lanes.py)
        self.slots = {}          # key -> _CacheSlot (insertion order = LRU order; equal content shares a slot)
        self.keep = {}           # key -> the tensors it identifies (kept alive)
        self.cache = None        # P2LLossCache of the slot bound by the last prepare()
        self._memo = {}
        self.generation = 0

    # per-lane scratch (the cached target features are shared: read-only while steps run)
    lanes_ok = True

    def _lane(self):
        k = lanes.current()
        st = self._lanes.get(k)
        if st is None:
            st = self._lanes[k] = _EngineLane()
        return st

    ws, ws_bytes, shape = _lane_attr('ws'), _lane_attr('ws_bytes'), _lane_attr('shape')
    _img16, _dimg16, _fwd_ticket = _lane_attr('img16'), _lane_attr('dimg16'), _lane_attr('fwd_ticket')

    def _alloc(self, B, H, W, dev):
        """workspace + image staging sized for the LARGEST chunk seen at this resolution: a
        ragged last chunk (32 samples = 9,9,9,5) alternates B every step, and re-allocating
        would also drop the cached target features of every chunk"""
        if self.res != (H, W):
            # resolution changed: caches are void, and so is every lane's scratch
            self.slots, self.keep, self._lanes, self.res = {}, {}, {}, (H, W)
        if self.shape is not None and B <= self.shape[0]:
            return
        if self.shape is not None:
            B = max(B, self.shape[0])
        nbytes = self.f_ws(B, H, W)
        if nbytes == 0:
            raise N.NativeError('loss workspace sizing rejected shape %s' % ((B, H, W),))
        self.ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        self.ws_bytes = nbytes
        self._img16 = torch.empty(B, H, W, 16, device=dev, dtype=torch.float32)
        self._dimg16 = torch.empty(B, H, W, 16, device=dev, dtype=torch.float32)
        self.shape = (B, H, W)
        self.generation += 1                         # captured HIP graphs hold the old pointers

    @property
    def img16(self):
        return self._img16

    @property
    def dimg16(self):
        return self._dimg16

    @staticmethod
    def _ident(t):
        return None if t is None else (t.data_ptr(), t._version, tuple(t.shape))

    @staticmethod
    def _conform(name, t, B, like):
        """the reference multiplies / subtracts these against output [B,3,H,W] with torch
        broadcasting (loss_functions.py:117-124,143-147): accept [H,W]-matching tensors with
        1 or 3 channels and 1 or B samples, and say so when they do not fit (the native
        kernels read 3*H*W floats per sample unconditionally)."""
        e = t
        if e.dim() == 3:
            e = e.unsqueeze(0)
        H, W = like.shape[2:]
        if e.dim() != 4 or tuple(e.shape[2:]) != (H, W) or e.size(1) not in (1, 3) or \
                e.size(0) not in (1, B):
            raise ValueError('%s of shape %s does not broadcast against the output %s'
                             % (name, tuple(t.shape), tuple(like.shape)))
        if e.size(0) != B or e.size(1) != 3:
            e = e.expand(B, 3, H, W)
        return e.contiguous().float()

    def bind(self, name, t, B, like=None):
        """-> contiguous fp32 [B,3,H,W] (None stays None, except `weight`), memoised on the
        source tensor's identity so that the target cache key stays stable."""
        if t is None:
            if name != 'weight':
                return None
            # reference returns an un-reduced map when weight is None; its only
            # consumer (closure.py:55) takes .view(b,-1).mean(1), which equals the
            # weighted form with unit weights.
            key = ('ones', B, tuple(like.shape[2:]), str(like.device))
            if self._memo.get(name, (None, None))[0] != key:
                self._memo[name] = (key, torch.ones(B, 3, like.size(2), like.size(3),
                                                    device=like.device), _StreamGuard().filled())
            # (filled on the stream of the lane that came first; the other lane's stream waits for it once)
            self._memo[name][2].reader()
            return self._memo[name][1]
        if t.dim() == 4 and tuple(t.shape) == (B, 3) + tuple(like.shape[2:]) and \
                t.is_contiguous() and t.dtype == torch.float32 and t.device == like.device:
            return t
        key = (self._ident(t), B)
        if self._memo.get(name, (None, None))[0] != key:
            self._memo[name] = (key, self._conform(name, t.to(like.device), B, like),
                                _StreamGuard().filled(), t)
        self._memo[name][2].reader()
        return self._memo[name][1]

    def _same_content(self, target, weight, loss_mask, use_lpips, B):
        """A generation of a CMA run starts from FRESH variables (reference base_cma_optimizer.py:79):
        new target / weight tensors with the old CONTENT; and the chunks of one population carry copies
        of the same default target.  Instead of running the LPIPS network over the targets again
        (round 5, tools/gen_overhead.py: the first step of a generation took 32 ms instead of 17), a slot
        whose tensors are unmodified since it was prepared and EQUAL to the new ones -- compared
        exactly, on the device, one host sync -- serves the new tensors too (a second key for the slot)."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return None                                   # (no host sync inside a graph capture)
        seen = set()
        for old_key in reversed(list(self.slots)):
            slot = self.slots[old_key]
            if id(slot) in seen:
                continue
            seen.add(id(slot))
            if getattr(slot, 'use_lpips', None) != use_lpips or slot.B != B or slot.held is None:
                continue
            # `held` may be conformed copies another lane's stream made in this very step: ordered by the
            # event recorded behind that lane's prepare
            slot.guard.reader()
            differ = None
            for new, old, ver in zip((target, weight, loss_mask), slot.held, slot.versions):
                if (new is None) != (old is None):
                    differ = True
                    break
                if new is None:
                    continue
                if old._version != ver or new.shape != old.shape or new.device != old.device:
                    differ = True                      # (modified in place since: its old content is gone)
                    break
                if new.data_ptr() == old.data_ptr():
                    continue
                d = (new != old).any()
                differ = d if differ is None else (differ | d)
            if differ is True:
                continue
            if differ is None or not bool(differ):
                return slot
        return None

    def prepare(self, out, target, weight, loss_mask, use_lpips):
        B, _, H, W = out.shape
        self._alloc(B, H, W, out.device)
        key = (self._ident(target), self._ident(weight), self._ident(loss_mask), use_lpips)
        slot = self.slots.pop(key, None)
        if slot is None:
            slot = self._same_content(target, weight, loss_mask, use_lpips, B)
        if slot is None:
            if len(self.slots) >= self.MAX_SLOTS:
                lru = next(iter(self.slots))
                old = self.slots.pop(lru)                         # evict the LRU key ...
                self.keep.pop(lru, None)
                shared = any(v is old for v in self.slots.values())
                slot = old if (old.B == B and not shared) else None   # ... and reuse its memory
                if slot is not None:
                    # ... once every stream that read its features is done with them (another lane may
                    # still be inside the forward / backward that uses the evicted chunk's slot)
                    slot.wait_for_readers()
            if slot is None:
                slot = _CacheSlot(self.f_cache, B, H, W, out.device, taps=7 if self.prefix == 'squeeze' else 5)
            vref = C.byref(self.vgg.desc) if use_lpips else None
            N.check(self.f_prepare(vref, N.ptr(target), N.ptr(weight), N.ptr(loss_mask), B, H, W,
                                   C.byref(slot.desc), N.ptr(self.ws), C.c_size_t(self.ws_bytes),
                                   N.stream()), 'p2l_%sloss_prepare' % self.prefix)
            # what the slot was prepared FROM (content comparisons), with the versions of that moment
            slot.held = (target, weight, loss_mask)
            slot.versions = tuple(None if t is None else t._version for t in slot.held)
            slot.use_lpips = use_lpips
            # another lane (stream) may be the next reader of these features
            slot.guard.filled()
        else:
            slot.guard.reader()                                    # (once per stream)
        if len(self.slots) >= 2 * self.MAX_SLOTS:                 # (keys, several may share a slot)
            lru = next(iter(self.slots))
            self.slots.pop(lru)
            self.keep.pop(lru, None)
        self.slots[key] = slot      # most recently used last
        # keep the key's tensors alive so that data_ptr identity stays meaningful
        self.keep[key] = (target, weight, loss_mask)
        self.cache = slot.desc
        return slot


def _channels(t):
    return None if t is None else (t.size(0) if t.dim() == 3 else t.size(1))


def _apply(eng, output, target, weight, loss_mask, beta, mode):
    B = output.size(0)
    # Reference broadcasting quirk (loss_functions.py:119-123): with a ONE-channel
    # `_weight = loss_mask * weight` the numerator sum(|t - o| * _weight) runs over the 3
    # image channels but the denominator sum(_weight) over one, i.e. the L1 term is 3x the
    # 3-channel-weight value.  (The LPIPS map has one channel itself, so its term is not.)
    wc, mc = _channels(weight), _channels(loss_mask)
    l1_factor = 3.0 if (wc == 1 and mc in (None, 1)) else 1.0
    if mode == 2:
        l1_factor = 1.0
    loss = _ProjLossFn.apply(output, eng.bind('target', target, B, like=output),
                             eng.bind('weight', weight, B, like=output),
                             eng.bind('loss_mask', loss_mask, B, like=output), eng,
                             float(beta) / l1_factor, mode)
    return loss if l1_factor == 1.0 else l1_factor * loss


class _ProjLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, weight, loss_mask, eng, beta, mode):
        # mode: 0 = L1 + beta*LPIPS, 1 = L1 only, 2 = LPIPS only
        lib = eng.lib
        B, _, H, W = output.shape
        out_c = output.contiguous().float()
        use_lpips = 0 if mode == 1 else 1
        slot = eng.prepare(out_c, target, weight, loss_mask, use_lpips)
        ctx.slot = slot
        N.check(lib.p2l_nchw3_to_nhwc16(N.ptr(out_c), N.ptr(eng.img16), B, H, W, N.stream()),
                'p2l_nchw3_to_nhwc16')
        loss = torch.empty(B, device=output.device, dtype=torch.float32)
        l1 = torch.empty_like(loss)
        lp = torch.empty_like(loss)
        vref = C.byref(eng.vgg.desc) if use_lpips else None
        N.check(eng.f_fwd(vref, N.ptr(eng.img16), N.ptr(target), N.ptr(weight),
                          N.ptr(loss_mask), C.byref(slot.desc), N.f32(beta),
                          use_lpips, B, H, W, N.ptr(eng.ws),
                          C.c_size_t(eng.ws_bytes), N.ptr(loss), N.ptr(l1),
                          N.ptr(lp), N.stream()), 'p2l_%sloss_fwd' % eng.prefix)
        if len(eng._lanes) > 1:          # (one stream: program order already protects an evicted slot)
            slot.used()
        ctx.eng, ctx.beta, ctx.mode = eng, beta, mode
        ctx.save_for_backward(out_c, target, weight, loss_mask if loss_mask is not None
                              else torch.empty(0))
        ctx.has_mask = loss_mask is not None
        eng._fwd_ticket += 1
        ctx.ticket = eng._fwd_ticket
        ctx.lane = lanes.current()
        eng.last_l1, eng.last_lpips = l1, lp
        if mode == 2:
            return lp
        return loss

    @staticmethod
    def backward(ctx, gloss):
        with lanes.use(ctx.lane):        # (autograd's thread: the scratch of the forward's lane)
            return _ProjLossFn._backward(ctx, gloss)

    @staticmethod
    def _backward(ctx, gloss):
        eng = ctx.eng
        lib = eng.lib
        out_c, target, weight, loss_mask = ctx.saved_tensors
        if not ctx.has_mask:
            loss_mask = None
        B, _, H, W = out_c.shape
        use_lpips = 0 if ctx.mode == 1 else 1
        if eng._fwd_ticket != ctx.ticket:
            raise N.NativeError('loss workspace was reused by a later forward before '
                                'backward(); use one loss object per in-flight graph')
        g = gloss.contiguous().float()
        if ctx.mode == 2:
            use_lpips = 2          # the LPIPS term alone (L1 accumulation switched off)
        N.check(eng.f_bwd(C.byref(eng.vgg.desc) if use_lpips else None,
                          N.ptr(eng.img16), N.ptr(target), N.ptr(weight),
                          N.ptr(loss_mask), C.byref(ctx.slot.desc), N.f32(ctx.beta),
                          use_lpips, N.ptr(g), B, H, W, N.ptr(eng.ws),
                          C.c_size_t(eng.ws_bytes), N.ptr(eng.dimg16), N.stream()),
                'p2l_%sloss_bwd' % eng.prefix)
        if len(eng._lanes) > 1:
            ctx.slot.used()
        dout = torch.empty(B, 3, H, W, device=out_c.device, dtype=torch.float32)
        N.check(lib.p2l_nhwc16_to_nchw3(N.ptr(eng.dimg16), N.ptr(dout), B, H, W, N.stream()),
                'p2l_nhwc16_to_nchw3')
        return dout, None, None, None, None, None, None


_VGG_PARAMS = {}
# net -> (packed-parameter class, synthetic weights, loader of the upstream checkpoint pair)
_LPIPS_NETS = {'vgg': (_VggLpipsParams, 'lpips_vgg_weights', 'load_lpips_vgg'),
               'alex': (_AlexLpipsParams, 'lpips_alex_weights', 'load_lpips_alex'),
               'squeeze': (_SqueezeLpipsParams, 'lpips_squeeze_weights', 'load_lpips_squeeze')}


def _vgg_params(net, weights, device):
    """packed LPIPS network parameters: 'alex' (the reference default,
    loss_functions.py:87), 'vgg' (BASELINE north_star) or 'squeeze'.  A weight dict keyed 'alex.conv*' /
    'vgg.conv*' / 'squeeze.conv0*' selects the network by itself."""
    if weights is not None:
        net = ('alex' if 'alex.conv0.weight' in weights else
               'squeeze' if 'squeeze.conv0.weight' in weights else 'vgg')
    if net in ('vgg16',):
        net = 'vgg'
    if net not in _LPIPS_NETS:
        raise NotImplementedError("lpips_net='%s': LPIPS networks with a native path are "
                                  "'alex', 'vgg' and 'squeeze' (all three lpips v0.1 ships)" % net)
    params_cls, synth, loader = _LPIPS_NETS[net]
    key = (net, id(weights), str(device), N.default_wfmt(), N.default_no_amax())
    if key not in _VGG_PARAMS:
        if weights is None:
            path = os.environ.get('P2L_LPIPS_%s_WEIGHTS' % net.upper())
            if path and ',' in path:
                # the two upstream files: torchvision backbone state_dict, lpips v0.1 linear layers
                # (reference loss_functions.py:131 -> lpips.LPIPS(net=...))
                from .utils import checkpoint
                backbone, lin = (torch.load(f.strip(), map_location='cpu') for f in path.split(',')[:2])
                w = getattr(checkpoint, loader)(backbone, lin)
            elif path:
                w = torch.load(path, map_location='cpu')
            else:
                warnings.warn('LPIPS-%s: no pretrained weights available (no network); using '
                              'seeded random-init weights of the same architecture' % net)
                w = getattr(synthetic, synth)()
        else:
            w = weights
        _VGG_PARAMS[key] = params_cls(w, device)
    return _VGG_PARAMS[key]


def _native_ok(output, target, weight):
    return (output.is_cuda and output.dim() == 4 and output.size(1) == 3 and
            target is not None)


class ProjectionLoss(nn.Module):
    """ The default loss that is used in the paper (reference loss_functions.py:86-100).

    lpips_net: 'alex' (reference default), 'vgg' or 'squeeze'; all three are native HIP plans.
    """

    def __init__(self, lpips_net='alex', beta=10, weights=None, device='cuda'):
        super().__init__()
        self.beta = beta
        self._engine = _LossEngine(_vgg_params(lpips_net, weights, device))
        self.rloss_fn = ReconstructionLoss()
        self.ploss_fn = PerceptualLoss(net=lpips_net, weights=weights, device=device)
        return

    def __call__(self, output, target, weight=None, loss_mask=None):
        return _apply(self._engine, output, target, weight, loss_mask, self.beta, 0)


class ReconstructionLoss(nn.Module):
    """ Reconstruction loss with spatial weighting (reference loss_functions.py:104-124) """

    def __init__(self, loss_type='l1'):
        super(ReconstructionLoss, self).__init__()
        if loss_type in ['l1', 1]:
            self.loss_fn = l1_loss
            self._native = True
        elif loss_type in ['l2', 2]:
            self.loss_fn = l2_loss
            self._native = False
        else:
            raise ValueError('Unknown loss_type {}'.format(loss_type))
        self._engine = None
        return

    def __call__(self, output, target, weight=None, loss_mask=None):
        if self._native and weight is not None and _native_ok(output, target, weight):
            if self._engine is None:
                self._engine = _LossEngine(None)
            return _apply(self._engine, output, target, weight, loss_mask, 0.0, 1)
        loss = self.loss_fn(output, target)
        if weight is not None:
            _weight = weight if loss_mask is None else (loss_mask * weight)
            n = torch.sum(loss * _weight, [1, 2, 3])
            d = torch.sum(_weight, [1, 2, 3])
            loss = n / d
        return loss


class PerceptualLoss(nn.Module):
    """ LPIPS loss with spatial weighting (reference loss_functions.py:127-148) """

    def __init__(self, net='vgg', use_gpu=True, weights=None, device='cuda'):
        super(PerceptualLoss, self).__init__()
        self._engine = _LossEngine(_vgg_params(net, weights, device))
        return

    def __call__(self, output, target, weight=None, loss_mask=None):
        return _apply(self._engine, output, target, weight, loss_mask, 1.0, 2)
