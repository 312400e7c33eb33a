"""SpatialTransform: scale + translation warp of targets / weights and the inverse
warp of generated images (reference pix2latent/transform/spatial_transform.py:10-108).

On the ROCm device the warp is one fused affine-grid + bilinear grid-sample HIP kernel
(`p2l_affine_grid_sample`) with a native backward to the images and to theta
(`p2l_affine_grid_sample_bwd`); CPU tensors (host-side preprocessing, golden tests) use the
same two torch ops the reference calls.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .transform_utils import compute_pre_alignment
from .base_transform import TransformTemplate


class _WarpFn(torch.autograd.Function):
    """F.grid_sample(ims, F.affine_grid(theta, ims.size())) on the device with its backward:
    `p2l_affine_grid_sample` / `p2l_affine_grid_sample_bwd` (d ims in gather form, d theta by a
    fixed-order reduction: no atomics, unlike ATen's grid_sampler backward)."""

    @staticmethod
    def forward(ctx, ims, theta):
        from .. import _native as N
        src = ims.contiguous().float()
        th = theta.contiguous().float().view(-1, 6)
        dst = torch.empty_like(src)
        B, C, H, W = src.shape
        N.check(N.lib().p2l_affine_grid_sample(N.ptr(src), N.ptr(th), N.ptr(dst), B, C, H, W,
                                               N.stream()), 'p2l_affine_grid_sample')
        ctx.save_for_backward(src, th)
        ctx.theta_shape = theta.shape
        return dst

    @staticmethod
    @torch.autograd.function.once_differentiable     # (a double backward raises instead of returning no gradient)
    def backward(ctx, dout):
        from .. import _native as N
        src, th = ctx.saved_tensors
        B, C, H, W = src.shape
        dout = dout.contiguous().float()
        want_src, want_th = ctx.needs_input_grad
        dsrc = torch.empty_like(src) if want_src else None
        dth = torch.empty_like(th) if want_th else None
        L = N.lib()
        nbytes = L.p2l_affine_grid_sample_bwd_ws_bytes(B, H, W) if want_th else 0
        ws = torch.empty(max(nbytes // 4, 1), device=src.device, dtype=torch.float32)
        import ctypes as C_
        N.check(L.p2l_affine_grid_sample_bwd(N.ptr(src), N.ptr(th), N.ptr(dout), N.ptr(dsrc), N.ptr(dth),
                                             B, C, H, W, N.ptr(ws), C_.c_size_t(nbytes), N.stream()),
                'p2l_affine_grid_sample_bwd')
        return dsrc, (dth.view(ctx.theta_shape) if want_th else None)


def _warp(ims, theta):
    """ims [B,C,H,W], theta [B,2,3] -> F.grid_sample(ims, F.affine_grid(theta, ims.size()))"""
    theta = theta.type_as(ims)
    assert theta.size(0) == ims.size(0), \
        'one transformation per image expected but got {} for {} images'.format(
            theta.size(0), ims.size(0))
    if ims.is_cuda:
        # the native kernel, differentiable in ims and theta (invertibility_loss, gradient-based
        # search of t): no ATen fallback on the device
        return _WarpFn.apply(ims, theta)
    # CPU tensors (host-side preprocessing, golden tests): the two torch ops the reference calls
    return F.grid_sample(ims, F.affine_grid(theta, list(ims.size()), align_corners=False),
                         align_corners=False)


class SpatialTransform(TransformTemplate):
    """
    Simple affine transformation of the image: isotropic scale and translation,
    parameter t = [s, t_x, t_y] (identity [1, 0, 0]).
    """

    def __init__(self,
                 t=[1., 0., 0.],
                 identity_t=[1., 0., 0.],
                 pre_align=None,
                 sensitivity=0.1):
        """
        Args:
            identity_t (list): the identity transformation parameter, the center of
                the parameter search. [Default: [1., 0., 0.]]
            pre_align (image): if not None, a mask image from which the initial
                alignment is computed.
            sensitivity (float): t = default_t + (sensitivity * delta_t)
        """
        self.identity_t = np.array(identity_t, dtype=np.float32)
        self.is_spatial = True       # __call__ (default + sensitivity * delta) is the base class'
        self.sensitivity = sensitivity

        self.t = t
        if pre_align is not None:
            self.t = compute_pre_alignment(pre_align)

        self._t = torch.Tensor(self.t)
        return

    def get_default_param(self, as_tensor=True):
        if as_tensor:
            return self._t
        return self.t

    def get_identity_param(self, as_tensor=True):
        if as_tensor:
            return torch.Tensor(self.identity_t)
        return self.identity_t

    @staticmethod
    def _theta(scale, shift):
        theta = torch.zeros(scale.size(0), 2, 3, dtype=scale.dtype, device=scale.device)
        theta[:, 0, 0] = scale
        theta[:, 1, 1] = scale
        theta[:, :, 2] = shift
        return theta

    def transform(self, ims, t):
        """ ims: b x c x h x w ; t: b x 3 = [scale, tx, ty] """
        return _warp(ims, self._theta(t[:, 0], t[:, 1:]))

    def invert_transform(self, ims, t):
        """ inverse of `transform` for the same t:
        invert_transform(transform(ims, t), t) ~ ims """
        return _warp(ims, self._theta(1.0 / t[:, 0], -(t[:, 1:] / t[:, :1])))

    def __str__(self):
        return 'SpatialTransform: {}'.format(super().__str__())
