"""Host-side helpers around the transformation search
(reference pix2latent/transform/transform_utils.py):

  * pre-alignment from a mask (:53-117): object bounding box of the binarised mask vs the
    BigGAN object statistics -> initial [scale, tx, ty];
  * `ComposeTransform` (:122-184): several transformations driven by ONE concatenated
    parameter vector, each slice re-weighted around that transformation's default;
  * `setup_transform_fn` (:15-50): builds the composition an example script asks for.
"""
import numpy as np
import torch

from ..utils.image import binarize


def compute_pre_alignment(weight):
    """ Precompute initialization based on BigGAN bias """
    dst_center, dst_size = get_biggan_stats()
    src_center, src_size = compute_stat_from_mask(binarize(weight))
    t = convert_to_t(src_center, src_size, dst_center, dst_size)
    return t.numpy()


def convert_to_t(src_center, src_size, dst_center, dst_size):
    """ transformation parameter that moves the object (center, size) onto the
    destination (center, size); scale follows the larger object side """
    src_center, src_size = np.array(src_center), np.array(src_size)
    dst_center, dst_size = np.array(dst_center), np.array(dst_size)

    scale_idx = np.argmax(src_size).squeeze()
    s = (src_size / dst_size)[scale_idx]
    dxy = (src_center - dst_center) * 2.
    t = np.array([s, *dxy[::-1]])
    return torch.from_numpy(t).float()


def get_biggan_stats():
    """ precomputed biggan statistics """
    center_of_mass = [137 / 255., 127 / 255.]
    object_size = [213 / 255., 210 / 255.]
    return center_of_mass, object_size


def compute_stat_from_mask(mask):
    """ Given a binarized mask 0, 1. Compute the object size and center """
    st_h, st_w, en_h, en_w = bbox_from_mask(mask)
    obj_size = obj_h, obj_w = en_h - st_h, en_w - st_w
    obj_center = (st_h + obj_h // 2, st_w + obj_w // 2)

    obj_size = (obj_size[0] / mask.size(1), obj_size[1] / mask.size(2))
    obj_center = (obj_center[0] / mask.size(1), obj_center[1] / mask.size(2))
    return obj_center, obj_size


def bbox_from_mask(mask):
    assert len(list(mask.size())) == 3, \
        'expected 3d tensor but got {}'.format(len(list(mask.size())))
    rows = (mask.mean(0).sum(1) != 0).nonzero()
    cols = (mask.mean(0).sum(0) != 0).nonzero()
    if rows.numel() > 0:
        tlc_h, brc_h = rows[0].item(), rows[-1].item()
    else:
        tlc_h, brc_h = 0, mask.size(1)  # max range if failed
    if cols.numel() > 0:
        tlc_w, brc_w = cols[0].item(), cols[-1].item()
    else:
        tlc_w, brc_w = 0, mask.size(2)
    return tlc_h, tlc_w, brc_h, brc_w


class ComposeTransform(object):
    """Chain of transformations sharing one parameter vector.

    `transform_list` holds transformations or (transformation, weight) pairs; a
    transformation owns `len(fn.t)` consecutive entries of the vector.  Different parameters
    live on different scales, so before use a slice is stretched around the transformation's
    default: `weight * (t - default) + default`.  Inversion walks the chain in the same
    order as the forward pass (as the reference does)."""

    def __init__(self, transform_list):
        assert type(transform_list) == list
        self.transform_list = [list(entry) if type(entry) in (tuple, list) else [entry, 1.0]
                               for entry in transform_list]
        self._t = [np.asarray(fn.t, dtype=np.float32) for fn, _ in self.transform_list]

    def get_param(self, as_tensor=False):
        """default parameters: list of per-transformation arrays, or one flat tensor"""
        if as_tensor:
            return torch.from_numpy(np.concatenate(self._t)).float()
        return self._t

    def get_opt_param(self):
        return np.concatenate([fn.get_opt_param() for fn, _ in self.transform_list])

    def reweight(self, t, weight, t_mean):
        return (weight * (t - t_mean)) + t_mean

    def slices(self):
        """[(transformation, weight, default, start, stop)] over the flat vector"""
        out, at = [], 0
        for (fn, w), default in zip(self.transform_list, self._t):
            out.append((fn, w, default, at, at + len(default)))
            at += len(default)
        return out

    def __call__(self, ims, t, invert=False, only_spatial=False):
        if t.size(0) == 1:
            t = t.repeat(ims.size(0), 1)
        for fn, w, default, lo, hi in self.slices():
            if only_spatial and not getattr(fn, 'is_spatial', False):
                continue
            centre = torch.from_numpy(default).type_as(t)
            ims = fn(ims, self.reweight(t[:, lo:hi], w, centre), invert=invert)
        return ims

    def __str__(self):
        return '<ComposeTransform\n\t{}\n>'.format(
            '\n\t'.join(str(fn) for fn, _ in self.transform_list))


#: colour transformations of the reference (transform/color_transform.py) are CPU PIL /
#: torchvision operations outside the hot path ("OUT")
_COLOR_NAMES = ('hue', 'gamma', 'saturation', 'brightness', 'contrast')


def setup_transform_fn(args, weight):
    """(transform_fn, t [1, K]) for an example script's options: `args.spatial_transform`
    (search the alignment), `args.align` (start from the mask's pre-alignment),
    `args.color_transform` (names; not provided by this package).  Returns (None, None) when
    nothing is requested."""
    from .spatial_transform import SpatialTransform
    chain = []
    if args.spatial_transform or args.align:
        chain.append((SpatialTransform(), 1.0))
    wanted = [c for c in _COLOR_NAMES if c in (getattr(args, 'color_transform', None) or [])]
    if wanted:
        raise NotImplementedError('colour transformations %s are outside this package '
                                  '(host-side PIL ops in the reference)' % wanted)
    if not chain:
        return None, None
    transform_fn = ComposeTransform(chain)
    t = [p.copy() for p in transform_fn.get_param()]
    if args.align:
        t[0] = np.asarray(compute_pre_alignment(weight), dtype=np.float32)
    return transform_fn, torch.from_numpy(np.concatenate(t)).unsqueeze(0).float()
