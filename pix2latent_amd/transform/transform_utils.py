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


# Where BigGAN puts its objects, as fractions of the image side (the constants of the reference's
# get_biggan_stats, transform_utils.py:87-91): centre of mass (row, column) and extent (rows, columns).
_BIGGAN_OBJECT = {'center': (137 / 255., 127 / 255.), 'size': (213 / 255., 210 / 255.)}


def get_biggan_stats():
    """(centre, size) of the typical BigGAN object, each as (row, column) image fractions"""
    return list(_BIGGAN_OBJECT['center']), list(_BIGGAN_OBJECT['size'])


def compute_stat_from_mask(mask):
    """(centre, size) of the bounding box of a binarised [C,H,W] mask, as (row, column) fractions
    of the image.  The centre is the box corner plus the INTEGER half extent (what the reference
    computes, transform_utils.py:94-102: an odd box rounds towards the corner)."""
    top, left, bottom, right = bbox_from_mask(mask)
    extent = (bottom - top, right - left)
    corner = (top, left)
    side = (mask.size(1), mask.size(2))
    center = tuple((corner[a] + extent[a] // 2) / side[a] for a in (0, 1))
    size = tuple(extent[a] / side[a] for a in (0, 1))
    return center, size


def convert_to_t(src_center, src_size, dst_center, dst_size):
    """[scale, tx, ty] of the SpatialTransform that carries an object at (src_center, src_size)
    onto (dst_center, dst_size), all in (row, column) image fractions: the scale is the size
    ratio along the object's LONGER side, the shift is twice the centre offset (grid coordinates
    span [-1, 1]) in (x, y) = (column, row) order.  float32 tensor, as the reference returns."""
    src_center, dst_center = np.asarray(src_center, dtype=np.float64), np.asarray(dst_center, dtype=np.float64)
    src_size, dst_size = np.asarray(src_size, dtype=np.float64), np.asarray(dst_size, dtype=np.float64)
    longer = int(np.argmax(src_size))
    scale = src_size[longer] / dst_size[longer]
    shift_rc = 2.0 * (src_center - dst_center)
    return torch.tensor([scale, shift_rc[1], shift_rc[0]], dtype=torch.float64).float()


def compute_pre_alignment(weight):
    """initial [scale, tx, ty] from a weight mask: the mask's object box mapped onto the place
    where BigGAN draws objects (reference transform_utils.py:53-58); numpy float32"""
    center, size = compute_stat_from_mask(binarize(weight))
    return convert_to_t(center, size, *get_biggan_stats()).numpy()


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
