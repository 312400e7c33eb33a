"""Pre-alignment from a mask (reference pix2latent/transform/transform_utils.py:53-117):
object bounding box of the binarised mask vs the BigGAN object statistics ->
initial [scale, tx, ty].  Host-side, tiny."""
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
