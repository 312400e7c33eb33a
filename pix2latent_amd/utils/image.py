"""Image helpers needed by the optimizers and the drop-in examples (subset of
reference pix2latent/utils/image.py: read :15-64, save :67-71, to_grid :74-76,
to_image :79-109, binarize :135-145).  PIL / torch only: cv2 and torchvision are
not available in this environment and are not needed for the hot path."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def make_grid(x, nrow=8, padding=2, pad_value=0.0):
    """[B,C,H,W] -> [C, rows*(H+pad)+pad, cols*(W+pad)+pad] collage with the
    layout of torchvision.utils.make_grid (which to_grid used in the reference)."""
    if x.dim() == 3:
        x = x.unsqueeze(0)
    B, C, H, W = x.shape
    if B == 1:
        return x[0]
    xmaps = min(nrow, B)
    ymaps = int(math.ceil(float(B) / xmaps))
    hh, ww = H + padding, W + padding
    grid = x.new_full((C, hh * ymaps + padding, ww * xmaps + padding), pad_value)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= B:
                break
            grid[:, yy * hh + padding: yy * hh + padding + H,
                 xx * ww + padding: xx * ww + padding + W] = x[k]
            k += 1
    return grid


def to_grid(x):
    grid_sz = int(np.ceil(np.sqrt(x.size(0))))
    return make_grid(x, grid_sz, pad_value=-1)


def to_image(output, to_cpu=True, denormalize=True, jpg_format=True,
             to_numpy=True, cv2_format=True):
    """ Formats torch tensor in the form BCHW -> BHWC """
    is_batched = True
    if len(list(output.size())) == 3:
        output = output.unsqueeze(0)
        is_batched = False
    tmp = output.detach().float()
    if to_cpu:
        tmp = tmp.cpu()
    tmp = tmp.permute(0, 2, 3, 1)
    if denormalize:
        tmp = (tmp + 1.0) / 2.0
    if jpg_format:
        tmp = (tmp * 255).int()
    if cv2_format and output.size(1) > 1:
        tmp = tmp[:, :, :, [2, 1, 0]]
    if to_numpy:
        tmp = tmp.numpy()
    if not is_batched:
        return tmp.squeeze(0)
    return tmp


def binarize(mask, min=0.0, max=1.0, eps=1e-3):
    """ used to convert continuous valued mask to binary mask """
    if type(mask) is torch.Tensor:
        assert mask.max() <= 1 + 1e-6, mask.max()
        assert mask.min() >= -1 - 1e-6, mask.min()
        mask = (mask > 1.0 - eps).float()
        return mask.clamp_(min, max)
    elif type(mask) is np.ndarray:
        mask = (mask > 1.0 - eps).astype(float)
        return np.clip(mask, min, max, out=mask)
    return False


def read(im_path, as_transformed_tensor=False, im_size=512, transform_style=None):
    """PIL-only version of the reference reader: Resize(short side, bilinear) ->
    CenterCrop -> [-1,1] ('biggan' / None), or pad-to-square -> Resize
    ('stylegan'/'stylegan2')."""
    from PIL import Image
    im = Image.open(im_path).convert('RGB')
    w, h = im.size
    if not as_transformed_tensor:
        return im
    if transform_style in ('stylegan', 'stylegan2'):
        side = max(h, w)
        canvas = Image.new('RGB', (side, side))
        canvas.paste(im, ((side - w) // 2, (side - h) // 2))
        im = canvas.resize((im_size, im_size), Image.BILINEAR)
    elif transform_style in (None, 'biggan'):
        if w <= h:
            nw, nh = im_size, int(im_size * h / w)
        else:
            nw, nh = int(im_size * w / h), im_size
        im = im.resize((nw, nh), Image.BILINEAR)
        left, top = int(round((nw - im_size) / 2.)), int(round((nh - im_size) / 2.))
        im = im.crop((left, top, left + im_size, top + im_size))
    else:
        raise ValueError(f'unknown transformation style {transform_style}')
    t = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.
    return (t - 0.5) / 0.5


def save(save_path, im):
    from PIL import Image
    if type(im) is torch.Tensor:
        im = to_image(im, cv2_format=False)
    Image.fromarray(np.asarray(im, dtype=np.uint8)).save(save_path)
    return True


def resize_area(img_uint8, factor):
    """HWC uint8 collage resize (replaces cv2.resize INTER_AREA in log_result)."""
    t = torch.from_numpy(np.asarray(img_uint8)).permute(2, 0, 1).unsqueeze(0).float()
    t = F.interpolate(t, scale_factor=factor, mode='area')
    return t[0].permute(1, 2, 0).round().byte().numpy()
