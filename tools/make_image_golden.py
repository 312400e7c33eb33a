"""tests/golden/image_read.npz: fixture for `pix2latent.utils.image.read`
(/root/reference/pix2latent/utils/image.py:15-64).

The reference composes torchvision transforms on a PIL image; torchvision is absent here (SURVEY
F3), so the expected tensors are computed with PIL + numpy DIRECTLY FROM THE BEHAVIOUR of that
composition, spelled independently of the product code (numpy slicing for the crop, a numpy canvas
for the pad):

  None / 'biggan': Resize(im_size)       short side -> im_size, long side -> int(im_size * long / short),
                                         PIL bilinear (which low-pass filters when it shrinks)
                   CenterCrop(im_size)   top = int(round((h - s) / 2.)), left likewise
  'stylegan[2]':   Pad((l, t, r, b))     zeros; the SHORT side is padded to the long one, the odd
                                         pixel goes to the bottom / right
                   Resize(im_size)       square -> im_size x im_size
  then ToTensor (uint8 / 255, CHW) and Normalize(0.5, 0.5).

Inputs are small seeded uint8 images (both orientations, odd sizes, an up-scaling case).
Run: python tools/make_image_golden.py"""
import os

import numpy as np
from PIL import Image


def synth(h, w, seed):
    g = np.random.RandomState(seed)
    ys, xs = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing='ij')
    im = np.stack([np.sin(6.0 * xs + 2.0 * ys), np.cos(9.0 * ys - 3.0 * xs), xs * ys], -1)
    im = (im - im.min()) / (im.max() - im.min())
    im = im + 0.08 * g.randn(h, w, 3)
    return (np.clip(im, 0, 1) * 255).astype(np.uint8)


def to_tensor_norm(arr_u8):
    t = arr_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.)
    return ((t - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def resize_short_side(im, size):
    w, h = im.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    return im.resize((nw, nh), Image.BILINEAR)


def expect_biggan(arr, size):
    a = np.asarray(resize_short_side(Image.fromarray(arr), size))
    h, w = a.shape[:2]
    top, left = int(round((h - size) / 2.)), int(round((w - size) / 2.))
    return to_tensor_norm(a[top:top + size, left:left + size])


def expect_stylegan(arr, size):
    h, w = arr.shape[:2]
    side = max(h, w)
    canvas = np.zeros((side, side, 3), np.uint8)
    top, left = ((w - h) // 2, 0) if h < w else (0, (h - w) // 2)
    canvas[top:top + h, left:left + w] = arr
    return to_tensor_norm(np.asarray(Image.fromarray(canvas).resize((size, size), Image.BILINEAR)))


def main():
    out = {}
    cases = [('wide', 45, 70, 32), ('tall', 71, 44, 32), ('odd', 37, 53, 24), ('upscale', 20, 30, 48),
             ('square', 40, 40, 32)]
    for i, (name, h, w, size) in enumerate(cases):
        arr = synth(h, w, 100 + i)
        out[name + '.input'] = arr
        out[name + '.size'] = np.int32(size)
        out[name + '.biggan'] = expect_biggan(arr, size)
        out[name + '.stylegan'] = expect_stylegan(arr, size)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                       'image_read.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
