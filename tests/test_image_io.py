"""PIL-only image IO (reference pix2latent/utils/image.py:15-109): the
reader's geometry (Resize short side -> CenterCrop | pad-to-square -> Resize), value range,
collage and save round trip, on synthetic files."""
import numpy as np
import pytest
import torch

PIL = pytest.importorskip('PIL')
from PIL import Image  # noqa: E402


def _write(path, h, w):
    """R = x ramp, G = y ramp, B = 255 in a centred 20 % box"""
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    im = np.zeros((h, w, 3), np.uint8)
    im[..., 0] = (xs * 255 // max(w - 1, 1)).astype(np.uint8)
    im[..., 1] = (ys * 255 // max(h - 1, 1)).astype(np.uint8)
    im[int(h * .4):int(h * .6), int(w * .4):int(w * .6), 2] = 255
    Image.fromarray(im).save(path)
    return im


@pytest.mark.parametrize('h,w', [(96, 160), (160, 96), (128, 128)])
def test_read_biggan_style_is_resize_then_center_crop(tmp_path, h, w):
    from pix2latent_amd.utils import image
    p = str(tmp_path / 'a.png')
    _write(p, h, w)
    t = image.read(p, as_transformed_tensor=True, im_size=64, transform_style='biggan')
    assert t.shape == (3, 64, 64) and t.dtype == torch.float32
    assert -1.0 <= t.min().item() and t.max().item() <= 1.0
    # the crop is centred: the blue box stays in the middle, the ramps stay symmetric
    blue = (t[2] > 0).float()
    ys, xs = torch.nonzero(blue, as_tuple=True)
    assert abs(ys.float().mean().item() - 31.5) < 2.5 and abs(xs.float().mean().item() - 31.5) < 2.5
    if w > h:      # wide image: left/right are cropped, so the x ramp no longer reaches its ends
        assert t[0, :, 0].mean().item() > -0.8 and t[0, :, -1].mean().item() < 0.8
        assert t[1, 0, :].mean().item() < -0.9 and t[1, -1, :].mean().item() > 0.9
    same = image.read(p, as_transformed_tensor=True, im_size=64, transform_style=None)
    assert torch.equal(t, same)                      # None == 'biggan' (reference :57-63)


def test_read_stylegan_style_pads_to_square(tmp_path):
    """examples/invert_stylegan2_cars_*.py: 384x512 car -> black bars above and below"""
    from pix2latent_amd.utils import image
    p = str(tmp_path / 'car.png')
    _write(p, 96, 128)
    t = image.read(p, as_transformed_tensor=True, im_size=64, transform_style='stylegan')
    assert t.shape == (3, 64, 64)
    assert torch.all(t[:, :6, :] == -1.0) and torch.all(t[:, -6:, :] == -1.0)   # (128-96)/2 * 64/128 = 8 rows
    assert t[0, 32, -1].item() > 0.9 and t[0, 32, 0].item() < -0.9              # full width kept
    with pytest.raises(ValueError):
        image.read(p, as_transformed_tensor=True, im_size=64, transform_style='nope')


def test_grid_save_roundtrip(tmp_path):
    from pix2latent_amd.utils import image
    g = torch.Generator().manual_seed(0)
    x = torch.rand(5, 3, 16, 16, generator=g) * 2 - 1
    grid = image.to_grid(x)                          # ceil(sqrt(5)) = 3 per row
    assert grid.dim() == 3 and grid.shape[0] == 3
    assert grid.shape[2] >= 3 * 16 and grid.shape[1] >= 2 * 16
    p = str(tmp_path / 'out.png')
    assert image.save(p, x[0])
    back = image.read(p, as_transformed_tensor=True, im_size=16, transform_style='biggan')
    assert (back - x[0]).abs().max().item() < 2.0 / 255 + 1e-6   # 8-bit quantisation only
    m = torch.tensor([[0.2, 0.9995], [1.0, -1.0]])
    assert torch.equal(image.binarize(m), torch.tensor([[0., 1.], [1., 0.]]))


def test_make_gif_and_video_soft_dependencies(tmp_path):
    """utils/video.py (reference utils/video.py:14-69): the module imports without cv2 / imageio /
    skvideo; make_gif falls back to PIL; make_video names the missing package"""
    import importlib.util
    import numpy as np
    import pytest
    from PIL import Image
    from pix2latent_amd.utils import video
    frames = [np.full((8, 12, 3), v, dtype=np.float32) for v in (0.0, 0.5, 1.0)]     # [0, 1] stack
    path = str(tmp_path / 'h.gif')
    video.make_gif(path, frames, duration=0.6)
    with Image.open(path) as g:
        assert g.n_frames == 3 and g.size == (12, 8)
        g.seek(2)
        assert np.asarray(g.convert('RGB')).max() >= 250           # rescaled to 0..255
    assert video.make_video(str(tmp_path / 'h.avi'), frames) is False
    if importlib.util.find_spec('cv2') is None:
        with pytest.raises(ImportError, match='cv2'):
            video.make_video(str(tmp_path / 'h.webm'), frames, duration=1.0)
    if importlib.util.find_spec('skvideo') is None:
        with pytest.raises(ImportError, match='skvideo'):
            video.make_video(str(tmp_path / 'h.mp4'), frames)


@pytest.mark.parametrize('case', ['wide', 'tall', 'odd', 'upscale', 'square'])
def test_read_matches_the_fixture(tmp_path, case):
    """tests/golden/image_read.npz (tools/make_image_golden.py): the reference's torchvision
    composition (pix2latent/utils/image.py:15-64) restated with PIL + numpy -- short-side bilinear
    Resize -> CenterCrop -> [-1, 1], and the 'stylegan' pad-to-square -> Resize variant"""
    import os
    from pix2latent_amd.utils import image
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'image_read.npz'))
    p = str(tmp_path / (case + '.png'))
    Image.fromarray(G[case + '.input']).save(p)                  # (PNG: lossless)
    size = int(G[case + '.size'])
    for style, key in ((None, 'biggan'), ('biggan', 'biggan'), ('stylegan', 'stylegan'), ('stylegan2', 'stylegan')):
        t = image.read(p, as_transformed_tensor=True, im_size=size, transform_style=style)
        assert t.shape == (3, size, size)
        assert np.array_equal(t.numpy(), G[case + '.' + key]), (case, style)
