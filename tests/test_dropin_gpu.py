"""The drop-in surface driven the way the reference's examples drive it -- through the `pix2latent`
alias, with the examples' own call sequences (written from the API, schedules shortened, temporary
PIL-made images standing in for examples/images/*):

  * /root/reference/examples/invert_biggan_basincma.py:37-125
  * /root/reference/examples/invert_stylegan2_cars_hybrid_ng.py:33-131 (nn.DataParallel(model))
  * /root/reference/examples/invert_biggan_with_transform.py:35-226 (SpatialTransform search, then
    a latent optimizer with the transforms registered)
  * /root/reference/pix2latent/edit/editor.py:16-22 (load_result of the saved variables)

Asserted: the structure the scripts consume (vars.input.z.data[i], [[steps, {'loss': ...}]], the
out[-1] image, opt.tracked, files on disk)."""
import os
import os.path as osp
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_images(tmp_path, h=200, w=300):
    """a photo-like RGB jpg and its object mask (white blob on black), as examples/images/ holds"""
    from PIL import Image
    ys, xs = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing='ij')
    rgb = np.stack([np.sin(7 * xs) * .5 + .5, np.cos(5 * ys) * .5 + .5, xs * ys], -1)
    fp, mask_fp = str(tmp_path / 'dog-example-153.jpg'), str(tmp_path / 'dog-example-153-mask.jpg')
    Image.fromarray((rgb * 255).astype(np.uint8)).save(fp, quality=95)
    m = (((xs - .5) / .3) ** 2 + ((ys - .5) / .35) ** 2 <= 1).astype(np.uint8) * 255
    Image.fromarray(np.stack([m, m, m], -1)).save(mask_fp, quality=95)
    return fp, mask_fp


def test_invert_biggan_basincma_sequence(dev, tmp_path):
    warnings.simplefilter('ignore')
    from pix2latent.model import BigGAN
    from pix2latent import VariableManager, save_variables
    from pix2latent.optimizer import BasinCMAOptimizer
    from pix2latent.utils import image
    import pix2latent.loss_functions as LF
    import pix2latent.utils.function_hooks as hook
    import pix2latent.distribution as dist

    fp, mask_fp = _write_images(tmp_path)
    truncate, lr = 2.0, 0.05
    model = BigGAN().cuda().eval()
    var_manager = VariableManager()
    loss_fn = LF.ProjectionLoss()                      # (alex, beta = 10: the examples' default)
    target = image.read(fp, as_transformed_tensor=True, im_size=256)
    weight = image.read(mask_fp, as_transformed_tensor=True, im_size=256)
    weight = ((weight + 1.) / 2.).clamp_(0.3, 1.0)
    assert target.shape == (3, 256, 256) and -1 <= target.min() and target.max() <= 1
    class_lbl = 153
    save_dir = str(tmp_path / 'results')

    var_manager.register(variable_name='z', shape=(128,), grad_free=True,
                         distribution=dist.TruncatedNormalModulo(sigma=1.0, trunc=truncate),
                         var_type='input', learning_rate=lr, hook_fn=hook.Clamp(truncate))
    var_manager.register(variable_name='c', shape=(128,),
                         default=model.get_class_embedding(class_lbl)[0],
                         var_type='input', learning_rate=0.01)
    var_manager.register(variable_name='target', shape=(3, 256, 256), requires_grad=False,
                         default=target, var_type='output')
    var_manager.register(variable_name='weight', shape=(3, 256, 256), requires_grad=False,
                         default=weight, var_type='output')

    opt = BasinCMAOptimizer(model, var_manager, loss_fn, max_batch_size=9, log=False)
    vars, out, loss = opt.optimize(meta_steps=2, grad_steps=2, last_grad_steps=3)

    # what the script does with the result (:113-125)
    vars.loss = loss
    os.makedirs(save_dir, exist_ok=True)
    save_variables(osp.join(save_dir, 'vars.npy'), vars)
    image.save(osp.join(save_dir, 'target.jpg'), target)
    image.save(osp.join(save_dir, 'mask.jpg'), image.binarize(weight))
    image.save(osp.join(save_dir, 'out.jpg'), out[-1])
    np.save(osp.join(save_dir, 'tracked.npy'), opt.tracked)
    for f in ('vars.npy', 'target.jpg', 'mask.jpg', 'out.jpg', 'tracked.npy'):
        assert osp.getsize(osp.join(save_dir, f)) > 0

    # structure
    pop = 18                                           # 4 + floor(3 ln 128)
    assert len(vars.input.z.data) == pop and vars.input.z.data[0].shape == (128,)
    assert len(vars.input.c.data) == pop
    assert isinstance(loss, list) and len(loss) == 1 and loss[0][0] == 2 * 2 + 3
    final = np.asarray(loss[0][1]['loss'])
    assert final.shape == (pop,) and np.isfinite(final).all()
    assert torch.is_tensor(out[-1]) and out[-1].dim() == 3 and out[-1].shape[0] == 3
    tr = opt.tracked
    # one entry per step() call: 2 x 2 + 3 Adam steps and the re-score in front of each of the 2 tells
    assert set(tr) == {'z', 'c'} and len(tr['z']) == 2 * 2 + 3 + 2
    assert tr['z'][0].shape == (pop, 128) and not tr['z'][0].is_cuda
    # (tracked BEFORE the hooks of the step run, as base_optimizer.py:87-88 does: the first entry of a
    #  generation is the asked population itself)
    assert all(torch.isfinite(t).all() for t in tr['z'])

    # edit/editor.py:16-22 load_result on the saved file
    var = np.load(osp.join(save_dir, 'vars.npy'), allow_pickle=True).item()
    idx = int(np.argmin(var.loss[-1][1]['loss']))
    z = var.input.z.data[idx].unsqueeze(0).float().cuda()
    c = var.input.c.data[idx].unsqueeze(0).float().cuda()
    with torch.no_grad():
        best = model(z, c)[0]
    assert best.shape == (3, 256, 256) and torch.isfinite(best).all()
    # (the recorded losses are those of the last forward pass, i.e. of the latents BEFORE the last
    #  Adam update: the saved candidate scores close to, not exactly, its recorded loss)
    with torch.no_grad():
        l_best = loss_fn(best.unsqueeze(0), target.unsqueeze(0).cuda(), weight.unsqueeze(0).cuda())
    assert np.isfinite(float(l_best)) and float(l_best) < float(final.max())


def test_invert_stylegan2_cars_hybrid_ng_sequence(dev, tmp_path):
    warnings.simplefilter('ignore')
    import torch.nn as nn
    from pix2latent.model.stylegan2 import StyleGAN2
    from pix2latent import VariableManager, save_variables
    from pix2latent.optimizer import HybridNevergradOptimizer
    from pix2latent.utils import image
    import pix2latent.loss_functions as LF
    import pix2latent.utils.function_hooks as hook
    import pix2latent.distribution as dist
    from PIL import Image

    # (the 480x360 car photo of the examples, synthesised)
    ys, xs = np.meshgrid(np.linspace(0, 1, 360), np.linspace(0, 1, 480), indexing='ij')
    car = np.stack([xs, ys, np.sin(9 * xs * ys) * .5 + .5], -1)
    filename = str(tmp_path / 'car-example.png')
    Image.fromarray((car * 255).astype(np.uint8)).save(filename)

    model = StyleGAN2(model='cars', search='z')        # 512 x 512, as the script builds it
    target = image.read(filename, as_transformed_tensor=True, im_size=512, transform_style='stylegan')
    assert target.shape == (3, 512, 512)
    assert torch.all(target[:, :60] == -1.0) and torch.all(target[:, -60:] == -1.0)   # the padded bars
    loss_mask = torch.zeros((3, 512, 512))
    loss_mask[:, 64:-64, :].data += 1.0
    weight = loss_mask

    model = nn.DataParallel(model)
    loss_fn = LF.ProjectionLoss()
    var_manager = VariableManager()
    var_manager.register(variable_name='z', shape=(512,), default=None, grad_free=True,
                         distribution=dist.TruncatedNormalModulo(sigma=1.0, trunc=2.0),
                         var_type='input', learning_rate=0.05,
                         hook_fn=hook.Compose(hook.NormalPerturb(sigma=0.05), hook.Clamp(trunc=2.0)))
    for name, default in (('target', target), ('weight', weight), ('loss_mask', loss_mask)):
        var_manager.register(variable_name=name, shape=(3, 512, 512), requires_grad=False,
                             default=default, var_type='output')

    opt = HybridNevergradOptimizer('CMA', model, var_manager, loss_fn, max_batch_size=9, log=False)
    opt.log_resize_factor = 0.5
    vars, out, loss = opt.optimize(num_samples=4, meta_steps=1, grad_steps=2, last_grad_steps=2)

    vars.loss = loss
    save_dir = str(tmp_path / 'results_sg2')
    os.makedirs(save_dir, exist_ok=True)
    save_variables(osp.join(save_dir, 'vars.npy'), vars)
    image.save(osp.join(save_dir, 'out.jpg'), out[-1])
    np.save(osp.join(save_dir, 'tracked.npy'), opt.tracked)
    assert len(vars.input.z.data) == 4 and vars.input.z.data[0].shape == (512,)
    assert loss[0][0] == 1 * 2 + 2
    final = np.asarray(loss[0][1]['loss'])
    assert final.shape == (4,) and np.isfinite(final).all()
    assert out[-1].shape[0] == 3 and out[-1].shape[-1] >= 512
    assert len(opt.tracked['z']) == 1 * 2 + 2 + 1 and opt.tracked['z'][0].shape == (4, 512)


def test_invert_biggan_with_transform_sequence(dev, tmp_path):
    warnings.simplefilter('ignore')
    from pix2latent.model import BigGAN
    from pix2latent import VariableManager
    from pix2latent.optimizer import GradientOptimizer
    from pix2latent.transform import TransformBasinCMAOptimizer, SpatialTransform
    from pix2latent.utils import image
    import pix2latent.loss_functions as LF
    import pix2latent.utils.function_hooks as hook
    import pix2latent.distribution as dist

    fp, mask_fp = _write_images(tmp_path)
    model = BigGAN().cuda().eval()
    loss_fn = LF.ProjectionLoss()
    target = image.read(fp, as_transformed_tensor=True, im_size=256)
    mask = image.read(mask_fp, as_transformed_tensor=True, im_size=256)
    weight = ((mask + 1.) / 2.).clamp_(0.3, 1.0)

    var_manager = VariableManager()
    var_manager.register(variable_name='z', shape=(128,),
                         distribution=dist.TruncatedNormalModulo(sigma=1.0, trunc=2.0),
                         var_type='input', learning_rate=0.05, hook_fn=hook.Clamp(2.0))
    var_manager.register(variable_name='c', shape=(128,), default=model.get_class_embedding(153)[0],
                         var_type='input', learning_rate=0.01)
    var_manager.register(variable_name='target', shape=(3, 256, 256), requires_grad=False,
                         default=target, var_type='output')
    var_manager.register(variable_name='weight', shape=(3, 256, 256), requires_grad=False,
                         default=weight, var_type='output')

    # ---- optimize (transformation), :103-147
    target_transform_fn = SpatialTransform(pre_align=mask)
    weight_transform_fn = SpatialTransform(pre_align=mask)
    tranform_params = target_transform_fn.get_default_param(as_tensor=True)
    assert tuple(tranform_params.size()) == (3,)
    var_manager.register(variable_name='t', shape=tuple(tranform_params.size()), requires_grad=False,
                         var_type='transform', grad_free=True)
    t_opt = TransformBasinCMAOptimizer(model, var_manager, loss_fn, max_batch_size=8, log=False)
    t_opt.register_transform(target_transform_fn, 't', 'target')
    t_opt.register_transform(weight_transform_fn, 't', 'weight')
    t_opt.set_variable_propagation('z')
    t_vars, (t_out, t_target, t_candidate), t_loss = t_opt.optimize(meta_steps=2, grad_steps=2)
    save_dir = str(tmp_path / 'results_t')
    os.makedirs(save_dir, exist_ok=True)
    image.save(osp.join(save_dir, 'transform_out.jpg'), t_out[-1])
    image.save(osp.join(save_dir, 'transform_target.jpg'), t_target[-1])
    image.save(osp.join(save_dir, 'transform_candidate.jpg'), t_candidate)
    np.save(osp.join(save_dir, 'transform_tracked.npy'), {'t': t_opt.transform_tracked})
    assert len(t_vars.transform.t.data) == 7          # 4 + floor(3 ln 3)
    assert t_candidate.shape == (3, 256, 256)
    assert len(t_opt.transform_tracked) == 2 and t_opt.transform_tracked[0].shape == (7, 3)
    t = t_opt.get_candidate()
    assert t.shape == (3,) and torch.isfinite(t).all()

    assert var_manager.edit_variable('t', {'default': t, 'grad_free': False})
    assert var_manager.edit_variable('z', {'learning_rate': 0.05})
    del t_opt, t_vars, t_out, t_target, t_candidate, t_loss
    model.zero_grad()
    torch.cuda.empty_cache()

    # ---- optimize (latent), the 'adam' branch :163-172
    var_manager.edit_variable('z', {'grad_free': False})
    opt = GradientOptimizer(model, var_manager, loss_fn, max_batch_size=9, log=False)
    opt.register_transform(target_transform_fn, 't', 'target')
    opt.register_transform(weight_transform_fn, 't', 'weight')
    vars, out, loss = opt.optimize(num_samples=3, grad_steps=4)
    assert len(vars.input.z.data) == 3 and loss[0][0] == 4
    l = np.asarray(loss[0][1]['loss'])
    assert l.shape == (3,) and np.isfinite(l).all()
    # the target the latents were fitted to is the WARPED one (apply_transform on step 0)
    warped = torch.stack(list(vars.output.target.data)).cpu()
    assert (warped - target.unsqueeze(0)).abs().max().item() > 1e-3 or torch.allclose(t, torch.tensor([1., 0., 0.]))
    image.save(osp.join(save_dir, 'out.jpg'), out[-1])
