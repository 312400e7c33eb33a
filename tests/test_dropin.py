"""The drop-in surface (north_star: "examples/invert_*.py are drop-in"): `import pix2latent` is an
alias package over pix2latent_amd.  CPU part: every module path and name the reference's examples
import (/root/reference/examples/*.py:1-17) resolves, and to the SAME objects as pix2latent_amd."""
import importlib
import sys

import pytest

# module -> names the examples import from it (from X import a, b / import X as Y)
EXAMPLE_IMPORTS = {
    'pix2latent': ['VariableManager', 'save_variables'],
    'pix2latent.optimizer': ['GradientOptimizer', 'CMAOptimizer', 'BasinCMAOptimizer',
                             'NevergradOptimizer', 'HybridNevergradOptimizer'],
    'pix2latent.model': ['BigGAN'],
    'pix2latent.model.stylegan2': ['StyleGAN2'],
    'pix2latent.model.biggan': ['BigGAN'],
    'pix2latent.utils': ['image', 'video', 'function_hooks'],
    'pix2latent.utils.function_hooks': ['Clamp', 'Compose', 'NormalPerturb', 'Normalize'],
    'pix2latent.utils.image': ['read', 'save', 'binarize', 'to_grid', 'to_image'],
    'pix2latent.utils.misc': ['set_seed'],
    'pix2latent.loss_functions': ['ProjectionLoss', 'ReconstructionLoss', 'PerceptualLoss',
                                  'l1_loss', 'l2_loss', 'masked_l1_loss', 'invertibility_loss'],
    'pix2latent.distribution': ['TruncatedNormalModulo', 'normal'],
    'pix2latent.transform': ['SpatialTransform', 'TransformBasinCMAOptimizer'],
    'pix2latent.variable_manager': ['VariableManager', 'split_vars', 'save_variables'],
}


@pytest.mark.parametrize('module', sorted(EXAMPLE_IMPORTS))
def test_example_imports_resolve(module):
    m = importlib.import_module(module)
    twin = importlib.import_module(module.replace('pix2latent', 'pix2latent_amd', 1))
    for name in EXAMPLE_IMPORTS[module]:
        assert hasattr(m, name), '%s has no %s' % (module, name)
        assert getattr(m, name) is getattr(twin, name)


def test_every_alias_is_the_implementation_module():
    import pix2latent  # noqa: F401
    aliases = [k for k in sys.modules if k.startswith('pix2latent.')]
    assert len(aliases) >= 20
    for k in aliases:
        twin = sys.modules.get(k.replace('pix2latent', 'pix2latent_amd', 1))
        assert twin is not None and sys.modules[k] is twin, k


def test_import_lines_of_the_examples_verbatim():
    """the import block of examples/invert_biggan_basincma.py:1-17 and
    invert_stylegan2_cars_hybrid_ng.py:8-16, executed as written"""
    ns = {}
    exec('from pix2latent.model import BigGAN\n'
         'from pix2latent.model.stylegan2 import StyleGAN2\n'
         'from pix2latent import VariableManager, save_variables\n'
         'from pix2latent.optimizer import BasinCMAOptimizer, HybridNevergradOptimizer, GradientOptimizer\n'
         'from pix2latent.utils import image, video\n'
         'from pix2latent.transform import SpatialTransform, TransformBasinCMAOptimizer\n'
         'import pix2latent.loss_functions as LF\n'
         'import pix2latent.utils.function_hooks as hook\n'
         'import pix2latent.distribution as dist\n', ns)
    assert ns['dist'].TruncatedNormalModulo(sigma=1.0, trunc=2.0) is not None
    assert callable(ns['hook'].Clamp(2.0))
