"""Drop-in alias: `import pix2latent` resolves to pix2latent_amd so that the
reference's examples/invert_*.py import lines keep working unchanged
(`from pix2latent import VariableManager`, `from pix2latent.optimizer import
BasinCMAOptimizer`, `import pix2latent.loss_functions as LF`,
`from pix2latent.model import BigGAN`, `from pix2latent.utils import image,
function_hooks as hook`, `import pix2latent.distribution as dist`)."""
import importlib
import sys

import pix2latent_amd as _impl
from pix2latent_amd import *  # noqa: F401,F403
from pix2latent_amd import distribution, VariableManager, save_variables  # noqa: F401

__version__ = _impl.__version__

for _name in ('distribution', 'variable_manager', 'loss_functions', 'optimizer', 'model',
              'utils', 'utils.image', 'utils.misc', 'utils.function_hooks', 'utils.video', 'parallel',
              'optimizer.closure', 'optimizer.base_optimizer',
              'optimizer.gradient_optimizer', 'optimizer.basincma_optimizer',
              'optimizer.cma_optimizer', 'optimizer.base_cma_optimizer',
              'optimizer.ng_optimizer', 'optimizer.hybrid_ng_optimizer',
              'optimizer.base_ng_optimizer', 'model.biggan', 'model.stylegan2', 'transform',
              'transform.spatial_transform', 'transform.transform_optimizer',
              'transform.transform_utils', 'transform.base_transform'):
    try:
        sys.modules['pix2latent.' + _name] = importlib.import_module('pix2latent_amd.' + _name)
    except Exception:  # pragma: no cover  (e.g. native library missing: raised on use)
        raise
