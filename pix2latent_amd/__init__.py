"""pix2latent_amd: MI355X-native engine for the pix2latent latent-inversion hot
path (population of latents -> frozen generator -> L1 + LPIPS loss -> backward
to the latents | ranking for CMA-ES), behind the reference's Python API
(`VariableManager`, `optimizer.*`, `loss_functions`, `model.BigGAN`)."""
from . import distribution
from .variable_manager import VariableManager, save_variables

__version__ = "0.1.0"
__all__ = ["optimizer", "utils", "model", "loss_functions", "distribution",
           "VariableManager", "save_variables"]
