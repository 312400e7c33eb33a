from .biggan import BigGAN
from .stylegan2 import StyleGAN2
