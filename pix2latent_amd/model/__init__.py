from .biggan import BigGAN
