"""Transform interface (reference pix2latent/transform/base_transform.py)."""


class TransformTemplate():

    def __init__(self):
        return

    def __call__(self):
        """ applies transformation to the image """
        raise NotImplementedError

    def get_default_param(self):
        raise NotImplementedError

    def get_identity_param(self):
        raise NotImplementedError

    def transform(self):
        raise NotImplementedError

    def invert_transform(self):
        raise NotImplementedError
