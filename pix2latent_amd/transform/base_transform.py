"""Common interface of image transformations searched by TransformBasinCMAOptimizer
(reference pix2latent/transform/base_transform.py: `TransformTemplate` with __call__,
get_default_param, get_identity_param, transform, invert_transform).

A transformation has a default parameter vector `t` (where the search is centred), an
identity parameter, a `sensitivity` that scales the search variable into parameter units,
and a forward / inverse pair acting on image batches."""
import numpy as np
import torch


class TransformTemplate(object):

    #: True for warps of the pixel grid (ComposeTransform(only_spatial=True) keeps only those)
    is_spatial = False

    def _abstract(self, what):
        raise NotImplementedError('%s does not implement %s()' % (type(self).__name__, what))

    def __call__(self, ims, delta_t, invert=False):
        """applies the transformation at `default + sensitivity * delta_t` (or its inverse)"""
        t = self.resolve(ims, delta_t)
        return self.invert_transform(ims, t) if invert else self.transform(ims, t)

    def resolve(self, ims, delta_t):
        """search variable -> parameter, on the device / dtype of the images"""
        base = torch.as_tensor(np.asarray(self.get_default_param(as_tensor=False),
                                          dtype=np.float32)).type_as(ims)
        return base + getattr(self, 'sensitivity', 1.0) * delta_t

    def get_opt_param(self):
        """parameters a search optimises over (default: all of them)"""
        return np.asarray(self.get_default_param(as_tensor=False), dtype=np.float32)

    def get_default_param(self, as_tensor=True):
        self._abstract('get_default_param')

    def get_identity_param(self, as_tensor=True):
        self._abstract('get_identity_param')

    def transform(self, ims, t):
        self._abstract('transform')

    def invert_transform(self, ims, t):
        self._abstract('invert_transform')
