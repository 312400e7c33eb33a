"""Minimal attribute dictionary (stands in for the reference's `easydict`
dependency, pix2latent/variable_manager.py:12,44,240): nested dicts become
attribute-accessible, `hasattr(v, 'transform')` style checks work, it pickles
like a plain dict."""


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super(AttrDict, self).__init__()
        if d is None:
            d = {}
        for k, v in dict(d, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        return v

    def __setitem__(self, k, v):
        super(AttrDict, self).__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)

    def __reduce__(self):
        return (AttrDict, (dict(self),))


edict = AttrDict
