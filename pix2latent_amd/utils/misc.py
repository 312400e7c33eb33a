"""Small helpers mirroring the names used from pix2latent/utils/misc.py
(set_seed :17-22, HiddenPrints :63-77, cprint :117-121, progress_print
:132-138).  Host-side only."""
import os
import random
import sys
import warnings

import numpy as np
import torch


def set_seed(i):
    """seeds torch / numpy / random (and, unlike the reference, the built-in
    CMA-ES takes its own `seed` argument)."""
    torch.manual_seed(i)
    np.random.seed(i)
    random.seed(i)
    return


def to_numpy(x):
    return x.detach().cpu().numpy()


def to_onehot(c):
    onehot = torch.zeros((1, 1000))
    onehot[:, c] = 1.0
    return onehot


class HiddenPrints:
    """ `with HiddenPrints(): ...` silences stdout """

    def __enter__(self):
        self._original_stdout = sys.stdout
        sys.stdout = open(os.devnull, 'w')

    def __exit__(self, exc_type, exc_val, exc_tb):
        sys.stdout.close()
        sys.stdout = self._original_stdout


_COLORS = {
    'b': '\033[94m', 'blue': '\033[94m', 'g': '\033[92m', 'green': '\033[92m',
    'y': '\033[93m', 'yellow': '\033[93m', 'r': '\033[91m', 'red': '\033[91m',
    'c': '\033[36m', 'cyan': '\033[36m', 'p': '\033[95m', 'pink': '\033[95m',
    'o': '\033[33m', 'orange': '\033[33m', 'lc': '\033[96m', 'lightcyan': '\033[96m',
    'lb': '\033[94m', 'lightblue': '\033[94m',
}
_END = '\033[0m'


def color_str(string, color):
    if color not in _COLORS:
        warnings.warn('Unknown color {}'.format(color))
        return string
    return '{}{}{}'.format(_COLORS[color], string, _END)


def cprint(print_str, color):
    print(color_str(print_str, color))
    return


def progress_print(phase, i, j, color='c', t=None):
    per = (100. * i) / j
    msg = '({}) progress {:.0f}% [{}/{}]'.format(color_str(phase, color), per, i, j)
    if t is not None:
        msg += ' ({:.3f} sec/iter)'.format(t)
    print(msg)
    return
