"""Attribute-access dict used for spectral_params / hyper_params (reference utils.py:1-8)."""


class Dict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__
