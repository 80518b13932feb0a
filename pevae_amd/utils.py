"""Pack / cast_type / kl_anneal_function with the reference's names and meaning (models/utils.py:23-43,
:204-227, :244-255).  Host-side plumbing only."""
import math

import torch

INT, LONG, FLOAT = 0, 1, 2      # models/utils.py:23-25


class Pack(dict):
    """dict with attribute access (models/utils.py:28-43)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def add(self, **kwargs):
        for k, v in kwargs.items():
            self[k] = v

    def copy(self):
        pack = Pack()
        for k, v in self.items():
            pack[k] = list(v) if type(v) is list else v
        return pack


def str2bool(v):
    return v.lower() in ("true", "1")


_DTYPES = {INT: torch.int32, LONG: torch.int64, FLOAT: torch.float32}


def cast_type(var, dtype, use_gpu):
    """models/utils.py:208-227 without the legacy torch.cuda.*Tensor type objects."""
    if dtype not in _DTYPES:
        raise ValueError("Unknown dtype")
    var = var.to(_DTYPES[dtype])
    return var.cuda() if use_gpu else var.cpu()


def kl_anneal_function(anneal_function, step, k, x0, warmup_step=0, warmup_value=0.0):
    """models/utils.py:244-255."""
    if step <= warmup_step:
        return warmup_value
    step -= warmup_step
    if anneal_function == "logistic":
        return float(1.0 / (1.0 + math.exp(-k * (step - x0))))
    elif anneal_function == "linear":
        return min(1.0, step / x0)
    elif anneal_function == "const":
        return k
