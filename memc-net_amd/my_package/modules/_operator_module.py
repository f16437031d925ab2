"""Shared front of the operator modules.

Every `my_package.modules.*Module` of the reference is the same thing: an `nn.Module` without parameters whose
`forward` hands its tensors to a stateless operator layer kept in `self.f`.  Here that is written once, as a class
factory; the per-operator files only declare the layer, the tensor names of `forward` and the constructor
arguments (the networks call both positionally, `MEMC_Net_star.py:266-285`)."""
import importlib
import inspect
import sys

from torch.nn import Module


class OperatorModule(Module):
    layer = None                                   # the operator layer class
    tensors = ()                                   # names of forward's positional tensors, in order
    options = ()                                   # (name, default) pairs accepted by the constructor

    def __init__(self, *args, **kwargs):
        Module.__init__(self)
        cls = type(self)
        if len(args) > len(cls.options):
            raise TypeError("%s() takes at most %d argument(s)" % (cls.__name__, len(cls.options)))
        values = [default for _, default in cls.options]
        values[:len(args)] = args
        for key, value in kwargs.items():
            names = [name for name, _ in cls.options]
            if key not in names:
                raise TypeError("%s() got an unexpected keyword argument %r" % (cls.__name__, key))
            values[names.index(key)] = value
        for (name, _), value in zip(cls.options, values):
            setattr(self, name, value)
        self.f = cls.layer(*values)

    def extra_repr(self):
        return "layer=%s" % type(self).layer.__name__


def _signature(names, defaults=()):
    params = [inspect.Parameter("self", inspect.Parameter.POSITIONAL_OR_KEYWORD)]
    for name in names:
        params.append(inspect.Parameter(name, inspect.Parameter.POSITIONAL_OR_KEYWORD,
                                        default=dict(defaults).get(name, inspect.Parameter.empty)))
    return inspect.Signature(params)


def operator_module(name, tensors, options=()):
    """Class `name` (a `...Module`) over the layer `my_package.functions.<name minus 'Module'>Layer`, defined in
    the caller's module so that pickling and `repr` see it where the reference has it; `__init__` and `forward`
    advertise the reference's signatures (named parameters, same defaults) and accept keywords accordingly."""
    layer_name = name[:-len("Module")] + "Layer"
    layer = getattr(importlib.import_module("my_package.functions." + layer_name), layer_name)
    scope = sys._getframe(1).f_globals
    forward_sig = _signature(tensors)

    def forward(self, *args, **kwargs):
        if kwargs or len(args) != len(tensors):
            args = forward_sig.bind(self, *args, **kwargs).args[1:]      # raises the usual TypeError
        return self.f(*args)

    def __init__(self, *args, **kwargs):
        OperatorModule.__init__(self, *args, **kwargs)

    forward.__signature__ = forward_sig
    __init__.__signature__ = _signature([n for n, _ in options], options)
    return type(name, (OperatorModule,), {
        "layer": layer, "tensors": tuple(tensors), "options": tuple(options), "forward": forward,
        "__init__": __init__, "__module__": scope["__name__"], "__doc__": scope.get("__doc__")})
