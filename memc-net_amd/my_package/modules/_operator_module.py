"""Shared front of the operator modules.

Every `my_package.modules.*Module` of the reference is the same thing: an `nn.Module` without parameters whose
`forward` hands its tensors to a stateless operator layer kept in `self.f`.  Here that is written once;
the per-operator files only name the layer and spell out the reference's constructor / forward signatures
(the networks call them positionally, `MEMC_Net_star.py:266-285`)."""
from torch.nn import Module


class OperatorModule(Module):
    layer = None                                   # the operator layer class, set by each subclass

    def _bind(self, *layer_args):
        self.f = type(self).layer(*layer_args)

    def extra_repr(self):
        return "layer=%s" % type(self).layer.__name__
