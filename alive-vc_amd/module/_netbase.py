"""Shared base of the three network classes: reference-compatible checkpoint I/O
(state_dict keys/shapes of module/schema.py) + one-time packing into the table the
.so consumes.  Not an nn.Module: inference only, no autograd."""
from collections import OrderedDict

import torch

from . import _native as nat
from . import synthetic


class PackedNet:
    MODEL_ID = -1

    @staticmethod
    def _schema():
        raise NotImplementedError

    @staticmethod
    def _pack(sd):
        raise NotImplementedError

    def __init__(self, seed=None):
        self._schema_d = self._schema()
        if seed is None:
            self._sd = OrderedDict((k, torch.zeros(shape)) for k, (shape, _, _) in self._schema_d.items())
        else:
            self._sd = synthetic.make_state_dict(self._schema_d, seed, self.PREFIX)
        self._device = torch.device("cpu")
        self._table = None
        self._ws = nat.Workspace()

    # --- nn.Module-like surface used by the reference's scripts ---
    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device == self._device:
            return self                                # keeps the packed table
        self._device = device
        self._sd = OrderedDict((k, v.to(self._device)) for k, v in self._sd.items())
        self._table = None
        return self

    def eval(self):
        return self

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self._sd.items())

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._schema_d if k not in sd]
        extra = [k for k in sd if k not in self._schema_d]
        if strict and (missing or extra):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:5]}, unexpected {extra[:5]}")
        for k, (shape, _, _) in self._schema_d.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shape)}")
                self._sd[k] = sd[k].detach().to(self._device, torch.float32).contiguous()
        self._table = None
        return self

    def table(self):
        if self._table is None:
            if self._device.type != "cuda":
                raise RuntimeError("this network runs on the MI355X only: call .to('cuda') first (no CPU path)")
            self._table = nat.WeightTable(self.MODEL_ID, self._pack(self._sd))
            # the packing ran as torch ops on the current stream; window batches on side streams (module/pipeline.py)
            # read the table without any dependency on that stream, so it must be complete before it is handed out
            torch.cuda.current_stream(self._device).synchronize()
        return self._table
