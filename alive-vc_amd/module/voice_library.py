"""Drop-in for /root/reference/module/voice_library.py:6-33.

On-disk format kept: torch.save({'tokens': float32[1, 768, M]})
(generate_voice_library.py:42).  The reference hard-wires M = 512; this class
reads M from the file so 50 k - 1 M vector banks load too, and still writes
files the reference can load when M == 512.
"""
from collections import OrderedDict

import torch

from .common import match_features


class VoiceLibrary:
    def __init__(self, num_tokens=512, hubert_dim=768):
        self.hubert_dim = hubert_dim
        self.tokens = torch.randn(1, hubert_dim, num_tokens)

    def to(self, device):
        self.tokens = self.tokens.to(device)
        return self

    def state_dict(self):
        return OrderedDict(tokens=self.tokens.detach().clone())

    def load_state_dict(self, sd, strict=True):
        t = sd["tokens"]
        if t.dim() != 3 or t.shape[0] != 1 or t.shape[1] != self.hubert_dim:
            raise ValueError(f"voice library tokens must be [1,{self.hubert_dim},M], got {tuple(t.shape)}")
        self.tokens = t.to(self.tokens.device, torch.float32).contiguous()
        return self

    def match(self, source, k=4, alpha=0.0):
        return match_features(source, self.tokens, k=k, alpha=alpha)

    def forward(self, source):
        return self.match(source)

    __call__ = forward
