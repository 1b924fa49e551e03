"""Mirror of the reference's models/monodepth2.py: `decoder(encoder(x))` (reference :8-16).  When both halves are HIP mirrors the
whole thing runs as ONE engine tape (features never leave NHWC, no autograd hand-off in between); any other encoder/decoder
pair falls back to plain composition of the two modules."""
import torch.nn as nn

from ._common import run_net


class monodepth2(nn.Module):
    def __init__(self, encoder, decoder):
        super(monodepth2, self).__init__()
        self.encoder = encoder
        self.decoder = decoder

    def _fused(self):
        return hasattr(self.encoder, "_hip_features") and hasattr(self.decoder, "_hip_decode")

    def forward(self, x):
        if not self._fused():
            return self.decoder(self.encoder(x))
        outs = run_net(self, x)
        return outs if self.training else outs[0]

    def _hot_parameters(self):
        return list(self.encoder._hot_parameters()) + list(self.decoder._hot_parameters())

    def _hip_forward(self, tape, sink, x):
        feats = self.encoder._hip_features(tape, sink, x)
        outs = self.decoder._hip_decode(tape, sink, feats)
        scales = sorted(self.decoder.scales)
        if self.training:
            return [outs[scales.index(s)] for s in (0, 1, 2, 3)]
        return [outs[scales.index(0)]]
