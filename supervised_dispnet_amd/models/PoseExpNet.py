"""Mirror of the reference's models/PoseExpNet.py (pose + explainability network) on the HIP engine.

Constructor `PoseExpNet(nb_ref_imgs=2, output_exp=False)`, `init_weights()`, state_dict keys (conv1.0 .. conv7.0, pose_pred,
upconv5.0 .. upconv1.0, predict_mask4 .. predict_mask1) and the forward contract -- `([mask1..mask4], pose)` in training
mode, `(mask1, pose)` in eval mode, masks None without `output_exp`, pose [B, nb_ref_imgs, 6] -- follow reference
models/PoseExpNet.py:20-95.  It feeds the photometric loss of BASELINE config 3.

Engine schedule: the target/reference frames are consumed as a virtual channel concat (no torch.cat copy) by the 7x7 stride-2
conv; `pose = 0.01 * mean_hw(pose_pred)` is one reduction kernel; the mask decoder's crops are the conv-transpose's output
extent, its sigmoid lives in the conv epilogue.
"""
import torch.nn as nn

from .. import engine
from .._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID_AFFINE, DN_MAX_OPERANDS
from ._common import run_net

_ENC_PLANES = (16, 32, 64, 128, 256, 256, 256)
_ENC_KERNELS = (7, 5, 3, 3, 3, 3, 3)
_DEC_PLANES = (256, 128, 64, 32, 16)


def _down(c_in, c_out, k):
    return nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=k, padding=(k - 1) // 2, stride=2), nn.ReLU(inplace=True))


def _up(c_in, c_out):
    return nn.Sequential(nn.ConvTranspose2d(c_in, c_out, kernel_size=4, stride=2, padding=1), nn.ReLU(inplace=True))


class PoseExpNet(nn.Module):
    def __init__(self, nb_ref_imgs=2, output_exp=False):
        super(PoseExpNet, self).__init__()
        self.nb_ref_imgs = nb_ref_imgs
        self.output_exp = output_exp
        c_in = 3 * (1 + nb_ref_imgs)
        for i, (c_out, k) in enumerate(zip(_ENC_PLANES, _ENC_KERNELS), start=1):
            setattr(self, "conv%d" % i, _down(c_in, c_out, k))
            c_in = c_out
        self.pose_pred = nn.Conv2d(_ENC_PLANES[6], 6 * nb_ref_imgs, kernel_size=1, padding=0)
        if output_exp:
            c_in = _ENC_PLANES[4]
            for i, c_out in zip((5, 4, 3, 2, 1), _DEC_PLANES):
                setattr(self, "upconv%d" % i, _up(c_in, c_out))
                c_in = c_out
            for i, c in zip((4, 3, 2, 1), _DEC_PLANES[1:]):
                setattr(self, "predict_mask%d" % i, nn.Conv2d(c, nb_ref_imgs, kernel_size=3, padding=1))
        self._rt = None

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        if 1 + self.nb_ref_imgs > DN_MAX_OPERANDS:
            raise NotImplementedError("more than %d reference frames" % (DN_MAX_OPERANDS - 1))
        outs = run_net(self, target_image, *ref_imgs)
        pose = outs[-1].reshape(outs[-1].shape[0], self.nb_ref_imgs, 6)
        masks = list(outs[:-1]) if self.output_exp else [None, None, None, None]
        if self.training:
            return masks, pose
        return masks[0], pose

    def _hot_parameters(self):
        return list(self.parameters())

    def _runtime(self):
        if self._rt is None:
            rt = {"pose_pred": engine.ConvLayer(self.pose_pred)}
            for i in range(1, 8):
                rt["conv%d" % i] = engine.ConvLayer(getattr(self, "conv%d" % i)[0])
            if self.output_exp:
                for i in range(1, 6):
                    rt["upconv%d" % i] = engine.ConvLayer(getattr(self, "upconv%d" % i)[0], transposed=True)
                for i in range(1, 5):
                    rt["predict_mask%d" % i] = engine.ConvLayer(getattr(self, "predict_mask%d" % i))
            self._rt = rt
        return self._rt

    def _hip_forward(self, tape, sink, *frames):
        rt = self._runtime()
        P = engine.Piece
        enc, cur = [], None
        for i in range(1, 8):
            pieces = [P(f) for f in frames] if i == 1 else [P(cur)]
            cur = engine.block_conv_act(tape, sink, pieces, rt["conv%d" % i], ACT_RELU)
            enc.append(cur)
        pose_map = engine.block_conv_act(tape, sink, [P(enc[6])], rt["pose_pred"], ACT_NONE)
        pose = engine.block_spatial_mean(tape, pose_map, 0.01)
        outs = []
        if self.output_exp:
            sizes = [(e.H, e.W) for e in enc[:4]][::-1] + [(frames[0].H, frames[0].W)]     # crop targets: conv4, conv3, conv2, conv1, input
            cur, ups = enc[4], {}
            for i, hw in zip((5, 4, 3, 2, 1), sizes):
                cur = engine.block_conv_act(tape, sink, [P(cur)], rt["upconv%d" % i], ACT_RELU, out_hw=hw)
                ups[i] = cur
            for i in (1, 2, 3, 4):
                outs.append(engine.block_conv_act(tape, sink, [P(ups[i])], rt["predict_mask%d" % i], ACT_SIGMOID_AFFINE, 1.0, 0.0))
            if not self.training:
                outs = outs[:1]
        return outs + [pose]
