"""PartialConv: mask-renormalised convolution (reference partial_conv.py:6-57), on the HIP kernels.

Same constructor and call contract as the reference class: PartialConv(*conv2d_args, multi_channel=False,
return_mask=True); forward(input, mask_in) -> (output, mask_out).  Only multi_channel=False (a 1-channel
mask, slide_winsize = kh*kw) exists on the hot path.  The reference does three extra passes per conv
(a 1-channel box-sum conv, `input*mask`, `raw*multiplier`); here the mask multiply rides in the conv's
operand gather and the renormalisation in its epilogue, so only the tiny box-sum kernel remains.
"""
import torch

from . import ops, ops_block
from .nn import Conv2d, _one


class PartialConv(Conv2d):

    def __init__(self, *args, **kwargs):
        self.multi_channel = kwargs.pop('multi_channel', False)
        self.return_mask = kwargs.pop('return_mask', True)
        if self.multi_channel:
            raise ops.P3DError('PartialConv(multi_channel=True) is not used by the reference networks and not implemented')
        super().__init__(*args, **kwargs)
        self.slide_winsize = self.kernel_size[0] * self.kernel_size[1]      # partial_conv.py:28

    def forward(self, input, mask_in, join_put=None, join_take=None):
        assert len(input.shape) == 4                                          # partial_conv.py:33
        k, stride, pad, dil = _one(self.kernel_size), _one(self.stride), _one(self.padding), _one(self.dilation)
        with torch.no_grad():
            mult, mask_out = ops.mask_count(mask_in, k, stride, pad, dil)    # partial_conv.py:35-43
        if self.bias is not None and self.bias.requires_grad and torch.is_grad_enabled() and input.dtype == torch.float16:
            raise ops.P3DError('PartialConv with a trainable bias under -half_acc: backward is not implemented (no reference network uses it)')
        if input.dtype == torch.float32 and not input.requires_grad and join_put is None and join_take is None and ops_block.stem_takes_x3(self, input, masked=True):
            # the 7x7 stride-2 stem of the partial families (partial_depthnet.py:177): the restated stem kernels with the two per-pixel factors
            output = ops_block.stem_conv(self, input, mask_in.contiguous(), mult)
        elif input.dtype == torch.float16:                                    # -half_acc: masks stay fp32 [B,1,H,W], activations NHWC fp16
            from . import ops_half
            output = ops_half.conv2d(input, self, stride, pad, dil, join_put, join_take, mask_in=mask_in.contiguous(), mult=mult)
        else:
            output = ops.conv2d(input, self.weight, self.bias, stride, pad, dil, mask_in=mask_in, mult=mult, join_put=join_put, join_take=join_take)
        if self.return_mask:
            return output, mask_out
        return output
