"""torch.autograd.Function wrappers of the fp16 kernels (`-half_acc`, reference depth_train.py:73-83,413-449).

Activations are torch.float16 tensors of logical shape [N, C, H, W] in torch.channels_last memory format, i.e. NHWC in memory:
what the f16 matrix-core kernels read with 16-B vectors.  Parameters stay fp32 (they are the masters the reference keeps in
`copy_params`); every convolution owns two fp16 weight images refreshed after each optimizer step (`refresh_weights`).
Weight / bias / BN-parameter gradients are produced in fp32, straight into the FlatAdam gradient buffer when there is one.
"""
import ctypes

import torch

from . import ops
from ._lib import P3DError, check, lib
from .ops import _desc, _grad_done, _grad_sink, _p, _stream, workspace

CL = torch.channels_last


def _need_half(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float16 or t.dim() != 4:
            raise P3DError('fp16 ops take 4-d torch.float16 tensors on a HIP device (got %s %s)' % (t.dtype, t.device))


def _cl(t):
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


def _empty(n, c, h, w, device):
    return torch.empty((n, c, h, w), dtype=torch.float16, device=device, memory_format=CL)


def pad8(c):
    return (c + 7) // 8 * 8


def to_half_nhwc(x, cpad=None, scale=1.0):
    """fp32 NCHW -> fp16 channels_last with the channel dimension zero-padded to cpad (input of the stem)."""
    ops._need_gpu(x)
    x = x.contiguous()
    n, c, h, w = x.shape
    cpad = pad8(c) if cpad is None else cpad
    out = _empty(n, cpad, h, w, x.device)
    check(lib().p3d_nchw_f32_to_nhwc_f16(_p(x), _p(out), n, c, h * w, cpad, scale, _stream()), 'p3d_nchw_f32_to_nhwc_f16')
    return out


class ToFloatFn(torch.autograd.Function):
    """fp16 channels_last -> fp32 NCHW (the regressor output entering the fp32 head); backward converts the gradient back."""

    @staticmethod
    def forward(ctx, x):
        _need_half(x)
        x = _cl(x)
        n, c, h, w = x.shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        check(lib().p3d_nhwc_f16_to_nchw_f32(_p(x), _p(out), n, c, h * w, 1.0, _stream()), 'p3d_nhwc_f16_to_nchw_f32')
        return out

    @staticmethod
    def backward(ctx, dy):
        return to_half_nhwc(dy.contiguous(), cpad=dy.shape[1])


def to_float(x):
    return ToFloatFn.apply(x)


class WeightImages:
    """fp16 images of one convolution's fp32 master weight: krsc [K,R,S,Cpad] and crsk [Cpad,R,S,K] (views of the model's image
    buffer when built by refresh_weights)."""

    def __init__(self, weight, need_dgrad=True, krsc=None, crsk=None):
        k, c, r, s = weight.shape
        self.c_real, self.cpad = c, pad8(c)
        self.krsc = torch.empty((k, r, s, self.cpad), dtype=torch.float16, device=weight.device) if krsc is None else krsc
        self.crsk = crsk if (crsk is not None or not need_dgrad) else torch.empty((self.cpad, r, s, k), dtype=torch.float16, device=weight.device)

    def refresh(self, weight):
        k, c, r, s = weight.shape
        check(lib().p3d_weight_images_f16(_p(weight.detach()), _p(self.krsc), _p(self.crsk), k, c, r * s, self.cpad, _stream()), 'p3d_weight_images_f16')


class _ImagePlan:
    """One fp16 buffer holding every weight image of a model and the device-side job table of the batched cast kernel."""

    def __init__(self, model, flat):
        import numpy as np
        from .nn import Conv2d
        convs = [m for m in model.modules() if isinstance(m, Conv2d)]
        rows, total = [], 0
        for m in convs:
            k, c, r, s = m.weight.shape
            cpad = pad8(c)
            n = k * r * s * cpad
            krsc_off = total
            total += (n + 7) // 8 * 8
            crsk_off = -1
            if c >= 8:                                         # the stem (3 or 1 input channels) never needs an input gradient
                crsk_off = total
                total += (n + 7) // 8 * 8
            w_off = (m.weight.data_ptr() - flat.data_ptr()) // 4
            if not (0 <= w_off and w_off + m.weight.numel() <= flat.numel()):
                raise P3DError('refresh_weights: a convolution weight does not live in the flat master buffer')
            rows.append((w_off, krsc_off, crsk_off, k, c, r * s, cpad))
        self.images = torch.empty(total, dtype=torch.float16, device=flat.device)
        table = np.array(rows, dtype=[('w', '<i8'), ('a', '<i8'), ('b', '<i8'), ('K', '<i4'), ('C', '<i4'), ('RS', '<i4'), ('Cpad', '<i4')])
        assert table.dtype.itemsize == 40
        self.table = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).to(flat.device)
        self.njobs = len(rows)
        self.flat_ptr = flat.data_ptr()
        for m, (w_off, ka, kb, k, c, rs, cpad) in zip(convs, rows):
            r, s = m.weight.shape[2:]
            n = k * rs * cpad
            krsc = self.images[ka:ka + n].view(k, r, s, cpad)
            crsk = self.images[kb:kb + n].view(cpad, r, s, k) if kb >= 0 else None
            m._h_images = WeightImages(m.weight, need_dgrad=kb >= 0, krsc=krsc, crsk=crsk)

    def run(self, flat):
        check(lib().p3d_weight_images_f16_batched(_p(flat), _p(self.images), _p(self.table), self.njobs, _stream()), 'p3d_weight_images_f16_batched')


def refresh_weights(model, flat=None):
    """(Re)build the fp16 weight images of every convolution of `model` from the fp32 masters: after construction, after
    load_state_dict and after every optimizer step (the reference's `h_param.data.copy_(c_param.data)`, depth_train.py:448-449).
    With `flat` (the FlatAdam parameter buffer the weights are views of) all images are cast by ONE kernel launch."""
    from .nn import Conv2d
    if flat is not None:
        ops._need_gpu(flat)
        plan = getattr(model, '_h_plan', None)
        if plan is None or plan.flat_ptr != flat.data_ptr():
            plan = _ImagePlan(model, flat)
            model._h_plan = plan
        plan.run(flat)
        return
    for m in model.modules():
        if isinstance(m, Conv2d):
            ops._need_gpu(m.weight)
            img = getattr(m, '_h_images', None)
            if img is None or img.krsc.device != m.weight.device:
                img = m._h_images = WeightImages(m.weight, need_dgrad=m.weight.shape[1] >= 8)
            img.refresh(m.weight)


class HConv2dFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, w, bias, images, stride, pad, dil, join_put=None, join_take=None, mask_in=None, mult=None):
        _need_half(x)
        ops._need_gpu(w, bias, mask_in, mult)
        x = _cl(x)
        n, c, h, wd = x.shape
        k, _, r, s = w.shape
        if c != images.cpad:
            raise P3DError('hconv2d: input has %d channels, the weight image expects %d' % (c, images.cpad))
        d = _desc((n, c, h, wd), (k, c, r, s), stride, pad, dil)
        y = _empty(n, k, d.Ho, d.Wo, x.device)
        with ops._Timed('fwd', d):
            check(lib().p3d_hconv2d_fwd(ctypes.byref(d), _p(x), _p(images.krsc), _p(bias), _p(mask_in), _p(mult), _p(y), _stream()), 'p3d_hconv2d_fwd')
        ctx.save_for_backward(x, mask_in, mult)
        ctx.cfg = (stride, pad, dil, tuple(w.shape))
        ctx.params = (w, bias)
        ctx.images = images
        ctx.joins = (join_put, join_take)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask_in, mult = ctx.saved_tensors
        stride, pad, dil, wshape = ctx.cfg
        w_param, b_param = ctx.params
        images = ctx.images
        dy = _cl(dy)
        n, c, h, wd = x.shape
        k, _, r, s = wshape
        d = _desc((n, c, h, wd), (k, c, r, s), stride, pad, dil)
        L, st = lib(), _stream()
        dx = dw = db = None
        if mult is not None:                                 # partial conv: dy * mult once, read by dgrad and wgrad (partial_conv.py:53)
            scaled = torch.empty_like(dy, memory_format=CL)
            check(L.p3d_hscale_pixels(_p(dy), _p(mult), _p(scaled), n * d.Ho * d.Wo, k, st), 'p3d_hscale_pixels')
            dy = scaled
        join_put, join_take = ctx.joins
        if join_take is not None:
            join_take.taken = True
        if ctx.needs_input_grad[0]:
            if images.crsk is None:
                raise P3DError('hconv2d: this layer was built without a dgrad weight image')
            joined = join_take.buf if join_take is not None else None
            if joined is not None and (joined.shape != x.shape or joined.dtype != torch.float16):
                raise P3DError('GradJoin: the shortcut gradient %s does not match the block input %s' % (tuple(joined.shape), tuple(x.shape)))
            if joined is not None:
                dx, join_take.buf = _cl(joined), None          # accumulate onto the shortcut's gradient (ops.GradJoin)
                d.accumulate = 1
            else:
                dx = _empty(n, c, h, wd, x.device)
            with ops._Timed('dgrad', d):
                check(L.p3d_hconv2d_dgrad(ctypes.byref(d), _p(dy), _p(images.crsk), _p(mask_in), _p(dx), st), 'p3d_hconv2d_dgrad')
            d.accumulate = 0
            if ops._stash(join_put, dx):
                dx = None
        if ctx.needs_input_grad[1]:
            sink = _grad_sink(w_param)
            dw = torch.empty(wshape, dtype=torch.float32, device=x.device) if sink is None else sink
            d.accumulate = 0 if sink is None else 1
            nbytes = L.p3d_hconv2d_wgrad_workspace_bytes(ctypes.byref(d))
            if ops.WGRAD_STREAM and sink is not None and not (ops.LAST_WGRAD_ON_LAUNCH and not ctx.needs_input_grad[0]):
                side = ops._side_stream(x.device)
                ops._queue_join()
                side.wait_stream(torch.cuda.current_stream())
                ws = ops._side_workspace(x.device, nbytes)
                with torch.cuda.stream(side):
                    with ops._Timed('wgrad', d):
                        check(L.p3d_hconv2d_wgrad(ctypes.byref(d), _p(dy), _p(x), _p(mask_in), _p(dw), images.c_real, 1.0, _p(ws), ws.numel(), _stream()),
                              'p3d_hconv2d_wgrad')
                for t in (dy, x, mask_in):
                    if t is not None:
                        t.record_stream(side)
            else:
                ws = workspace(x.device, nbytes)
                with ops._Timed('wgrad', d):
                    check(L.p3d_hconv2d_wgrad(ctypes.byref(d), _p(dy), _p(x), _p(mask_in), _p(dw), images.c_real, 1.0, _p(ws), ws.numel(), st), 'p3d_hconv2d_wgrad')
            d.accumulate = 0
            if sink is not None:
                dw = None
                _grad_done(w_param)
        if b_param is not None and ctx.needs_input_grad[2]:
            sink = _grad_sink(b_param)
            db = torch.empty(k, dtype=torch.float32, device=x.device) if sink is None else sink
            check(L.p3d_hconv2d_bgrad(_p(dy), n * d.Ho * d.Wo, k, _p(db), 1.0, 0 if sink is None else 1, st), 'p3d_hconv2d_bgrad')
            if sink is not None:
                db = None
                _grad_done(b_param)
        return dx, dw, db, None, None, None, None, None, None, None, None


def conv2d(x, module, stride, pad, dil, join_put=None, join_take=None, mask_in=None, mult=None):
    images = getattr(module, '_h_images', None)
    if images is None:
        raise P3DError('fp16 convolution without weight images: call ops_half.refresh_weights(model) first')
    return HConv2dFn.apply(x, module.weight, module.bias, images, stride, pad, dil, join_put, join_take, mask_in, mult)


class HBatchNormActFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, res, relu, training, momentum, eps, res_join=None):
        _need_half(x, res)
        ops._need_gpu(gamma, beta, running_mean, running_var)
        x = _cl(x)
        res = None if res is None else _cl(res)
        n, c, h, w = x.shape
        L, st = lib(), _stream()
        y = _empty(n, c, h, w, x.device)
        ws = workspace(x.device, L.p3d_hbn_workspace_bytes(c))
        if training:
            coef = torch.empty((c, 4), dtype=torch.float32, device=x.device)
            check(L.p3d_hbn_train_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y), _p(coef),
                                      n * h * w, c, momentum, eps, int(relu), _p(ws), ws.numel(), st), 'p3d_hbn_train_fwd')
            ctx.save_for_backward(x, y if (relu and res is not None) else None, coef)
        else:
            check(L.p3d_hbn_eval_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y), n * h * w, c, eps, int(relu),
                                     _p(ws), ws.numel(), st), 'p3d_hbn_eval_fwd')
            coef = None
            if any(ctx.needs_input_grad[:3]):                       # frozen statistics inside a training step (-do_freeze)
                coef = torch.empty((c, 4), dtype=torch.float32, device=x.device)
                check(L.p3d_hbn_eval_coef(_p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(coef), c, eps, st), 'p3d_hbn_eval_coef')
            ctx.save_for_backward(x, y if (relu and res is not None) else None, coef)
        ctx.cfg = (bool(relu), bool(training), res is not None)
        ctx.params = (gamma, beta)
        ctx.res_join = res_join
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, coef = ctx.saved_tensors
        relu, training, has_res = ctx.cfg
        dy = _cl(dy)
        n, c, h, w = x.shape
        L, st = lib(), _stream()
        dx = _empty(n, c, h, w, x.device)
        dres = None
        if has_res and ctx.needs_input_grad[5]:
            dres = _empty(n, c, h, w, x.device) if relu else dy
        g_param, b_param = ctx.params
        g_sink, b_sink = _grad_sink(g_param), _grad_sink(b_param)
        direct = g_sink is not None and b_sink is not None
        dgamma = g_sink if direct else torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = b_sink if direct else torch.empty_like(dgamma)
        ws = workspace(x.device, L.p3d_hbn_workspace_bytes(c))
        bwd = L.p3d_hbn_train_bwd if training else L.p3d_hbn_frozen_bwd
        check(bwd(_p(dy), _p(x), _p(y), _p(coef), _p(dx), _p(dres) if (dres is not None and relu) else None,
                  _p(dgamma), _p(dbeta), n * h * w, c, int(relu), int(direct), _p(ws), ws.numel(), st), 'p3d_hbn_train_bwd')
        if direct:
            dgamma = dbeta = None
            _grad_done(g_param)
            _grad_done(b_param)
        if dres is not None and relu and ops._stash(ctx.res_join, dres):     # (without ReLU dres aliases dy: never hand that out)
            dres = None
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, res=None, relu=False, training=True, momentum=0.1, eps=1e-5, res_join=None):
    return HBatchNormActFn.apply(x, gamma, beta, running_mean, running_var, res, relu, training, momentum, eps, res_join)


class HMaxPoolFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        _need_half(x)
        x = _cl(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = _empty(n, c, ho, wo, x.device)
        idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
        check(lib().p3d_hmaxpool3x3s2_fwd(_p(x), _p(y), _p(idx), n, h, w, c, _stream()), 'p3d_hmaxpool3x3s2_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dy = _cl(dy)
        dx = _empty(n, c, h, w, dy.device)
        check(lib().p3d_hmaxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), n, h, w, c, _stream()), 'p3d_hmaxpool3x3s2_bwd')
        return dx


def maxpool3x3s2(x):
    return HMaxPoolFn.apply(x)


class HConcatFn(torch.autograd.Function):
    """torch.cat((x, y), dim=1) of the Fusion block (fusionnet.py:138) on NHWC fp16 tensors."""

    @staticmethod
    def forward(ctx, x, y):
        _need_half(x, y)
        x, y = _cl(x), _cl(y)
        n, ca, h, w = x.shape
        cb = y.shape[1]
        if y.shape[0] != n or y.shape[2:] != x.shape[2:]:
            raise P3DError('hconcat: shapes %s and %s do not match' % (tuple(x.shape), tuple(y.shape)))
        out = _empty(n, ca + cb, h, w, x.device)
        check(lib().p3d_hconcat(_p(x), _p(y), _p(out), n * h * w, ca, cb, 0, _stream()), 'p3d_hconcat')
        ctx.dims = (n, ca, cb, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, ca, cb, h, w = ctx.dims
        dout = _cl(dout)
        dx, dy = _empty(n, ca, h, w, dout.device), _empty(n, cb, h, w, dout.device)
        check(lib().p3d_hconcat(_p(dx), _p(dy), _p(dout), n * h * w, ca, cb, 1, _stream()), 'p3d_hconcat')
        return dx, dy


def concat(x, y):
    return HConcatFn.apply(x, y)


class HReluFn(torch.autograd.Function):
    """standalone F.relu on fp16 tensors (depthnet.py:197-198 skip_relu variants)"""

    @staticmethod
    def forward(ctx, x):
        _need_half(x)
        x = _cl(x)
        y = torch.empty_like(x, memory_format=CL)
        check(lib().p3d_hrelu(_p(x), None, _p(y), x.numel(), _stream()), 'p3d_hrelu')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _cl(dy)
        dx = torch.empty_like(y, memory_format=CL)
        check(lib().p3d_hrelu(_p(y), _p(dy), _p(dx), y.numel(), _stream()), 'p3d_hrelu')
        return dx


def relu(x):
    return HReluFn.apply(x)


# ---- one autograd node per residual block on the fp16 kernels (p3d_hblock_fwd / p3d_hblock_bwd) -----------------------------------------------------------------------
# The fp16 step is bound by the host: ~330 Python-level calls per direction for ResNet-50, 15 ms of enqueue work for 16.7 ms of step.  The block executor makes
# one C call per block and direction over the same per-layer entry points (depthnet.py:40-56,96-116 under model.half(), depth_train.py:73-83).
import os as _os

HALF_BLOCKS = _os.environ.get('P3D_HALF_BLOCKS', '1') != '0'
HALF_MASK = _os.environ.get('P3D_HALF_MASK', '1') != '0'          # the closing ReLU's mask bytes instead of the block output in backward (A/B)
_vp = ctypes.c_void_p


class HBlockIO(ctypes.Structure):
    """struct p3d_hblock_io"""
    _fields_ = [('x', _vp), ('out', _vp), ('w_krsc', _vp * 4), ('w_crsk', _vp * 4), ('c', _vp * 4), ('a', _vp * 4), ('coef', _vp * 4), ('gamma', _vp * 4), ('beta', _vp * 4),
                ('running_mean', _vp * 4), ('running_var', _vp * 4), ('dout', _vp), ('dc', _vp * 4), ('da', _vp * 4), ('dx', _vp), ('dw', _vp * 4), ('dgamma', _vp * 4),
                ('dbeta', _vp * 4), ('c_real', ctypes.c_int32 * 4), ('out_mask', _vp)]


class _HPlan:
    def __init__(self, block, x_shape):
        from . import ops_block
        d = ops_block.BlockDesc()
        d.nconv = len(block._chain)
        d.has_downsample = int(block.downsample is not None)
        d.relu_out = 1
        shape = tuple(x_shape)
        self.shapes = {}
        self.layers = ops_block._layers(block)
        for slot, conv, bn in self.layers:
            src = tuple(x_shape) if slot in (0, 3) else shape
            k, _, r, s_ = conv.weight.shape
            cd = _desc(src, (k, src[1], r, s_), ops_block._one(conv.stride), ops_block._one(conv.padding), ops_block._one(conv.dilation))
            d.conv[slot] = cd
            d.eps[slot] = bn.eps
            d.momentum[slot] = 0.1 if bn.momentum is None else bn.momentum
            self.shapes[slot] = (cd.N, cd.K, cd.Ho, cd.Wo)
            if slot != 3:
                shape = self.shapes[slot]
        self.desc = d
        self.out_shape = shape
        self.ok = block.downsample is None or self.shapes[3] == shape
        m, sd = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().p3d_hblock_workspace_bytes(ctypes.byref(d), ctypes.byref(m), ctypes.byref(sd)), 'p3d_hblock_workspace_bytes')
        self.main_bytes, self.side_bytes = m.value, sd.value


def block_usable(block, x):
    """The fp16 block executor takes this call: dense block on fp16 NHWC tensors, every BatchNorm computing batch statistics, closing ReLU, weight images present."""
    if not HALF_BLOCKS or block.partial or block.skip_relu or x.dtype != torch.float16 or not x.is_cuda or x.dim() != 4:
        return False
    from . import ops_block
    for _, conv, bn in ops_block._layers(block):
        img = getattr(conv, '_h_images', None)
        if (not bn.training or not (bn.affine and bn.track_running_stats) or conv.bias is not None or type(conv).__name__ != 'Conv2d' or img is None or img.crsk is None
                or pad8(conv.out_channels) != conv.out_channels):
            return False
    cache = block.__dict__.setdefault('_hblk_plans', {})
    plan = cache.get(tuple(x.shape))
    if plan is None:
        plan = cache[tuple(x.shape)] = _HPlan(block, x.shape)
    return plan.ok and x.shape[1] == ops_block._layers(block)[0][1]._h_images.cpad


class HResidualBlockFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, block, *params):
        x = _cl(x)
        plan = block.__dict__['_hblk_plans'][tuple(x.shape)]
        dev = x.device
        io = HBlockIO()
        io.x = x.data_ptr()
        out = _empty(*plan.out_shape, dev)
        io.out = out.data_ptr()
        cs, acts, coefs = {}, {}, {}
        last = plan.desc.nconv - 1
        for slot, conv, bn in plan.layers:
            n, k, ho, wo = plan.shapes[slot]
            cs[slot] = _empty(n, k, ho, wo, dev)
            coefs[slot] = torch.empty((k, 4), dtype=torch.float32, device=dev)
            io.c[slot], io.coef[slot] = cs[slot].data_ptr(), coefs[slot].data_ptr()
            if slot != last:
                acts[slot] = _empty(n, k, ho, wo, dev)
                io.a[slot] = acts[slot].data_ptr()
            io.w_krsc[slot] = conv._h_images.krsc.data_ptr()
            io.gamma[slot], io.beta[slot] = bn.weight.data_ptr(), bn.bias.data_ptr()
            io.running_mean[slot], io.running_var[slot] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            if getattr(bn, '_ticked', False):
                bn._ticked = False
            else:
                bn.num_batches_tracked.add_(1)
        # the closing ReLU's mask as one byte per 8 outputs: backward reads it in place of `out` (p3d_hblock_fuse_sums(1); ignored otherwise)
        n, k, ho, wo = plan.out_shape
        out_mask = torch.empty(n * ho * wo * (k // 8), dtype=torch.uint8, device=dev)
        if HALF_MASK:
            io.out_mask = out_mask.data_ptr()
        ws = workspace(dev, plan.main_bytes)
        check(lib().p3d_hblock_fwd(ctypes.byref(plan.desc), ctypes.byref(io), _p(ws), ws.numel(), _stream()), 'p3d_hblock_fwd')
        ctx.block, ctx.plan = block, plan
        ctx.fused = HALF_MASK and bool(lib().p3d_hblock_fuse_sums(-1))
        ctx.saved = (cs, acts, coefs, out_mask)
        ctx.save_for_backward(x, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import ops_block
        if ctx.saved is None:
            raise P3DError('residual_block (fp16): backward called a second time on the same graph; run the forward again')
        block, plan = ctx.block, ctx.plan
        x, out = ctx.saved_tensors
        cs, acts, coefs, out_mask = ctx.saved
        ctx.saved = None
        dev = x.device
        dout = _cl(dout)
        d = plan.desc
        last = d.nconv - 1
        need_dx = bool(ctx.needs_input_grad[0])
        params = []
        for slot, conv, bn in plan.layers:
            params += [(slot, 'dw', conv.weight), (slot, 'dgamma', bn.weight), (slot, 'dbeta', bn.bias)]
        sinks = [_grad_sink(p) for _, _, p in params]
        direct = all(s_ is not None for s_ in sinks)
        grads = sinks if direct else [torch.empty(p.shape, dtype=torch.float32, device=dev) for _, _, p in params]
        io = HBlockIO()
        io.x, io.out, io.dout = x.data_ptr(), out.data_ptr(), dout.data_ptr()
        if ctx.fused:                                      # (forward wrote the mask bytes only with the sums from the epilogues)
            io.out_mask = out_mask.data_ptr()
        keep = []
        for slot, conv, bn in plan.layers:
            n, k, ho, wo = plan.shapes[slot]
            io.c[slot], io.coef[slot] = cs[slot].data_ptr(), coefs[slot].data_ptr()
            if slot in acts:
                io.a[slot] = acts[slot].data_ptr()
            io.w_crsk[slot] = conv._h_images.crsk.data_ptr()
            io.c_real[slot] = conv._h_images.c_real
            dc = _empty(n, k, ho, wo, dev)
            keep.append(dc)
            io.dc[slot] = dc.data_ptr()
            if slot < last:
                da = _empty(n, k, ho, wo, dev)
                keep.append(da)
                io.da[slot] = da.data_ptr()
        dres = _empty(*plan.out_shape, dev)                # the gradient that enters the shortcut; with an identity shortcut it becomes dx
        io.da[3] = dres.data_ptr()
        dx = None
        if need_dx:
            if d.has_downsample:
                dx = _empty(*x.shape, dev)
                io.dx = dx.data_ptr()
            else:
                dx = dres
        for (slot, kind, _), g in zip(params, grads):
            getattr(io, kind)[slot] = g.data_ptr()
        desc = ops_block.BlockDesc.from_buffer_copy(d)
        desc.need_dx, desc.accumulate_grads = int(need_dx), int(direct)
        ws = workspace(dev, plan.main_bytes)
        two = ops.WGRAD_STREAM and direct
        if two:
            side = ops._side_stream(dev)
            ops._queue_join()
            sws = ops._side_workspace(dev, plan.side_bytes)
            side_handle = _vp(side.cuda_stream)
        else:
            sws = ops._second_workspace(dev, plan.side_bytes)
            side_handle = None
        check(lib().p3d_hblock_bwd(ctypes.byref(desc), ctypes.byref(io), _p(ws), ws.numel(), _p(sws), sws.numel(), _stream(), side_handle), 'p3d_hblock_bwd')
        if two:
            for t in [x, dout] + list(acts.values()) + keep:        # freed by autograd while the second stream may still read them
                t.record_stream(side)
        if direct:
            for _, _, p in params:
                _grad_done(p)
            return (dx, None) + (None,) * len(params)
        return (dx, None) + tuple(grads)


def residual_block(block, x):
    from . import ops_block
    params = []
    for _, conv, bn in ops_block._layers(block):
        params += [conv.weight, bn.weight, bn.bias]
    return HResidualBlockFn.apply(x, block, *params)
