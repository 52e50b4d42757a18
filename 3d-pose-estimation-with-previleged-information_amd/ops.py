"""torch.autograd.Function wrappers over the C ABI (include/p3d_hip.h).

Each Function replaces the torch operator the reference calls at the cited line; arithmetic
happens only in the hand-written HIP kernels.  Tensors must be fp32 on a HIP device -- there is
no CPU or eager-PyTorch fallback: anything else raises P3DError.
"""
import ctypes
import os

import torch

from ._lib import ConvDesc, P3DError, check, lib

CRITERIA = {'SmoothL1': 0, 'L1': 1, 'MSE': 2}


def _need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise P3DError('p3d ops run only on a HIP device (got a %s tensor); there is no CPU fallback' % t.device)
        if t.dtype != torch.float32:
            raise P3DError('p3d ops are fp32 only (got %s)' % t.dtype)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    """Raw handle of the current HIP stream of the current device (the C fast path: this runs once per kernel launch)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


_workspaces = {}

# When a list, every conv launch is bracketed by a pair of events on the launch stream and
# (kind, algorithmic_flops, start, end) is appended: bench.py's live per-kernel timing.
PROFILE = None


class _Timed:
    def __init__(self, kind, d):
        self.kind = kind
        self.flops = 2.0 * d.N * d.K * d.Ho * d.Wo * d.C * d.R * d.S

    def __enter__(self):
        if PROFILE is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.end.record()
            PROFILE.append((self.kind, self.flops, self.start, self.end))
        return False



def workspace(device, nbytes):
    """Grow-only scratch buffer per device; ops on one stream are ordered, so it can be shared."""
    ws = _workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


# Weight-gradient kernels leave the critical path of backward (dy -> BN backward -> dgrad -> next layer): they only feed the
# optimizer.  With P3D_WGRAD_STREAM != 0 they are launched on a second HIP stream, so the matrix-core-bound wgrad of layer i
# overlaps the HBM-bound BN backward and the grid tails of layer i-1's kernels instead of queueing behind them.
# join_side_stream() orders the launch stream after everything issued there (before an all-reduce / the optimizer reads .grad).
WGRAD_STREAM = os.environ.get('P3D_WGRAD_STREAM', '1') != '0'
# -half_acc: the weight gradient of a network's FIRST layer (no data gradient behind it) is the last kernel of the backward pass: queued on the second stream it waits
# behind the weight gradients still pending there while the launch stream has nothing left to do; on the launch stream it runs beside them (tools/r04/r4_t.sh: 14.04 / 14.03 ms
# against 14.09 / 14.11; P3D_LAST_WGRAD_MAIN=0: A/B).  The fp32 step measured no difference (28.17 / 28.11 / 28.10 against 28.29 / 28.05 / 27.99) and keeps the second stream.
LAST_WGRAD_ON_LAUNCH = os.environ.get('P3D_LAST_WGRAD_MAIN', '1') != '0'
_side_streams = {}
_side_workspaces = {}


_second_workspaces = {}


def _second_workspace(device, nbytes):
    """A second grow-only scratch buffer on the launch stream (the block executor's weight-gradient slabs when no second stream is used)."""
    ws = _second_workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _second_workspaces[device] = ws
    return ws


SIDE_STREAM_KIND = os.environ.get('P3D_SIDE_STREAM', 'probe')       # probe | torch (a stream from PyTorch's pool) | low | normal | high (a fixed priority class)
SIDE_STREAM_OVERLAPS = {}                                              # device -> what the probe found (None: not probed)


def cu_mask_words(spec, total=256):
    """'N' -> the first N of `total` compute units, 'N:hi' -> the last N, '0x...' -> the mask itself, as little-endian 32-bit words.  (KFD deals the bits of a
    queue's mask round-robin over the eight XCDs: a contiguous run of N bits is N / 8 CUs on each.)"""
    spec = str(spec)
    if spec.lower().startswith('0x'):
        value = int(spec, 16)
    else:
        n, _, where = spec.partition(':')
        n = max(1, min(int(n), total))
        value = ((1 << n) - 1) << (total - n if where == 'hi' else 0)
    return [(value >> (32 * i)) & 0xFFFFFFFF for i in range(total // 32)]


def masked_stream(device, spec):
    """A HIP stream whose kernels run only on the compute units `spec` names (cu_mask_words), as a torch ExternalStream."""
    words = cu_mask_words(spec)
    arr = (ctypes.c_uint32 * len(words))(*words)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        check(lib().p3d_stream_create_cumask(arr, len(words), ctypes.byref(handle)), 'stream_create_cumask')
    return torch.cuda.ExternalStream(handle.value, device=device)


SIDE_STREAM_CUS = os.environ.get('P3D_SIDE_CUS', '')                  # e.g. '192': the weight-gradient stream may use 192 of the 256 compute units


def _side_stream(device):
    """The weight-gradient stream.  HIP multiplexes the streams of one priority onto four hardware queues: a stream taken from PyTorch's pool landed on
    the launch stream's own queue whenever an RCCL communicator had been created first (rocprofv3: every kernel of the step on one queue), which costs
    the overlap plus a queue barrier per cross-stream event, +2.3 ms per step.  So the stream is chosen by measurement:
    p3d_stream_create_beside creates candidates until two spin kernels, one per stream, demonstrably run side by side."""
    st = _side_streams.get(device)
    if st is None:
        if SIDE_STREAM_CUS:
            st = masked_stream(device, SIDE_STREAM_CUS)
            SIDE_STREAM_OVERLAPS[device] = None
        elif SIDE_STREAM_KIND == 'torch':
            st = torch.cuda.Stream(device=device)
            SIDE_STREAM_OVERLAPS[device] = None
        else:
            handle, found = ctypes.c_void_p(), ctypes.c_int32(-1)
            with torch.cuda.device(device):
                if SIDE_STREAM_KIND == 'probe':
                    check(lib().p3d_stream_create_beside(ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream), ctypes.byref(handle), ctypes.byref(found)), 'stream_create_beside')
                else:
                    check(lib().p3d_stream_create({'low': 1, 'normal': 0, 'high': -1}[SIDE_STREAM_KIND], ctypes.byref(handle)), 'stream_create')
            st = torch.cuda.ExternalStream(handle.value, device=device)
            SIDE_STREAM_OVERLAPS[device] = None if found.value < 0 else bool(found.value)
        _side_streams[device] = st
    return st


def _side_workspace(device, nbytes):
    ws = _side_workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        with torch.cuda.stream(_side_stream(device)):
            ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _side_workspaces[device] = ws
    return ws


_join_queued = False


_event_ring, _event_next = [], [0]


def _mark_ready():
    """Record 'everything launched so far on the current stream' into a recycled event (creating an event per call costs ~20 us of host
    time; a recorded event may be re-recorded once the wait on it has been enqueued, which happens right after in the same call)."""
    if len(_event_ring) < 64:
        _event_ring.append(torch.cuda.Event())
    ev = _event_ring[_event_next[0] % len(_event_ring)]
    _event_next[0] += 1
    ev.record()
    return ev


def join_side_stream(device=None):
    global _join_queued
    _join_queued = False
    for dev, st in _side_streams.items():
        if device is None or dev == device:
            torch.cuda.current_stream(dev).wait_stream(st)


def _queue_join():
    """First side-stream launch of a backward pass: have the autograd engine join the streams when the pass ends, so whoever
    called .backward() sees complete gradients on the launch stream (like any other autograd result)."""
    global _join_queued
    if not _join_queued:
        _join_queued = True
        torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)


def _grad_sink(param):
    """The preallocated .grad of a parameter owned by FlatAdam (flagged _p3d_direct_grad), or None.
    Backward kernels accumulate straight into it (the buffer is zeroed once per step), so autograd's own
    per-parameter add kernel and its temporary are skipped; the Function then returns None for that input."""
    if param is not None and getattr(param, '_p3d_direct_grad', False) and param.grad is not None:
        return param.grad
    return None


def _grad_done(param):
    ready = getattr(param, '_p3d_grad_ready', None)       # set by dist.GradReducer: counts the bucket down
    if ready is not None:
        ready()


def conv_out(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _desc(x_shape, w_shape, stride, pad, dil, c_offset=0, c_total=None, accumulate=0):
    n, c, h, w = x_shape
    k, cw, r, s = w_shape
    d = ConvDesc()
    d.N, d.C, d.H, d.W = n, c, h, w
    d.K, d.R, d.S = k, r, s
    d.stride, d.pad, d.dil = stride, pad, dil
    d.Ho, d.Wo = conv_out(h, r, stride, pad, dil), conv_out(w, s, stride, pad, dil)
    d.c_total = cw if c_total is None else c_total
    d.c_offset = c_offset
    d.accumulate = accumulate
    return d


# --------------------------------------------------------------------------------------------
class GradJoin:
    """Where a tensor fans out inside a residual block (block input -> first conv + shortcut), autograd would sum the two
    gradients with an extra elementwise pass (read 2, write 1 over the block input).  A GradJoin hands the gradient that is
    produced FIRST (the shortcut's: residual gradient of the closing BN, or the dgrad of the downsample conv) to the op that runs
    LAST (the first conv of the main path, created first in forward, hence last in backward), whose dgrad kernel then accumulates
    into that buffer (`accumulate = 1`).  The producer returns None to autograd, the consumer returns the joined buffer.
    Order independence: the consumer marks the join `taken` when its backward runs; a consumer that finds the slot empty behaves
    normally, and a producer that runs AFTER the consumer (or whose consumer needs no input gradient) sees `taken` and returns its
    gradient to autograd like any other Function, so the shortcut gradient is never dropped.  A gradient that was stashed but never
    picked up (the consumer's backward did not run at all) is an error: the first stash of a pass queues an engine callback that
    raises P3DError when the pass ends (`check_joins`), whoever called .backward()."""
    __slots__ = ('buf', 'taken', '__weakref__')
    _live = None

    def __init__(self):
        self.buf = None
        self.taken = False
        if GradJoin._live is None:
            import weakref
            GradJoin._live = weakref.WeakSet()
        GradJoin._live.add(self)


def pending_joins():
    return 0 if GradJoin._live is None else sum(1 for j in GradJoin._live if j.buf is not None)


_join_check_queued = False


def check_joins():
    """Raise if a shortcut gradient was stashed in a GradJoin and never consumed (it would silently be missing from the input gradient)."""
    global _join_check_queued
    _join_check_queued = False
    n = pending_joins()
    if n:
        for j in GradJoin._live:
            j.buf = None
        raise P3DError('%d shortcut gradient(s) were produced but never joined (ops.GradJoin): the backward pass did not reach the first '
                       'convolution of a residual block' % n)


def _stash(join, grad):
    """Producer side of a GradJoin: True if `grad` was handed to the join (return None to autograd), False if the consumer has already run."""
    global _join_check_queued
    if join is None or join.taken:
        return False
    join.buf = grad
    if not _join_check_queued:
        _join_check_queued = True
        torch.autograd.Variable._execution_engine.queue_callback(check_joins)
    return True


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (depthnet.py:16-33,65-89,138,156) and, with mask_in/mult, PartialConv's masked,
    renormalised convolution (partial_conv.py:45-53).  join_put / join_take: see GradJoin."""

    @staticmethod
    def forward(ctx, x, w, bias, mask_in, mult, stride, pad, dil, join_put=None, join_take=None):
        _need_gpu(x, w, bias, mask_in, mult)
        x, w = x.contiguous(), w.contiguous()
        d = _desc(x.shape, w.shape, stride, pad, dil)
        if x.shape[1] != w.shape[1]:
            raise P3DError('conv2d: input has %d channels, weight expects %d' % (x.shape[1], w.shape[1]))
        y = torch.empty((d.N, d.K, d.Ho, d.Wo), dtype=torch.float32, device=x.device)
        L = lib()
        nbytes = L.p3d_conv2d_fwd_workspace_bytes(ctypes.byref(d))
        ws = workspace(x.device, nbytes) if nbytes else None
        with _Timed('fwd', d):
            check(L.p3d_conv2d_fwd(ctypes.byref(d), _p(x), _p(w), _p(bias), _p(mask_in), _p(mult), _p(y), _p(ws), ws.numel() if ws is not None else 0,
                                   _stream()), 'p3d_conv2d_fwd')
        ctx.save_for_backward(x, w, mask_in, mult)
        ctx.cfg = (stride, pad, dil, bias is not None)
        ctx.params = (w, bias)
        ctx.joins = (join_put, join_take)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mask_in, mult = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.cfg
        join_put, join_take = ctx.joins
        dy = dy.contiguous()
        d = _desc(x.shape, w.shape, stride, pad, dil)
        L = lib()
        st = _stream()
        dx = dw = db = None
        # dy is complete at this point of the launch stream: the wgrad stream waits for THIS event, not for the dgrad launched below
        dy_ready = _mark_ready() if (WGRAD_STREAM and ctx.needs_input_grad[1]) else None
        if join_take is not None:
            join_take.taken = True                            # producers that run later return their gradient to autograd themselves
        if ctx.needs_input_grad[0]:
            joined = join_take.buf if join_take is not None else None
            if joined is not None:
                if joined.shape != x.shape or joined.dtype != x.dtype:
                    raise P3DError('GradJoin: the shortcut gradient %s does not match the block input %s' % (tuple(joined.shape), tuple(x.shape)))
                dx, join_take.buf = joined, None              # accumulate onto the shortcut's gradient: no separate add pass
                d.accumulate = 1
            else:
                dx = torch.empty_like(x)
            ws = workspace(x.device, L.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d)))
            with _Timed('dgrad', d):
                check(L.p3d_conv2d_dgrad(ctypes.byref(d), _p(dy), _p(w), _p(mult), _p(mask_in), _p(dx), _p(ws), ws.numel(), st), 'p3d_conv2d_dgrad')
            d.accumulate = 0
            if _stash(join_put, dx):
                dx = None                                     # the consumer returns it
        w_param, b_param = ctx.params
        if ctx.needs_input_grad[1]:
            sink = _grad_sink(w_param)
            dw = torch.empty_like(w) if sink is None else sink
            d.accumulate = 0 if sink is None else 1
            nbytes = L.p3d_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
            if WGRAD_STREAM and sink is not None:
                side = _side_stream(x.device)
                _queue_join()
                side.wait_event(dy_ready)                               # dy (and the zeroed gradient buffer) are ready
                ws = _side_workspace(x.device, nbytes)
                with torch.cuda.stream(side):
                    with _Timed('wgrad', d):
                        check(L.p3d_conv2d_wgrad(ctypes.byref(d), _p(dy), _p(x), _p(mult), _p(mask_in), _p(dw), _p(ws), ws.numel(), _stream()),
                              'p3d_conv2d_wgrad')
                for t in (dy, x, mult, mask_in):                        # freed by autograd while the side stream may still read them
                    if t is not None:
                        t.record_stream(side)
            else:
                ws = workspace(x.device, nbytes)
                with _Timed('wgrad', d):
                    check(L.p3d_conv2d_wgrad(ctypes.byref(d), _p(dy), _p(x), _p(mult), _p(mask_in), _p(dw), _p(ws), ws.numel(), st), 'p3d_conv2d_wgrad')
            if sink is not None:
                dw = None
                _grad_done(w_param)
        if has_bias and ctx.needs_input_grad[2]:
            sink = _grad_sink(b_param)
            db = torch.empty(d.K, dtype=torch.float32, device=x.device) if sink is None else sink
            if mult is not None:      # PartialConv with bias: d out / d b = mask_out (partial_conv.py:48-51)
                check(L.p3d_conv2d_bgrad_masked(_p(dy), _p(mult), d.N, d.K, d.Ho * d.Wo, _p(db), 0 if sink is None else 1, st), 'p3d_conv2d_bgrad_masked')
            else:
                check(L.p3d_conv2d_bgrad(_p(dy), d.N, d.K, d.Ho * d.Wo, _p(db), 0 if sink is None else 1, st), 'p3d_conv2d_bgrad')
            if sink is not None:
                db = None
                _grad_done(b_param)
        return dx, dw, db, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, dil=1, mask_in=None, mult=None, join_put=None, join_take=None):
    return Conv2dFn.apply(x, w, bias, mask_in, mult, stride, pad, dil, join_put, join_take)


class ConvCat1x1Fn(torch.autograd.Function):
    """conv(cat([x, y], 1), w) without materialising the concat (fusionnet.py:138-139):
    two passes over disjoint input-channel windows of w, the second accumulating."""

    @staticmethod
    def forward(ctx, x, y, w):
        _need_gpu(x, y, w)
        x, y, w = x.contiguous(), y.contiguous(), w.contiguous()
        c1, c2 = x.shape[1], y.shape[1]
        if w.shape[1] != c1 + c2 or x.shape[0] != y.shape[0] or x.shape[2:] != y.shape[2:]:
            raise P3DError('conv_cat: shapes %s %s do not match weight %s' % (tuple(x.shape), tuple(y.shape), tuple(w.shape)))
        L, st = lib(), _stream()
        d1 = _desc(x.shape, w.shape, 1, 0, 1, c_offset=0, c_total=c1 + c2)
        d2 = _desc(y.shape, w.shape, 1, 0, 1, c_offset=c1, c_total=c1 + c2, accumulate=1)
        out = torch.empty((d1.N, d1.K, d1.Ho, d1.Wo), dtype=torch.float32, device=x.device)
        # (each half runs on the x3 kernels: its weight image is built per call from the half's input-channel window of w)
        ws = workspace(x.device, max(L.p3d_conv2d_fwd_workspace_bytes(ctypes.byref(d1)), L.p3d_conv2d_fwd_workspace_bytes(ctypes.byref(d2)), 16))
        check(L.p3d_conv2d_fwd(ctypes.byref(d1), _p(x), _p(w), None, None, None, _p(out), _p(ws), ws.numel(), st), 'p3d_conv2d_fwd')
        check(L.p3d_conv2d_fwd(ctypes.byref(d2), _p(y), _p(w), None, None, None, _p(out), _p(ws), ws.numel(), st), 'p3d_conv2d_fwd')
        ctx.save_for_backward(x, y, w)
        ctx.w_param = w
        return out

    @staticmethod
    def backward(ctx, dy):
        x, y, w = ctx.saved_tensors
        dy = dy.contiguous()
        c1, c2 = x.shape[1], y.shape[1]
        L, st = lib(), _stream()
        d1 = _desc(x.shape, w.shape, 1, 0, 1, c_offset=0, c_total=c1 + c2)
        d2 = _desc(y.shape, w.shape, 1, 0, 1, c_offset=c1, c_total=c1 + c2)
        dx = dyy = dw = None
        wsd = workspace(x.device, max(L.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d1)), L.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d2)), 16))
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(L.p3d_conv2d_dgrad(ctypes.byref(d1), _p(dy), _p(w), None, None, _p(dx), _p(wsd), wsd.numel(), st), 'p3d_conv2d_dgrad')
        if ctx.needs_input_grad[1]:
            dyy = torch.empty_like(y)
            check(L.p3d_conv2d_dgrad(ctypes.byref(d2), _p(dy), _p(w), None, None, _p(dyy), _p(wsd), wsd.numel(), st), 'p3d_conv2d_dgrad')
        if ctx.needs_input_grad[2]:
            sink = _grad_sink(ctx.w_param)
            dw = torch.empty_like(w) if sink is None else sink
            for d, inp in ((d1, x), (d2, y)):
                d.accumulate = 0 if sink is None else 1
                ws = workspace(x.device, L.p3d_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
                check(L.p3d_conv2d_wgrad(ctypes.byref(d), _p(dy), _p(inp), None, None, _p(dw), _p(ws), ws.numel(), st), 'p3d_conv2d_wgrad')
            if sink is not None:
                dw = None
                _grad_done(ctx.w_param)
        return dx, dyy, dw


def conv_cat1x1(x, y, w):
    return ConvCat1x1Fn.apply(x, y, w)


def mask_count(mask, kernel, stride, pad, dil):
    """partial_conv.py:35-43 (no_grad): mask [N,1,H,W] -> (mult, mask_out) [N,1,Ho,Wo]."""
    _need_gpu(mask)
    mask = mask.contiguous()
    n, one, h, w = mask.shape
    d = _desc((n, 1, h, w), (1, 1, kernel, kernel), stride, pad, dil)
    mult = torch.empty((n, 1, d.Ho, d.Wo), dtype=torch.float32, device=mask.device)
    mask_out = torch.empty_like(mult)
    check(lib().p3d_mask_count_fwd(ctypes.byref(d), _p(mask), _p(mult), _p(mask_out), _stream()), 'p3d_mask_count_fwd')
    return mult, mask_out


def nonzero_mask(x):
    """(x != 0).float()  (partial_depthnet.py:215)"""
    _need_gpu(x)
    x = x.contiguous()
    m = torch.empty_like(x)
    check(lib().p3d_nonzero_mask(_p(x), _p(m), x.numel(), _stream()), 'p3d_nonzero_mask')
    return m


# --------------------------------------------------------------------------------------------
class BatchNormActFn(torch.autograd.Function):
    """act(bn(x) + res): nn.BatchNorm2d + optional residual add + optional F.relu
    (depthnet.py:42-56,98-116).  running_mean/running_var are updated in place in training."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, res, relu, training, momentum, eps, res_join=None):
        _need_gpu(x, gamma, beta, running_mean, running_var, res)
        x = x.contiguous()
        res = None if res is None else res.contiguous()
        n, c, h, w = x.shape
        L, st = lib(), _stream()
        y = torch.empty_like(x)
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            invstd = torch.empty_like(mean)
            ws = workspace(x.device, L.p3d_bn_workspace_bytes(n, c, h * w))
            check(L.p3d_bn_train_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y), _p(mean), _p(invstd),
                                     n, c, h * w, momentum, eps, int(relu), _p(ws), ws.numel(), st), 'p3d_bn_train_fwd')
            # the ReLU mask of a residual-free layer is recomputed from x in backward: y is not read (nor kept alive) for it
            ctx.save_for_backward(x, y if (relu and res is not None) else None, gamma, mean, invstd, beta)
        else:
            check(L.p3d_bn_eval_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y),
                                    n, c, h * w, eps, int(relu), st), 'p3d_bn_eval_fwd')
            ctx.save_for_backward(x, y if relu else None, gamma, running_mean, running_var, beta)
        ctx.cfg = (bool(relu), bool(training), eps, res is not None)
        ctx.params = (gamma, beta)
        ctx.res_join = res_join
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, s1, s2, beta = ctx.saved_tensors
        relu, training, eps, has_res = ctx.cfg
        dy = dy.contiguous()
        n, c, h, w = x.shape
        L, st = lib(), _stream()
        dx = torch.empty_like(x)
        dres = None
        if has_res and ctx.needs_input_grad[5]:
            dres = torch.empty_like(x) if relu else dy      # without ReLU the residual gradient is dy itself
        g_param, b_param = ctx.params
        g_sink, b_sink = _grad_sink(g_param), _grad_sink(b_param)
        direct = g_sink is not None and b_sink is not None
        dgamma = g_sink if direct else torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = b_sink if direct else torch.empty_like(dgamma)
        ws = workspace(x.device, L.p3d_bn_workspace_bytes(n, c, h * w))
        dres_ptr = _p(dres) if (dres is not None and relu) else None
        if training:
            check(L.p3d_bn_train_bwd(_p(dy), _p(x), _p(y), _p(gamma), _p(beta), _p(s1), _p(s2), _p(dx), dres_ptr, _p(dgamma), _p(dbeta),
                                     n, c, h * w, int(relu), int(direct), _p(ws), ws.numel(), st), 'p3d_bn_train_bwd')
        else:
            check(L.p3d_bn_eval_bwd(_p(dy), _p(x), _p(y), _p(gamma), _p(s1), _p(s2), _p(dx), dres_ptr, _p(dgamma), _p(dbeta),
                                    n, c, h * w, eps, int(relu), int(direct), _p(ws), ws.numel(), st), 'p3d_bn_eval_bwd')
        if direct:
            dgamma = dbeta = None
            _grad_done(g_param)
            _grad_done(b_param)
        if dres is not None and relu and _stash(ctx.res_join, dres):    # (without ReLU dres aliases dy: never hand that out)
            dres = None
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, res=None, relu=False, training=True, momentum=0.1, eps=1e-5, res_join=None):
    return BatchNormActFn.apply(x, gamma, beta, running_mean, running_var, res, relu, training, momentum, eps, res_join)


class ReluFn(torch.autograd.Function):
    """standalone F.relu (depthnet.py:197-198 skip_relu variants)"""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib().p3d_relu_fwd(_p(x), _p(y), x.numel(), _stream()), 'p3d_relu_fwd')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib().p3d_relu_bwd(_p(dy), _p(y), _p(dx), y.numel(), _stream()), 'p3d_relu_bwd')
        return dx


def relu(x):
    if x.dtype == torch.float16:
        from . import ops_half
        return ops_half.relu(x)
    return ReluFn.apply(x)


# --------------------------------------------------------------------------------------------
class MaxPool3x3S2Fn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (depthnet.py:140,192)"""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = x.contiguous()
        n, c, h, w = x.shape
        ho, wo = conv_out(h, 3, 2, 1, 1), conv_out(w, 3, 2, 1, 1)
        y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
        need_idx = ctx.needs_input_grad[0]
        idx = torch.empty((n, c, ho, wo), dtype=torch.uint8, device=x.device) if need_idx else None
        check(lib().p3d_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), n * c, h, w, _stream()), 'p3d_maxpool3x3s2_fwd')
        ctx.save_for_backward(idx)
        ctx.in_shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
        check(lib().p3d_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), n * c, h, w, _stream()), 'p3d_maxpool3x3s2_bwd')
        return dx


def maxpool3x3s2(x):
    return MaxPool3x3S2Fn.apply(x)


class StemTailFn(torch.autograd.Function):
    """maxpool(relu(bn(x))) of the stem in training mode as ONE node (p3d_stem_tail_fwd / bwd): the BatchNorm output (268 MB at batch 64) is neither written nor
    read; bit-identical to BatchNormActFn + MaxPool3x3S2Fn (depthnet.py:139-140)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps):
        _need_gpu(x, gamma, beta, running_mean, running_var)
        x = x.contiguous()
        n, c, h, w = x.shape
        L, st = lib(), _stream()
        y = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=x.device)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        ws = workspace(x.device, L.p3d_bn_workspace_bytes(n, c, h * w))
        check(L.p3d_stem_tail_fwd(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y), _p(idx), _p(mean), _p(invstd), n, c, h, w, momentum, eps,
                                  _p(ws), ws.numel(), st), 'p3d_stem_tail_fwd')
        ctx.save_for_backward(x, idx, gamma, beta, mean, invstd)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, idx, gamma, beta, mean, invstd = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = dy.contiguous()
        L, st = lib(), _stream()
        dx = torch.empty_like(x)
        g_param, b_param = ctx.params
        g_sink, b_sink = _grad_sink(g_param), _grad_sink(b_param)
        direct = g_sink is not None and b_sink is not None
        dgamma = g_sink if direct else torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = b_sink if direct else torch.empty_like(dgamma)
        ws = workspace(x.device, L.p3d_bn_workspace_bytes(n, c, h * w))
        check(L.p3d_stem_tail_bwd(_p(dy), _p(idx), _p(x), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(dx), _p(dgamma), _p(dbeta), n, c, h, w, int(direct),
                                  _p(ws), ws.numel(), st), 'p3d_stem_tail_bwd')
        if direct:
            dgamma = dbeta = None
            _grad_done(g_param)
            _grad_done(b_param)
        return dx, dgamma, dbeta, None, None, None, None


STEM_TAIL = os.environ.get('P3D_STEM_TAIL', '1') != '0'        # A/B switch: 0 = BatchNorm + ReLU and the max pool of the stem as two nodes


def stem_tail_usable(x, bn, pool):
    if not (STEM_TAIL and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bn.training and bn.affine and bn.track_running_stats and torch.is_grad_enabled()):
        return False
    one = lambda v: v[0] if isinstance(v, (tuple, list)) else v
    if (one(pool.kernel_size), one(pool.stride), one(pool.padding), one(pool.dilation)) != (3, 2, 1, 1) or pool.ceil_mode:
        return False
    n, c, h, w = x.shape
    return bool(lib().p3d_stem_tail_supported(n, c, h, w))


def stem_tail(x, bn):
    """maxpool(relu(bn(x))) with bn's running statistics and batch counter updated as nn.BatchNorm2d.forward does (nn.py)."""
    if getattr(bn, '_ticked', False):
        bn._ticked = False
    else:
        bn.num_batches_tracked.add_(1)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return StemTailFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps)


# --------------------------------------------------------------------------------------------
class SoftArgmax3dFn(torch.autograd.Function):
    """utils.to_heatmap followed by utils.decode (utils.py:154-194), fused."""

    @staticmethod
    def forward(ctx, z, depth, num_joints, height, width, depth_range):
        _need_gpu(z)
        z = z.contiguous()
        b = z.shape[0]
        if tuple(z.shape[1:]) != (depth * num_joints, height, width):
            raise P3DError('softargmax3d: z %s is not [B, %d*%d, %d, %d]' % (tuple(z.shape), depth, num_joints, height, width))
        coords = torch.empty((b, num_joints, 3), dtype=torch.float32, device=z.device)
        check(lib().p3d_softargmax3d_fwd(_p(z), _p(coords), b, depth, num_joints, height, width, depth_range, _stream()), 'p3d_softargmax3d_fwd')
        ctx.save_for_backward(z)
        ctx.cfg = (depth, num_joints, height, width, depth_range)
        return coords

    @staticmethod
    def backward(ctx, dcoords):
        (z,) = ctx.saved_tensors
        depth, num_joints, height, width, depth_range = ctx.cfg
        dcoords = dcoords.contiguous()
        dz = torch.empty_like(z)
        check(lib().p3d_softargmax3d_bwd(_p(dcoords), _p(z), _p(dz), z.shape[0], depth, num_joints, height, width, depth_range, _stream()),
              'p3d_softargmax3d_bwd')
        return dz, None, None, None, None, None


def softargmax3d(z, depth, num_joints, height, width, depth_range):
    return SoftArgmax3dFn.apply(z, depth, num_joints, height, width, float(depth_range))


class PoseLossFn(torch.autograd.Function):
    """Loss block of Trainer.vanilla_train (depth_train.py:397-405): returns (loss, spec_cam)."""

    @staticmethod
    def forward(ctx, relat, true_cam, true_val, key_index, loss_div, criterion, count_override):
        _need_gpu(relat, true_cam)
        relat, true_cam = relat.contiguous(), true_cam.contiguous()
        if not true_val.is_cuda:
            raise P3DError('pose_loss: true_val must be on the HIP device')
        val = true_val.contiguous().view(torch.uint8) if true_val.dtype == torch.bool else true_val.contiguous()
        if val.dtype != torch.uint8:
            raise P3DError('pose_loss: true_val must be bool or uint8')
        b, j, _ = relat.shape
        loss = torch.empty(1, dtype=torch.float32, device=relat.device)
        spec = torch.empty_like(relat)
        drelat = torch.empty_like(relat)
        check(lib().p3d_pose_loss_fwd_bwd(_p(relat), _p(true_cam), _p(val), _p(loss), _p(spec), _p(drelat), b, j, key_index,
                                          loss_div, CRITERIA[criterion], 1.0, _p(count_override), _stream()), 'p3d_pose_loss_fwd_bwd')
        ctx.save_for_backward(drelat)
        ctx.mark_non_differentiable(spec)
        return loss.view(()), spec

    @staticmethod
    def backward(ctx, dloss, dspec):
        (drelat,) = ctx.saved_tensors
        return drelat * dloss, None, None, None, None, None, None


def pose_loss(relat, true_cam, true_val, key_index, loss_div, criterion='SmoothL1', count_override=None):
    """count_override: optional 1-element fp32 device tensor replacing 3*n_valid as the mean's divisor."""
    return PoseLossFn.apply(relat, true_cam, true_val, int(key_index), float(loss_div), criterion, count_override)


class MaskedLossFn(torch.autograd.Function):
    """criterion(pred.view(-1, C)[valid.view(-1)], target.view(-1, C)[valid.view(-1)]) with mean reduction (train.py:94,112)."""

    @staticmethod
    def forward(ctx, pred, target, valid, criterion, count_override):
        _need_gpu(pred, target)
        if not valid.is_cuda:
            raise P3DError('masked_loss: valid must be on the HIP device')
        pred, target = pred.contiguous(), target.contiguous()
        val = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.contiguous()
        if val.dtype != torch.uint8 or pred.shape != target.shape or val.numel() * pred.shape[-1] != pred.numel():
            raise P3DError('masked_loss: pred %s / target %s / valid %s do not match' % (tuple(pred.shape), tuple(target.shape), tuple(valid.shape)))
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        check(lib().p3d_masked_loss_fwd_bwd(_p(pred), _p(target), _p(val), _p(loss), _p(dpred), val.numel(), pred.shape[-1], CRITERIA[criterion],
                                            _p(count_override), _stream()), 'p3d_masked_loss_fwd_bwd')
        ctx.save_for_backward(dpred)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (dpred,) = ctx.saved_tensors
        return dpred * dloss, None, None, None, None


def masked_loss(pred, target, valid, criterion='SmoothL1', count_override=None):
    return MaskedLossFn.apply(pred, target, valid, criterion, count_override)


class ReconCamFn(torch.autograd.Function):
    """utils.get_recon_cam (utils.py:335-366) with its analytic backward."""

    @staticmethod
    def forward(ctx, spec_mat, relat_cam, intrinsics):
        _need_gpu(spec_mat, relat_cam, intrinsics)
        spec_mat, relat_cam, intrinsics = spec_mat.contiguous(), relat_cam.contiguous(), intrinsics.contiguous().float()
        b, j, _ = relat_cam.shape
        if tuple(spec_mat.shape) != (b, j, 2) or tuple(intrinsics.shape) != (b, 3, 3) or relat_cam.shape[2] != 3:
            raise P3DError('get_recon_cam: spec_mat %s / relat_cam %s / intrinsics %s' % (tuple(spec_mat.shape), tuple(relat_cam.shape), tuple(intrinsics.shape)))
        recon = torch.empty_like(relat_cam)
        check(lib().p3d_recon_cam_fwd(_p(spec_mat), _p(relat_cam), _p(intrinsics), _p(recon), b, j, _stream()), 'p3d_recon_cam_fwd')
        ctx.save_for_backward(spec_mat, relat_cam, intrinsics)
        return recon

    @staticmethod
    def backward(ctx, drecon):
        spec_mat, relat_cam, intrinsics = ctx.saved_tensors
        b, j, _ = relat_cam.shape
        dspec, drelat = torch.empty_like(spec_mat), torch.empty_like(relat_cam)
        check(lib().p3d_recon_cam_bwd(_p(drecon.contiguous()), _p(spec_mat), _p(relat_cam), _p(intrinsics), _p(dspec), _p(drelat), b, j, _stream()),
              'p3d_recon_cam_bwd')
        return dspec, drelat, None


def recon_cam(spec_mat, relat_cam, intrinsics):
    return ReconCamFn.apply(spec_mat, relat_cam, intrinsics)


DISTILL_MODES = {'l2': 0, 'sigmoid': 1, 'bce': 2}


class DistillFn(torch.autograd.Function):
    """Trainer.distill (depth_train.py:115-129): feature-distillation loss of the student against the (no-grad) teacher.
    Returns (weight * loss, loss); the gradient of the first output w.r.t. the student features is produced in the forward
    launch already (scaled by `weight`), so backward is free when the incoming gradient is 1 (loss = cam_loss + weight*dist)."""

    @staticmethod
    def forward(ctx, teach, student, atten, mode, weight, unit_grad):
        _need_gpu(teach, student, atten)
        teach, student, atten = teach.contiguous(), student.contiguous(), atten.contiguous()
        b, c, h, w = student.shape
        if teach.shape != student.shape or tuple(atten.shape) != (b, 1, h, w):
            raise P3DError('distill: teacher %s / student %s / attention %s do not match' % (tuple(teach.shape), tuple(student.shape), tuple(atten.shape)))
        L = lib()
        loss = torch.empty(1, dtype=torch.float32, device=student.device)
        need = ctx.needs_input_grad[1]
        ds = torch.empty_like(student) if need else None
        ws = workspace(student.device, L.p3d_distill_workspace_bytes(b))
        check(L.p3d_distill_fwd_bwd(_p(teach), _p(student), _p(atten), _p(loss), _p(ds), b, c, h * w, DISTILL_MODES[mode], float(weight),
                                    _p(ws), ws.numel(), _stream()), 'p3d_distill_fwd_bwd')
        ctx.save_for_backward(ds)
        ctx.unit_grad = unit_grad
        raw = loss.view(())
        weighted = raw * weight if weight != 1.0 else raw.clone()
        ctx.mark_non_differentiable(raw)
        return weighted, raw

    @staticmethod
    def backward(ctx, dweighted, draw):
        (ds,) = ctx.saved_tensors
        if ds is not None and not ctx.unit_grad:
            ds = ds * dweighted
        return None, ds, None, None, None, None


def distill_loss(teach, student, atten, mode='l2', weight=1.0, unit_grad=False):
    return DistillFn.apply(teach, student, atten, mode, float(weight), unit_grad)


# --------------------------------------------------------------------------------------------
def l2norm_sq_accum(flat, accum):
    check(lib().p3d_l2norm_sq_accum(_p(flat), flat.numel(), _p(accum), _stream()), 'p3d_l2norm_sq_accum')


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, max_norm=0.0, norm_sq=None, grad_scale=1.0):
    check(lib().p3d_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, int(step), float(max_norm),
                              _p(norm_sq), float(grad_scale), _stream()), 'p3d_adam_step')


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, weight_decay, state, max_norm, norm_sq, grad_scale, skip_nonfinite, scratch):
    """Adam with the step counter / overflow skip on the device (state: int32[2] = steps taken, steps skipped)."""
    check(lib().p3d_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, _p(state), float(max_norm),
                                  _p(norm_sq), float(grad_scale), int(skip_nonfinite), _p(scratch), _stream()), 'p3d_adam_step_dev')


def augment_colour_(img, params):
    """In place on img [B,3,H,W] holding 0..255 values; params [B,4] (augment_colour.py:48-67)."""
    _need_gpu(img, params)
    b, c, h, w = img.shape
    if c != 3 or not img.is_contiguous():
        raise P3DError('augment_colour: need a contiguous [B,3,H,W] image')
    check(lib().p3d_augment_colour(_p(img), _p(params.contiguous()), b, h, w, _stream()), 'p3d_augment_colour')
    return img


def augment_erase_(img, rects, colour):
    """In place: fill rects [B,4] int32 (x0,y0,x1,y1) with colour [B,C] (augment_occluder.py:84-105)."""
    _need_gpu(img, colour)
    b, c, h, w = img.shape
    if rects.dtype != torch.int32 or not img.is_contiguous():
        raise P3DError('augment_erase: rects must be int32 and img contiguous')
    check(lib().p3d_augment_erase(_p(img), _p(rects.contiguous()), _p(colour.contiguous()), b, c, h, w, _stream()), 'p3d_augment_erase')
    return img


def augment_occlude_(img, bank, alpha, plan, max_pixels, truncate=True):
    """In place: paste one occluder per image (augment_occluder.py:7-55).  bank [P,C] fp32, alpha [P] fp32 or None, plan [B,8] int32 (augment.plan_paste)."""
    _need_gpu(img, bank, alpha)
    b, c, h, w = img.shape
    if plan.dtype != torch.int32 or not plan.is_cuda or tuple(plan.shape) != (b, 8) or not img.is_contiguous() or bank.shape[-1] != c:
        raise P3DError('augment_occlude: plan must be int32 [B,8] on the device, img contiguous, bank [P,C]')
    check(lib().p3d_augment_occlude(_p(img), _p(bank.contiguous()), _p(None if alpha is None else alpha.contiguous()), _p(plan.contiguous()), b, c, h, w,
                                    int(max_pixels), int(bool(truncate)), _stream()), 'p3d_augment_occlude')
    return img


IMAGENET_MEAN = (0.485, 0.456, 0.406)       # depth_datasets.py:78-79
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize_rgb_(img, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """In place: transforms.ToTensor() + Normalize on a contiguous [B,3,H,W] fp32 device tensor holding 0..255 values."""
    _need_gpu(img)
    b, c, h, w = img.shape
    if c != 3 or not img.is_contiguous():
        raise P3DError('normalize_rgb: need a contiguous [B,3,H,W] image')
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib().p3d_normalize_rgb(_p(img), b, h * w, m3, s3, _stream()), 'p3d_normalize_rgb')
    return img


def warp_crops(frames, homography, out_hw):
    """Batch of crop re-projections (depth_datasets.py:153-193): frames [B,Hs,Ws,C] uint8 or fp32 (interleaved, as decoded),
    homography [B,3,3] fp32 mapping crop pixels to frame pixels -> [B,C,Ho,Wo] fp32 (0..255 values for uint8 sources)."""
    if not frames.is_cuda or frames.dtype not in (torch.uint8, torch.float32) or frames.dim() != 4:
        raise P3DError('warp_crops: frames must be a [B,Hs,Ws,C] uint8 / fp32 tensor on the HIP device')
    _need_gpu(homography)
    frames, homography = frames.contiguous(), homography.contiguous()
    b, hs, ws, c = frames.shape
    if tuple(homography.shape) != (b, 3, 3):
        raise P3DError('warp_crops: homography must be [B,3,3]')
    ho, wo = out_hw
    out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=frames.device)
    check(lib().p3d_warp_crops(_p(frames), int(frames.dtype == torch.uint8), _p(homography), _p(out), b, hs, ws, c, ho, wo, _stream()), 'p3d_warp_crops')
    return out


def set_x3(on):
    """Exact-fp32 convolution kernels on the bf16 matrix pipe (csrc/p3d_fx.hip), ON by default; set_x3(False) (or P3D_X3=0) keeps every layer on the
    fp32-MFMA kernels.  Returns the previous setting."""
    global X3_EPOCH
    X3_EPOCH += 1                       # (ops_block caches per-block plans that depend on it)
    return bool(lib().p3d_x3_enable(int(bool(on))))


X3_EPOCH = 0
WEIGHT_EPOCH = 0          # bumped whenever parameters are written behind torch's back (FlatAdam's kernels, broadcasts into / restores of flat_p): cached per-weight derivatives (ops_block.weight_images) are stale


def weights_changed():
    """Call after writing parameters through anything torch's version counters do not see on the parameter itself: the optimizer kernels, a
    broadcast or copy into FlatAdam.flat_p (the parameters are views of it, and an in-place write to the base does not bump a view's _version)."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def act_image(x, mode=0, x2=None, table=None, masked=False, out=None):
    """Pre-split activation image of a fp32 NCHW tensor (p3d_fx_act_image): uint8 tensor of 6 bytes per element = three bf16 planes [N][C/16][H*W][16].
    mode 0: x; 1: relu(x * sc + sh); 2: A * (masked ? x * [x2 * sc + sh > 0] : x) + B * x2 + K, constants per channel from `table` [C][8]."""
    _need_gpu(x)
    x = x.contiguous()
    n, c, h, w = x.shape
    nbytes = lib().p3d_fx_act_image_bytes(n, c, h * w)
    img = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if out is None else out
    if img.numel() != nbytes or img.dtype != torch.uint8 or not img.is_contiguous():
        raise P3DError('act_image: `out` must be a contiguous uint8 tensor of %d bytes' % nbytes)
    check(lib().p3d_fx_act_image(mode, _p(x), None if x2 is None else _p(x2.contiguous()), None if table is None else _p(table.contiguous()), int(bool(masked)),
                                 _p(img), n, c, h * w, _stream()), 'p3d_fx_act_image')
    return img


def conv2d_img(pass_, x_shape, w, stride, pad, dil, x_img=None, dy_img=None, x=None, bias=None, accumulate_into=None):
    """One pass of a convolution on pre-split image operands (p3d_fx_conv_*_img; what the block executor launches): 'fwd' -> y from x_img,
    'dgrad' -> dx from dy_img, 'wgrad' -> dw from dy_img and x_img (or the fp32 x).  For tests and tools/conv_bench.py."""
    _need_gpu(w)
    w = w.contiguous()
    d = _desc(x_shape, w.shape, stride, pad, dil)
    which = {'fwd': 0, 'dgrad': 1, 'wgrad': 2}[pass_]
    ws = workspace(w.device, lib().p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), which))
    if which == 0:
        y = torch.empty((d.N, d.K, d.Ho, d.Wo), dtype=torch.float32, device=w.device)
        check(lib().p3d_fx_conv_fwd_img(ctypes.byref(d), _p(x_img), _p(w), None, None if bias is None else _p(bias), _p(y), _p(ws), ws.numel(), _stream()), 'p3d_fx_conv_fwd_img')
        return y
    if which == 1:
        dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=w.device) if accumulate_into is None else accumulate_into
        d.accumulate = int(accumulate_into is not None)
        check(lib().p3d_fx_conv_dgrad_img(ctypes.byref(d), _p(dy_img), _p(w), None, _p(dx), _p(ws), ws.numel(), _stream()), 'p3d_fx_conv_dgrad_img')
        return dx
    dw = torch.empty_like(w)
    check(lib().p3d_fx_conv_wgrad_img(ctypes.byref(d), _p(dy_img), None if x is None else _p(x.contiguous()), None if x_img is None else _p(x_img), _p(dw), _p(ws),
                                      ws.numel(), _stream()), 'p3d_fx_conv_wgrad_img')
    return dw


def profile_convs(on):
    """HIP-event brackets around every conv launch inside the library (p3d_profile_enable); returns the previous setting."""
    return lib().p3d_profile_enable(int(on))          # 0 off, 1 brackets, 2 brackets + kernel-end marks (collect_conv_profile(kernel_only=True))


def collect_conv_profile(kernel_only=False):
    """Synchronises; dict(kind -> (ms, algorithmic flops, launches)) for 'fwd', 'dgrad', 'wgrad' since the last collect.  kernel_only: (ms, flops, launches,
    ms of the conv kernels alone -- without the split-K / slab sums queued behind them)."""
    ms, kms, fl, n = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 3)()
    check(lib().p3d_profile_collect2(ms, kms, fl, n), 'p3d_profile_collect2')
    if kernel_only:
        return {k: (float(ms[i]), float(fl[i]), int(n[i]), float(kms[i])) for i, k in enumerate(('fwd', 'dgrad', 'wgrad'))}
    return {k: (float(ms[i]), float(fl[i]), int(n[i])) for i, k in enumerate(('fwd', 'dgrad', 'wgrad'))}


def conv_path_stats(reset=False):
    """Conv launches and algorithmic flops per path since the last reset: dict(x3=dict(fwd, dgrad, wgrad), fp32=dict(...)), each (launches, flops)."""
    counts, flops = (ctypes.c_uint64 * 6)(), (ctypes.c_double * 6)()
    lib().p3d_conv_path_stats(counts, flops, int(bool(reset)))
    names = ('fwd', 'dgrad', 'wgrad')
    return dict(x3={n: (int(counts[i]), float(flops[i])) for i, n in enumerate(names)},
                fp32={n: (int(counts[3 + i]), float(flops[3 + i])) for i, n in enumerate(names)})


def reproject_crops(frames, params20, out_hw, round_u8=True):
    """Batch of cameralib.reproject_image calls (cameralib.py:378-443): frames [B,Hs,Ws,C] uint8 / fp32 as decoded, params20 [B,20] fp32
    (cameralib.reproject_params) -> [B,C,Ho,Wo] fp32."""
    if not frames.is_cuda or frames.dtype not in (torch.uint8, torch.float32) or frames.dim() != 4:
        raise P3DError('reproject_crops: frames must be a [B,Hs,Ws,C] uint8 / fp32 tensor on the HIP device')
    _need_gpu(params20)
    frames, params20 = frames.contiguous(), params20.contiguous()
    b, hs, ws, c = frames.shape
    if tuple(params20.shape) != (b, 20) or params20.dtype != torch.float32:
        raise P3DError('reproject_crops: params20 must be fp32 [B,20]')
    ho, wo = out_hw
    out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=frames.device)
    check(lib().p3d_reproject_crops(_p(frames), int(frames.dtype == torch.uint8), _p(params20), _p(out), b, hs, ws, c, ho, wo, int(bool(round_u8)),
                                    _stream()), 'p3d_reproject_crops')
    return out


def enhance_depth_(crops, threshold, nexponent, factor=None):
    """In place: depth_datasets.enhance_ntu / enhance_pku (depth_datasets.py:39-56) on PNG-unit depth crops, after the optional utils.to_depth
    division by `factor` (same shape)."""
    _need_gpu(crops, factor)
    if crops.dtype != torch.float32 or not crops.is_contiguous() or (factor is not None and (factor.shape != crops.shape or not factor.is_contiguous())):
        raise P3DError('enhance_depth_: contiguous fp32 crops (and factor of the same shape) expected')
    check(lib().p3d_enhance_depth(_p(crops), _p(factor), crops.numel(), float(threshold), int(bool(nexponent)), _stream()), 'p3d_enhance_depth')
    return crops


def conv_bn_eval(x, conv, bn, res=None, relu=False):
    """Inference only (no autograd): conv (no bias) + BatchNorm with frozen statistics (+ residual, + ReLU) as one kernel launch
    (model.eval() forward of a residual block, depthnet.py:42-56,98-116)."""
    from .nn import _one
    _need_gpu(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, res)
    x = x.contiguous()
    res = None if res is None else res.contiguous()
    d = _desc(x.shape, conv.weight.shape, _one(conv.stride), _one(conv.padding), _one(conv.dilation))
    y = torch.empty((d.N, d.K, d.Ho, d.Wo), dtype=torch.float32, device=x.device)
    L = lib()
    ws = workspace(x.device, L.p3d_conv2d_bn_eval_fwd_workspace_bytes(ctypes.byref(d)))
    check(L.p3d_conv2d_bn_eval_fwd(ctypes.byref(d), _p(x), _p(conv.weight.detach()), _p(bn.weight.detach()), _p(bn.bias.detach()), _p(bn.running_mean),
                                   _p(bn.running_var), bn.eps, _p(res), int(relu), _p(y), _p(ws), ws.numel(), _stream()), 'p3d_conv2d_bn_eval_fwd')
    return y


def can_fuse_eval(x, conv, bn):
    """The fused inference kernel applies when nothing needs a gradient, BN uses its running statistics and the conv is plain."""
    return (not torch.is_grad_enabled() and not bn.training and x.dtype == torch.float32 and conv.bias is None
            and type(conv).__name__ == 'Conv2d' and bn.track_running_stats)
