"""One autograd node per residual block: `residual_block(block, x)` runs depthnet.BasicBlock / Bottleneck.forward (depthnet.py:40-56,96-116)
through p3d_block_fwd / p3d_block_bwd (csrc/p3d_block.hip): the BatchNorm layers between two convolutions live inside the convolution kernels,
and the host makes one C call per block and direction instead of a dozen per-layer autograd nodes.

Used by _trunk._ResidualBlock.forward for dense fp32 blocks whose BatchNorm layers are in training mode; everything else (partial convolutions,
-half_acc, frozen / evaluation BatchNorm, shapes the fused kernels do not take) stays on the per-layer path of ops.py.
"""
import ctypes
import os

import torch

from . import ops
from ._lib import ConvDesc, P3DError, check, lib

_vp = ctypes.c_void_p

# P3D_BLOCK_SIDE=0: keep the weight-gradient kernels of the block executor on the launch stream.  By default they run on the second HIP stream
# (p3d_block_bwd orders them with events), as in the per-layer path; the step is bitwise reproducible either way
# (tests/test_step_gpu.py::test_training_is_bitwise_reproducible).
BLOCK_SIDE_STREAM = os.environ.get('P3D_BLOCK_SIDE', '1') != '0'
# P3D_OUT_MASK=0: the backward pass reads the block's fp32 output for the closing ReLU's mask instead of the mask bytes forward leaves (p3d_block_io.out_mask = NULL)
USE_OUT_MASK = os.environ.get('P3D_OUT_MASK', '1') != '0'
# P3D_TAIL_SUMS=1 (opt-in, measured SLOWER: profiles/r04_summary.md section 3): a block takes the channel sums its backward pass opens with from the epilogue of the data
# gradient that wrote dout (the consumer block's p3d_block_io.tail_*) instead of its own pass over dout (block_open_bwd).  13 of ResNet-50's 16 opening passes go away, but
# the data gradient's epilogue then waits on uncoalesced reads of c_last / c_ds / the mask bytes: the step is 0.2 - 0.4 ms LONGER.  Kept tested, off by default.
USE_TAIL_SUMS = os.environ.get('P3D_TAIL_SUMS', '0') == '1'
TAIL_ROWS = 16                   # P3D_TAIL_ROWS
TAIL_STATS = {'reduced': 0, 'opening_passes_skipped': 0}     # (tests: how often a consumer reduced its producer's sums / a producer found them valid)


class BlockDesc(ctypes.Structure):
    """struct p3d_block_desc"""
    _fields_ = [('nconv', ctypes.c_int32), ('has_downsample', ctypes.c_int32), ('relu_out', ctypes.c_int32), ('need_dx', ctypes.c_int32),
                ('accumulate_grads', ctypes.c_int32), ('masked', ctypes.c_int32), ('reserved', ctypes.c_int32 * 2), ('eps', ctypes.c_float * 4), ('momentum', ctypes.c_float * 4),
                ('conv', ConvDesc * 4)]


class BlockIO(ctypes.Structure):
    """struct p3d_block_io"""
    _fields_ = [('x', _vp), ('out', _vp), ('w', _vp * 4), ('wimg', _vp * 4), ('wimgT', _vp * 4), ('c', _vp * 4), ('aimg', _vp * 4), ('table', _vp * 4), ('gamma', _vp * 4), ('beta', _vp * 4),
                ('running_mean', _vp * 4), ('running_var', _vp * 4), ('dout', _vp), ('gbuf', _vp), ('dcimg', _vp * 4), ('da', _vp * 4), ('dx', _vp), ('dw', _vp * 4),
                ('dgamma', _vp * 4), ('dbeta', _vp * 4), ('out_mask', _vp),
                ('tail_c_last', _vp), ('tail_table_last', _vp), ('tail_c_ds', _vp), ('tail_table_ds', _vp), ('tail_mask', _vp), ('tail_partial', _vp), ('tail_sums', _vp),
                ('open_sums', _vp), ('pix_in', _vp * 4), ('pix_out', _vp * 4)]


def _one(v):
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


def _layers(block):
    """[(slot, conv, bn)]: slots 0..nconv-1 the main chain, slot 3 the downsample pair."""
    pairs = [(i, getattr(block, c), getattr(block, b)) for i, (c, b) in enumerate(block._chain)]
    if block.downsample is not None:
        pairs.append((3, block.downsample[0], block.downsample[1]))
    return pairs


class _Plan:
    """Per (block, input shape): the descriptor, workspace sizes and output shapes.  None if the fused executor does not take the block."""

    def __init__(self, block, x_shape, masked=False):
        d = BlockDesc()
        d.nconv = len(block._chain)
        d.masked = int(masked)
        d.has_downsample = int(block.downsample is not None)
        d.relu_out = int(not block.skip_relu)
        shape = tuple(x_shape)
        self.shapes = {}
        for slot, conv, bn in _layers(block):
            src = tuple(x_shape) if slot in (0, 3) else shape
            cd = ops._desc(src, conv.weight.shape, _one(conv.stride), _one(conv.padding), _one(conv.dilation))
            d.conv[slot] = cd
            d.eps[slot] = bn.eps
            d.momentum[slot] = 0.1 if bn.momentum is None else bn.momentum
            self.shapes[slot] = (cd.N, cd.K, cd.Ho, cd.Wo)
            if slot != 3:
                shape = self.shapes[slot]
        self.desc = d
        self.out_shape = shape
        self.ok = bool(lib().p3d_block_supported(ctypes.byref(d))) and (block.downsample is not None or not block.skip_relu)
        if self.ok and block.downsample is not None and self.shapes[3] != shape:
            self.ok = False
        self.main_bytes = self.side_bytes = 0
        if self.ok:
            m, s = ctypes.c_size_t(), ctypes.c_size_t()
            check(lib().p3d_block_workspace_bytes(ctypes.byref(d), ctypes.byref(m), ctypes.byref(s)), 'p3d_block_workspace_bytes')
            self.main_bytes, self.side_bytes = m.value, s.value
        ks = [self.shapes[slot][1] for slot, _, _ in _layers(block)]
        self.table_rows = sum(ks)
        self.slots = [slot for slot, _, _ in _layers(block)]
        self.sets = []                  # _Buffers owned by this plan (see there)
        # as a consumer: can this block's backward pass reduce the opening sums of the block that produced its input? (p3d_block_io.tail_*)
        self.tail_ok = bool(self.ok and lib().p3d_block_tail_supported(ctypes.byref(d)))
        self.tail_bytes = int(lib().p3d_block_tail_partial_bytes(ctypes.byref(d))) if self.tail_ok else 0
        self.in_shape = tuple(x_shape)

    def acquire(self, device):
        """A free buffer set (or a new one).  Two are kept per plan (a block that runs twice before its backward: labelled + unlabelled batch of semi_train);
        a third concurrent execution gets a set that dies with its autograd node."""
        for b in self.sets:
            if not b.held and b.device == device:
                break
        else:
            b = _Buffers(self, device)
            if len(self.sets) < 2:
                self.sets.append(b)
        if b.side_done is not None:     # the previous backward's weight gradients (second stream) read these buffers: order this stream behind them
            torch.cuda.current_stream(device).wait_event(b.side_done)
        return b


_spare_shapes = set()


class _Buffers:
    """The device memory of ONE execution of a block, owned by its plan and used again by the next step: the conv outputs c_i, the activation images a_i and the
    BatchNorm tables (written by forward, read by backward), and the backward's own scratch (gradient images, the fp32 gradients between layers).  Taking these
    from the caching allocator per call cost more than its bookkeeping: tensors the weight-gradient stream reads must be `record_stream`ed, which makes their
    re-use depend on GPU timing, and every so often a step found no free block and sat in hipMalloc -- 2.4 - 2.8 s on some boxes of the pool, once per
    process, in the middle of the timed region (profiles/r03_summary.md section 8)."""

    def __init__(self, plan, device):
        self.device, self.held, self.side_done, self.bwd = device, False, None, None
        self.tail = None                # (partial scratch, sums [C][TAIL_ROWS][3] fp64): where a consumer block leaves this block's opening sums
        self.open_ready = None          # (data_ptr, _version) of the gradient tensor those sums were reduced over
        f32 = dict(dtype=torch.float32, device=device)
        self.tables = torch.empty((plan.table_rows, 8), **f32)
        self.c = {slot: torch.empty(plan.shapes[slot], **f32) for slot in plan.slots}
        # a_slot = relu(bn(c_slot)) exists only as a pre-split image (three bf16 planes: 6 B per element)
        self.act = {slot: torch.empty(6 * self.c[slot].numel(), dtype=torch.uint8, device=device) for slot in plan.slots if slot < plan.desc.nconv - 1}
        # which outputs the closing ReLU let through: one byte per four elements, all the backward pass needs of `out`
        n_out = plan.out_shape[0] * plan.out_shape[1] * plan.out_shape[2] * plan.out_shape[3]
        self.mask = torch.empty(n_out // 4, dtype=torch.uint8, device=device) if (plan.desc.relu_out and USE_OUT_MASK) else None
        # the block's output (= the next block's input) and its gradient stay with the caching allocator; a little slack per distinct shape keeps the few
        # `record_stream`ed tensors that remain (the block input) from ever forcing a hipMalloc in steady state
        key = (plan.out_shape, str(device))
        if key not in _spare_shapes:
            _spare_shapes.add(key)
            spare = [torch.empty(plan.out_shape, **f32) for _ in range(3)]
            del spare

    def tail_buffers(self, plan, nbytes):
        if self.tail is None or self.tail[0].numel() < nbytes:
            self.tail = (torch.empty(nbytes, dtype=torch.uint8, device=self.device),
                         torch.empty((plan.out_shape[1], TAIL_ROWS, 3), dtype=torch.float64, device=self.device))
        return self.tail

    def backward_scratch(self, plan):
        if self.bwd is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            dcimg = {slot: torch.empty(6 * self.c[slot].numel(), dtype=torch.uint8, device=self.device) for slot in plan.slots}    # image of d c_slot: what conv slot's wgrad and dgrad read
            da = {slot: torch.empty(plan.shapes[slot], **f32) for slot in plan.slots if slot < plan.desc.nconv - 1}
            # g = dout * [out > 0]: with an identity shortcut it IS dx (a fresh tensor per call); with a downsample branch and mask bytes nobody needs it as a tensor
            gbuf = torch.empty(plan.out_shape, **f32) if (plan.desc.has_downsample and self.mask is None) else None
            self.bwd = (dcimg, da, gbuf)
        return self.bwd


class _OwnedBuffers:
    """Named byte buffers owned by one module for one execution in flight (the per-layer twins of _Buffers: the operand images of ConvImagesFn / StemConvFn)."""

    def __init__(self, device):
        self.device, self.held, self.side_done, self.t = device, False, None, {}

    def tensor(self, name, nbytes):
        t = self.t.get(name)
        if t is None or t.numel() != nbytes:
            t = self.t[name] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return t

    def mark_side(self, side):
        if self.side_done is None:
            self.side_done = torch.cuda.Event()
        self.side_done.record(side)


def _owned(module, key, device):
    cache = module.__dict__.get('_owned_bufs')
    if cache is None:
        cache = module.__dict__['_owned_bufs'] = _Transient()
    sets = cache.pop(key, None)
    if sets is None:
        if len(cache) >= PLAN_LIMIT:
            for old in list(cache):
                if not any(b.held for b in cache[old]):
                    del cache[old]
                    break
        sets = []
    cache[key] = sets
    for b in sets:
        if not b.held and b.device == device:
            break
    else:
        b = _OwnedBuffers(device)
        if len(sets) < 2:
            sets.append(b)
    if b.side_done is not None:
        torch.cuda.current_stream(device).wait_event(b.side_done)
    return b


class _Lease:
    """Marks a buffer set as in use for the life of an autograd node (released when the node dies: after backward, or when the graph is dropped)."""

    def __init__(self, bufs):
        self.bufs = bufs
        bufs.held = True

    def __del__(self):
        self.bufs.held = False


# Every convolution that has asked for weight images is registered here; when one of them finds its images stale (the optimizer stepped), ALL stale ones are
# rebuilt by ONE launch (p3d_fx_weight_images_batched): 54 launches of a few microseconds each otherwise sit on the forward critical path of ResNet-50.
_image_convs = []            # weak references, in order of first use
_image_members = None        # WeakSet of the same modules (a deep copy of a registered module carries `_fx_images` along but is not a member)
_image_tables = {}           # (conv ids) -> device job table


def _image_key(w):
    return (w._version, ops.WEIGHT_EPOCH, w.data_ptr())


def _rebuild_stale_images(device):
    import weakref
    import numpy as np
    live = []
    for ref in _image_convs:
        conv = ref()
        if conv is not None and '_fx_images' in conv.__dict__:
            live.append(conv)
    _image_convs[:] = [weakref.ref(c) for c in live]
    stale = [c for c in live if c.weight.device == device and c.__dict__['_fx_images'][0] != _image_key(c.weight)]
    if not stale:
        return
    # the job table: its bytes are also the cache key (Python ids and device addresses are both re-used after a module dies, so only the full record identifies a job)
    rows = np.zeros(len(stale), dtype=[('w', '<u8'), ('f', '<u8'), ('b', '<u8'), ('K', '<i4'), ('C', '<i4'), ('RS', '<i4'), ('pad', '<i4')])
    most = 0
    for i, c in enumerate(stale):
        k, cc, r, s_ = c.weight.shape
        _, fwd, bwd = c.__dict__['_fx_images']
        rows[i] = (c.weight.data_ptr(), fwd.data_ptr(), bwd.data_ptr(), k, cc, r * s_, 0)
        most = max(most, fwd.numel(), bwd.numel())
    assert rows.dtype.itemsize == 40
    ident = rows.tobytes()
    entry = _image_tables.get(ident)
    if entry is None:
        table = torch.from_numpy(rows.view(np.uint8).reshape(-1).copy()).to(device)
        blocks = max(1, min(1024, (most // 48 + 255) // 256))         # 16-B chunk positions of the largest image / 256 threads, capped (blocks stride; measured: 256 -> 195 us, 1024 -> 168, 4096 -> 202 per ResNet-50 rebuild)
        if len(_image_tables) > 16:
            _image_tables.clear()
        entry = _image_tables[ident] = (table, blocks)
    table, blocks = entry
    check(lib().p3d_fx_weight_images_batched(ops._p(table), len(stale), blocks, ops._stream()), 'p3d_fx_weight_images_batched')
    for c in stale:
        _, fwd, bwd = c.__dict__['_fx_images']
        c.__dict__['_fx_images'] = (_image_key(c.weight), fwd, bwd)


# P3D_IMAGES_EARLY=1: the optimizer step queues the rebuild of every weight image on the second stream right behind the Adam kernel, so that it runs beside the next
# step's stem (whose own small image is not part of the batch) instead of on the launch stream in front of the first residual block.
IMAGES_EARLY = os.environ.get('P3D_IMAGES_EARLY', '0') == '1'
_images_event = {}


def rebuild_images_early(device):
    if not (IMAGES_EARLY and ops.WGRAD_STREAM and USE_WEIGHT_IMAGES) or torch.cuda.is_current_stream_capturing():
        return
    side = ops._side_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))            # the optimizer's kernels
    with torch.cuda.stream(side):
        _rebuild_stale_images(device)
    ev = torch.cuda.Event()
    ev.record(side)
    _images_event[device] = ev


def weight_images(conv):
    """(forward image, data-gradient image) of conv.weight as device byte tensors (p3d_fx_weight_images*), rebuilt when the weight has changed:
    in-place edits through torch bump weight._version; writes behind torch's back (the optimizer kernels, broadcasts into the flat buffer) bump ops.WEIGHT_EPOCH."""
    import weakref
    w = conv.weight
    if _images_event:
        ev = _images_event.pop(w.device, None)
        if ev is not None:
            torch.cuda.current_stream(w.device).wait_event(ev)
    cached = conv.__dict__.get('_fx_images')
    if cached is not None and cached[0] == _image_key(w):
        return cached[1], cached[2]
    global _image_members
    if _image_members is None:
        _image_members = weakref.WeakSet()
    if cached is None or conv not in _image_members or cached[1].device != w.device:
        k, c, r, s_ = w.shape
        fb, bb = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().p3d_fx_weight_image_bytes(k, c, r * s_, ctypes.byref(fb), ctypes.byref(bb)), 'p3d_fx_weight_image_bytes')
        fwd = torch.empty(fb.value, dtype=torch.uint8, device=w.device)
        bwd = torch.empty(bb.value, dtype=torch.uint8, device=w.device)
        conv.__dict__['_fx_images'] = (None, fwd, bwd)
        if conv not in _image_members:
            _image_members.add(conv)
            _image_convs.append(weakref.ref(conv))
    _rebuild_stale_images(w.device)
    cached = conv.__dict__['_fx_images']
    return cached[1], cached[2]


USE_WEIGHT_IMAGES = os.environ.get('P3D_WEIGHT_IMAGES', '1') != '0'


PLAN_LIMIT = 3


class _Transient(dict):
    """Per-module caches of plans and device buffers: never copied or pickled with the module (a deep copy starts with an empty cache)."""

    def __deepcopy__(self, memo):
        return _Transient()

    def __reduce__(self):
        return (_Transient, ())


def _buffer_sets(model):
    """Every buffer set (_Buffers, _OwnedBuffers) the modules of `model` own."""
    for module in model.modules():
        for plan in (module.__dict__.get('_blk_plans') or {}).values():
            for b in plan.sets:
                yield b
        for sets in (module.__dict__.get('_owned_bufs') or {}).values():
            for b in sets:
                yield b


def reset_side_events(model):
    """Forget the "last reader on the weight-gradient stream" events of every buffer set.  Call with the device idle (after a synchronize): an event last recorded
    INSIDE a stream capture must not be waited on by eager work or by the next capture (graphed.GraphedStep re-captures on every learning-rate change)."""
    for b in _buffer_sets(model):
        b.side_done = None


def release_buffers(model):
    """Give the plan-owned device memory of `model` back to the allocator (forward buffers, activation / gradient images, backward scratch: ~13 GB for ResNet-50 at
    batch 64): for a caller that switches to evaluation or ends an epoch and wants the memory; the next training step builds the plans again.
    Buffer sets still held by a live autograd graph stay alive with it.  Returns the number of plans dropped."""
    dropped = 0
    for module in model.modules():
        for name in ('_blk_plans', '_owned_bufs'):
            cache = module.__dict__.get(name)
            if cache:
                dropped += len(cache)
                cache.clear()
    return dropped


def plan_for(block, x, masked=False):
    cache = block.__dict__.get('_blk_plans')
    if cache is None:
        cache = block.__dict__['_blk_plans'] = _Transient()
    key = (tuple(x.shape), ops.X3_EPOCH, bool(masked))
    plan = cache.pop(key, None)
    if plan is None:
        # a plan owns device buffers: keep those of the PLAN_LIMIT most recently used input shapes (a training loop has one or two: the batch and the epoch's
        # last, smaller one), drop the oldest idle one beyond that
        if len(cache) >= PLAN_LIMIT:
            for old in list(cache):
                if not any(b.held for b in cache[old].sets):
                    del cache[old]
                    break
        plan = _Plan(block, x.shape, masked)
    cache[key] = plan               # (re-inserted: the dict's order is the order of last use)
    return plan


# P3D_MASKED_BLOCKS=0: the partial-convolution blocks of the partial families stay on the per-layer path (one autograd node per conv / BatchNorm; A/B)
MASKED_BLOCKS = os.environ.get('P3D_MASKED_BLOCKS', '1') != '0'


def usable(block, x, veil=None):
    """The fused executor takes this call: fp32 block on the GPU, every BatchNorm computing batch statistics, supported shapes; dense, or (veil given) with
    partial convolutions in its main chain (partial_depthnet.py:62-75,140-157; the downsample branch is a dense conv there too)."""
    masked = veil is not None
    if bool(block.partial) != masked or x.dtype != torch.float32 or not x.is_cuda or (masked and not MASKED_BLOCKS):
        return False
    if masked and not (veil.is_cuda and veil.dtype == torch.float32 and veil.dim() == 4 and veil.shape[1] == 1 and veil.shape[0] == x.shape[0] and veil.shape[2:] == x.shape[2:]):
        return False
    for slot, conv, bn in _layers(block):
        want = 'PartialConv' if (masked and slot != 3) else 'Conv2d'
        if not bn.training or not (bn.affine and bn.track_running_stats) or conv.bias is not None or type(conv).__name__ != want:
            return False
    return plan_for(block, x, masked).ok


class ResidualBlockFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, block, veil, *params):
        x = x.contiguous()
        plan = plan_for(block, x, veil is not None)
        layers = _layers(block)
        L = lib()
        io = BlockIO()
        io.x = x.data_ptr()
        out = torch.empty(plan.out_shape, dtype=torch.float32, device=x.device)
        io.out = out.data_ptr()
        bufs = plan.acquire(x.device)
        lease = _Lease(bufs)
        bufs.open_ready = None
        # the block whose output this input is (residual_block tags its result): its buffers, for the tail sums of backward
        ctx.producer = getattr(x, '_p3d_block_out', None) if (USE_TAIL_SUMS and plan.tail_ok) else None
        tables, cs, acts, row = bufs.tables, bufs.c, bufs.act, 0
        if bufs.mask is not None:
            io.out_mask = bufs.mask.data_ptr()
        # partial convolutions: the veil chain of the block (partial_conv.py:35-43, one tiny box-sum kernel per conv) gives every conv its two per-pixel factors
        pix = None
        veil_out = None
        if veil is not None:
            pix = {}
            v = veil.contiguous()
            with torch.no_grad():
                for slot, conv, _ in layers:
                    if slot == 3:
                        continue
                    mult, v_next = ops.mask_count(v, _one(conv.kernel_size), _one(conv.stride), _one(conv.padding), _one(conv.dilation))
                    pix[slot] = (v, mult)
                    io.pix_in[slot], io.pix_out[slot] = v.data_ptr(), mult.data_ptr()
                    v = v_next
            veil_out = v
        ctx.pix = pix
        for slot, conv, bn in layers:
            io.w[slot], io.c[slot] = conv.weight.data_ptr(), cs[slot].data_ptr()
            if USE_WEIGHT_IMAGES:
                io.wimg[slot] = weight_images(conv)[0].data_ptr()
            if slot in acts:
                io.aimg[slot] = acts[slot].data_ptr()
            io.table[slot] = tables.data_ptr() + row * 32
            row += plan.shapes[slot][1]
            io.gamma[slot], io.beta[slot] = bn.weight.data_ptr(), bn.bias.data_ptr()
            io.running_mean[slot], io.running_var[slot] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            if getattr(bn, '_ticked', False):             # counted by the network's pre-hook (_trunk._tick_batchnorm)
                bn._ticked = False
            else:
                bn.num_batches_tracked.add_(1)
        ws = ops.workspace(x.device, plan.main_bytes)
        check(L.p3d_block_fwd(ctypes.byref(plan.desc), ctypes.byref(io), ops._p(ws), ws.numel(), ops._stream()), 'p3d_block_fwd')
        ctx.block, ctx.plan = block, plan
        ctx.saved = (lease, bufs)                          # (a plain attribute: the plan's buffers are never inputs / outputs of another node)
        ctx.save_for_backward(x, out)
        block.__dict__['_last_exec'] = (bufs, plan)        # residual_block() tags the returned tensor with it
        if veil is not None:
            ctx.mark_non_differentiable(veil_out)
            return out, veil_out
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        block, plan = ctx.block, ctx.plan
        x, out = ctx.saved_tensors
        if ctx.saved is None:
            raise P3DError('residual_block: backward called a second time on the same graph (retain_graph / re-entrant checkpointing): the block executor '
                           'releases its saved activations after the first backward; run the forward again')
        lease, bufs = ctx.saved
        ctx.saved = None
        cs, tables, acts = bufs.c, bufs.tables, bufs.act
        layers = _layers(block)
        L = lib()
        dout = dout.contiguous()
        d = plan.desc
        need_dx = bool(ctx.needs_input_grad[0])
        params = []
        for slot, conv, bn in layers:
            params += [(slot, 'dw', conv.weight), (slot, 'dgamma', bn.weight), (slot, 'dbeta', bn.bias)]
        sinks = [ops._grad_sink(p) for _, _, p in params]
        direct = all(s is not None for s in sinks)
        grads = sinks if direct else [torch.empty_like(p) for _, _, p in params]
        io = BlockIO()
        io.x, io.out, io.dout = x.data_ptr(), out.data_ptr(), dout.data_ptr()
        if bufs.mask is not None:
            io.out_mask = bufs.mask.data_ptr()
        if ctx.pix is not None:
            for slot, (v, mult) in ctx.pix.items():
                io.pix_in[slot], io.pix_out[slot] = v.data_ptr(), mult.data_ptr()
        # This block as the producer: the block that consumed `out` reduced the opening sums over the gradient it wrote -- valid if that very tensor arrives here
        # untouched (another consumer's gradient would have been added by autograd: a new tensor, or an in-place add that bumps the version).
        if bufs.open_ready is not None and bufs.open_ready == (dout.data_ptr(), dout._version) and bufs.tail is not None:
            io.open_sums = bufs.tail[1].data_ptr()
            TAIL_STATS['opening_passes_skipped'] += 1
        bufs.open_ready = None
        # This block as the consumer: hand the producer's tensors to the data gradient that writes dx last.
        prod = ctx.producer
        ctx.producer = None
        tail_for = None
        if prod is not None and need_dx:
            pb, pp = prod
            plast = pp.desc.nconv - 1
            if pb.held and pp.out_shape == plan.in_shape and pb.device == x.device and (pb.mask is not None) == bool(pp.desc.relu_out):
                prow = 0
                ptab = {}
                for slot in pp.slots:
                    ptab[slot] = pb.tables.data_ptr() + prow * 32
                    prow += pp.shapes[slot][1]
                scratch, sums = pb.tail_buffers(pp, plan.tail_bytes)
                io.tail_c_last, io.tail_table_last = pb.c[plast].data_ptr(), ptab[plast]
                if pp.desc.has_downsample:
                    io.tail_c_ds, io.tail_table_ds = pb.c[3].data_ptr(), ptab[3]
                if pb.mask is not None:
                    io.tail_mask = pb.mask.data_ptr()
                io.tail_partial, io.tail_sums = scratch.data_ptr(), sums.data_ptr()
                tail_for = pb
                TAIL_STATS['reduced'] += 1
        row = 0
        for slot, conv, bn in layers:
            io.w[slot], io.c[slot] = conv.weight.data_ptr(), cs[slot].data_ptr()
            if USE_WEIGHT_IMAGES:
                io.wimgT[slot] = weight_images(conv)[1].data_ptr()
            if slot in acts:
                io.aimg[slot] = acts[slot].data_ptr()
            io.table[slot] = tables.data_ptr() + row * 32
            row += plan.shapes[slot][1]
            io.gamma[slot], io.beta[slot] = bn.weight.data_ptr(), bn.bias.data_ptr()
        for (slot, kind, _), g in zip(params, grads):
            getattr(io, kind)[slot] = g.data_ptr()
        dcimg, da, gbuf = bufs.backward_scratch(plan)
        if gbuf is None and not (d.has_downsample and bufs.mask is not None):
            gbuf = torch.empty_like(out)
        if gbuf is not None:
            io.gbuf = gbuf.data_ptr()
        for slot, _, _ in layers:
            io.dcimg[slot] = dcimg[slot].data_ptr()
            if slot in da:
                io.da[slot] = da[slot].data_ptr()
        dx = None
        if need_dx:
            if d.has_downsample:
                dx = torch.empty_like(x)
                io.dx = dx.data_ptr()
            else:
                dx = gbuf                                   # identity shortcut: conv1's data gradient is added onto g in place
        desc = BlockDesc.from_buffer_copy(d)
        desc.need_dx, desc.accumulate_grads = int(need_dx), int(direct)
        ws = ops.workspace(x.device, plan.main_bytes)
        two = ops.WGRAD_STREAM and direct and BLOCK_SIDE_STREAM
        if two:
            side = ops._side_stream(x.device)
            ops._queue_join()
            sws = ops._side_workspace(x.device, plan.side_bytes)
            side_handle = _vp(side.cuda_stream)
        else:
            sws = ops._second_workspace(x.device, plan.side_bytes)
            side_handle = None
        check(L.p3d_block_bwd(ctypes.byref(desc), ctypes.byref(io), ops._p(ws), ws.numel(), ops._p(sws), sws.numel(), ops._stream(), side_handle), 'p3d_block_bwd')
        if tail_for is not None and dx is not None:
            tail_for.open_ready = (dx.data_ptr(), dx._version)      # the producer's backward checks that this is what it receives
        if two:
            x.record_stream(side)          # the one allocator-owned tensor the second stream reads (first conv's and the downsample's weight gradients)
            if ctx.pix is not None:
                ctx.pix[0][0].record_stream(side)          # (and, for a masked block, conv 1's mask_in: its weight gradient multiplies x by it)
            if bufs.side_done is None:
                bufs.side_done = torch.cuda.Event()
            bufs.side_done.record(side)    # the plan's buffers: their next user (plan.acquire) orders itself behind this
        del lease
        if direct:
            for _, _, p in params:
                ops._grad_done(p)
            return (dx, None, None) + (None,) * len(params)
        return (dx, None, None) + tuple(grads)


def residual_block(block, x, veil=None):
    """One autograd node for the whole block.  veil: the block's input mask (a block of partial convolutions); returns (out, veil_out) then."""
    params = []
    for _, conv, bn in _layers(block):
        params += [conv.weight, bn.weight, bn.bias]
    res = ResidualBlockFn.apply(x, block, veil, *params)
    out = res[0] if veil is not None else res
    last = block.__dict__.pop('_last_exec', None)
    if last is not None and USE_TAIL_SUMS:
        out._p3d_block_out = last        # a consumer block that receives THIS tensor object as its input may reduce our opening sums in its backward pass
    return res


# ---- a single convolution on image operands (the 3x3 `regressor` behind layer4: depthnet.py:156,199) ------------------------------------------------------
# Multi-tap and strided convolutions gain most from operands that are split once instead of per channel tile and filter tap (profiles/r03_summary.md:
# ResNet-50's 2048 -> 272 3x3 regressor 3.33 -> 2.78 ms per step over its three passes, image passes included); plain 1x1 layers do not repay the extra
# pass, so they stay on ops.conv2d.
IMAGE_CONVS = os.environ.get('P3D_IMAGE_CONVS', '1') != '0'


def conv_takes_images(conv, x):
    if not (IMAGE_CONVS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and type(conv).__name__ == 'Conv2d'):
        return False
    k = conv.kernel_size[0]
    if k < 2 or conv.in_channels < 64:
        return False
    cache = conv.__dict__.setdefault('_img_ok', {})
    key = (tuple(x.shape), ops.X3_EPOCH)
    ok = cache.get(key)
    if ok is None:
        d = ops._desc(x.shape, conv.weight.shape, _one(conv.stride), _one(conv.padding), _one(conv.dilation))
        ok = cache[key] = lib().p3d_fx_conv_img_supported(ctypes.byref(d)) == 7
    return ok


class ConvImagesFn(torch.autograd.Function):
    """conv(x, w) + bias with x, dy (and w) travelling as pre-split images: forward = image pass of x + p3d_fx_conv_fwd_img; backward = image pass of dy +
    p3d_fx_conv_dgrad_img on the launch stream + p3d_fx_conv_wgrad_img (dy image, the x image kept from forward) on the weight-gradient stream."""

    @staticmethod
    def forward(ctx, x, conv, w, bias):
        x = x.contiguous()
        L = lib()
        stride, pad, dil = _one(conv.stride), _one(conv.padding), _one(conv.dilation)
        d = ops._desc(x.shape, w.shape, stride, pad, dil)
        bufs = _owned(conv, ('images', tuple(x.shape)), x.device)
        ctx.lease = _Lease(bufs)
        x_img = ops.act_image(x, out=bufs.tensor('x', 6 * x.numel()))
        y = torch.empty((d.N, d.K, d.Ho, d.Wo), dtype=torch.float32, device=x.device)
        wimg = weight_images(conv)[0] if USE_WEIGHT_IMAGES else None
        ws = ops.workspace(x.device, L.p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), 0))
        with ops._Timed('fwd', d):
            check(L.p3d_fx_conv_fwd_img(ctypes.byref(d), ops._p(x_img), ops._p(w), ops._p(wimg), ops._p(bias), ops._p(y), ops._p(ws), ws.numel(), ops._stream()),
                  'p3d_fx_conv_fwd_img')
        ctx.conv, ctx.cfg, ctx.x_shape = conv, (stride, pad, dil), tuple(x.shape)
        ctx.save_for_backward(x_img)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x_img,) = ctx.saved_tensors
        conv = ctx.conv
        w, bias = conv.weight, conv.bias
        stride, pad, dil = ctx.cfg
        L, st = lib(), ops._stream()
        dy = dy.contiguous()
        d = ops._desc(ctx.x_shape, w.shape, stride, pad, dil)
        lease, ctx.lease = ctx.lease, None
        if lease is None:
            raise P3DError('conv2d_images: backward called a second time on the same graph; run the forward again')
        bufs = lease.bufs
        dy_img = ops.act_image(dy, out=bufs.tensor('dy', 6 * dy.numel()))
        dy_ready = ops._mark_ready() if (ops.WGRAD_STREAM and ctx.needs_input_grad[2]) else None
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dy.device)
            wimgT = weight_images(conv)[1] if USE_WEIGHT_IMAGES else None
            ws = ops.workspace(dy.device, L.p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), 1))
            with ops._Timed('dgrad', d):
                check(L.p3d_fx_conv_dgrad_img(ctypes.byref(d), ops._p(dy_img), ops._p(w), ops._p(wimgT), ops._p(dx), ops._p(ws), ws.numel(), st), 'p3d_fx_conv_dgrad_img')
        if ctx.needs_input_grad[2]:
            sink = ops._grad_sink(w)
            dw = torch.empty_like(w) if sink is None else sink
            d.accumulate = 0 if sink is None else 1
            nbytes = L.p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), 2)
            if ops.WGRAD_STREAM and sink is not None:
                side = ops._side_stream(dy.device)
                ops._queue_join()
                side.wait_event(dy_ready)
                sws = ops._side_workspace(dy.device, nbytes)
                with torch.cuda.stream(side):
                    with ops._Timed('wgrad', d):
                        check(L.p3d_fx_conv_wgrad_img(ctypes.byref(d), ops._p(dy_img), None, ops._p(x_img), ops._p(dw), ops._p(sws), sws.numel(), ops._stream()),
                              'p3d_fx_conv_wgrad_img')
                bufs.mark_side(side)                        # both images are the module's own buffers: their next user orders itself behind this
            else:
                sws = ops.workspace(dy.device, nbytes)
                with ops._Timed('wgrad', d):
                    check(L.p3d_fx_conv_wgrad_img(ctypes.byref(d), ops._p(dy_img), None, ops._p(x_img), ops._p(dw), ops._p(sws), sws.numel(), st), 'p3d_fx_conv_wgrad_img')
            d.accumulate = 0
            if sink is not None:
                dw = None
                ops._grad_done(w)
        if bias is not None and ctx.needs_input_grad[3]:
            sink = ops._grad_sink(bias)
            db = torch.empty(d.K, dtype=torch.float32, device=dy.device) if sink is None else sink
            check(L.p3d_conv2d_bgrad(ops._p(dy), d.N, d.K, d.Ho * d.Wo, ops._p(db), 0 if sink is None else 1, st), 'p3d_conv2d_bgrad')
            if sink is not None:
                db = None
                ops._grad_done(bias)
        return dx, None, dw, db


def conv2d_images(conv, x):
    return ConvImagesFn.apply(x, conv, conv.weight, conv.bias)


# ---- the stem conv1 (7x7, stride 2, Cin = 3 or 1: depthnet.py:138) on the x3 kernels -------------------------------------------------------------------------
def stem_takes_x3(conv, x, masked=False):
    """conv1 on the restated stem kernels?  masked: as a partial convolution (PartialConv stems of partial_depthnet / partial_fusionnet: per-pixel factors)."""
    if not (IMAGE_CONVS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and type(conv).__name__ == ('PartialConv' if masked else 'Conv2d') and conv.bias is None):
        return False
    if (conv.kernel_size[0], _one(conv.stride), _one(conv.padding), _one(conv.dilation)) != (7, 2, 3, 1) or conv.in_channels > 4:
        return False
    n, c, h, w = x.shape
    if masked:
        return MASKED_STEM and bool(lib().p3d_stem_masked_supported(n, c, h, w, conv.out_channels))
    return bool(lib().p3d_stem_supported(n, c, h, w, conv.out_channels))


# P3D_MASKED_STEM=0: the 1-channel partial-convolution stems stay on the fp32-MFMA kernel (the round-3 path; A/B)
MASKED_STEM = os.environ.get('P3D_MASKED_STEM', '1') != '0'


def stem_weight_image(conv):
    """The restated stem's forward weight image, rebuilt when the weight has changed (same key as weight_images)."""
    w = conv.weight
    key = (w._version, ops.WEIGHT_EPOCH, w.data_ptr())
    cached = conv.__dict__.get('_stem_image')
    if cached is not None and cached[0] == key:
        return cached[1]
    k, c = w.shape[0], w.shape[1]
    img = cached[1] if cached is not None else torch.empty(lib().p3d_stem_weight_image_bytes(k), dtype=torch.uint8, device=w.device)
    ws = ops.workspace(w.device, k * 256 * 4)
    check(lib().p3d_stem_weight_image(ops._p(w.detach()), k, c, ops._p(img), ops._p(ws), ws.numel(), ops._stream()), 'p3d_stem_weight_image')
    conv.__dict__['_stem_image'] = (key, img)
    return img


class StemConvFn(torch.autograd.Function):
    """conv1(x): the space-to-depth image of the batch is built once in forward and read again by the weight gradient (no data gradient: x is the input)."""

    @staticmethod
    def forward(ctx, x, conv, w, mask_in=None, mult=None):
        """mask_in [N,1,H,W] / mult [N,1,H/2,W/2]: the partial-convolution stem (partial_conv.py:45-52): y = conv(x * mask_in) * mult; the input factor goes into
        the space-to-depth image, the output factor into the forward epilogue and, in backward, onto dy as the weight gradient fetches it."""
        if x.requires_grad:
            raise P3DError('StemConvFn: a data gradient for the network input is not implemented (the reference never asks for one)')
        x = x.contiguous()
        L = lib()
        n, c, h, wd = x.shape
        k = w.shape[0]
        bufs = _owned(conv, ('stem', tuple(x.shape)), x.device)
        ctx.lease = _Lease(bufs)
        x_img = bufs.tensor('x', L.p3d_stem_image_bytes(n, h, wd))
        if mask_in is not None:
            mask_in, mult = mask_in.contiguous(), mult.contiguous()
            assert tuple(mask_in.shape) == (n, 1, h, wd) and tuple(mult.shape) == (n, 1, h // 2, wd // 2) and mask_in.dtype == mult.dtype == torch.float32
        check(L.p3d_stem_image_masked(ops._p(x), ops._p(mask_in) if mask_in is not None else None, ops._p(x_img), n, c, h, wd, ops._stream()), 'p3d_stem_image')
        y = torch.empty((n, k, h // 2, wd // 2), dtype=torch.float32, device=x.device)
        check(L.p3d_stem_fwd_masked(ops._p(x_img), ops._p(stem_weight_image(conv)), ops._p(y), ops._p(mult) if mult is not None else None, n, c, h, wd, k, ops._stream()),
              'p3d_stem_fwd')
        ctx.conv, ctx.shape = conv, (n, c, h, wd, k)
        ctx.mult = mult
        ctx.save_for_backward(x_img)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x_img,) = ctx.saved_tensors
        n, c, h, wd, k = ctx.shape
        w = ctx.conv.weight
        L = lib()
        dy = dy.contiguous()
        dw = None
        mult = ctx.mult
        mp = ops._p(mult) if mult is not None else None
        lease, ctx.lease = ctx.lease, None
        if lease is None:
            raise P3DError('stem_conv: backward called a second time on the same graph; run the forward again')
        if ctx.needs_input_grad[2]:
            sink = ops._grad_sink(w)
            dw = torch.empty_like(w) if sink is None else sink
            nbytes = L.p3d_stem_workspace_bytes(n, h, wd, k)
            if ops.WGRAD_STREAM and sink is not None:
                dy_ready = ops._mark_ready()
                side = ops._side_stream(dy.device)
                ops._queue_join()
                side.wait_event(dy_ready)
                sws = ops._side_workspace(dy.device, nbytes)
                with torch.cuda.stream(side):
                    check(L.p3d_stem_wgrad_masked(ops._p(dy), mp, ops._p(x_img), ops._p(dw), n, c, h, wd, k, 1, ops._p(sws), sws.numel(), ops._stream()), 'p3d_stem_wgrad')
                dy.record_stream(side)                      # (allocator-owned: the stem BatchNorm's data gradient)
                if mult is not None:
                    mult.record_stream(side)
                lease.bufs.mark_side(side)
            else:
                sws = ops.workspace(dy.device, nbytes)
                check(L.p3d_stem_wgrad_masked(ops._p(dy), mp, ops._p(x_img), ops._p(dw), n, c, h, wd, k, 0 if sink is None else 1, ops._p(sws), sws.numel(), ops._stream()),
                      'p3d_stem_wgrad')
            if sink is not None:
                dw = None
                ops._grad_done(w)
        return None, None, dw, None, None


def stem_conv(conv, x, mask_in=None, mult=None):
    return StemConvFn.apply(x, conv, conv.weight, mask_in, mult)
