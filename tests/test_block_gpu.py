"""The fused residual-block executor (p3d_block_fwd / p3d_block_bwd: one C call per block and direction, BatchNorm inside the convolution kernels)
against the per-layer path (one autograd node per conv / BN, stand-alone BatchNorm passes) on the same module and data, and against float64 PyTorch:
block output, input gradient, every parameter gradient and the BatchNorm running statistics (depthnet.py:40-56,96-116)."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get('P3D_X3', '1') == '0', reason='the block executor needs the x3 kernels (P3D_X3=0 runs every layer on the fp32-MFMA path)')]

#        kind          inplanes planes stride dil  N  H   downsample
CASES = [('bottleneck', 512, 128, 1, 1, 4, 16, False),       # identity shortcut (layerN.1+)
         ('bottleneck', 128, 128, 1, 1, 3, 32, True),        # a stride-1 downsample (the layer1.0 pattern at 128 channels)
         ('bottleneck', 256, 128, 2, 1, 4, 32, True),        # layer2.0 / layer3.0: stride 2 on the 3x3, strided 1x1 downsample
         ('bottleneck', 256, 128, 1, 2, 2, 16, True),        # layer4.0 at -stride 16: dilation 2, stride-1 downsample
         ('basic', 128, 128, 1, 1, 4, 16, False),
         ('basic', 128, 256, 2, 1, 4, 32, True),
         ('basic', 128, 256, 1, 4, 2, 32, True),             # -stride 8 geometry: dilation 4
         ('bottleneck', 256, 64, 1, 1, 4, 32, False),        # layer1.1+: 64-channel convs in half-dead 128-row tiles, their weight gradients on the fp32-MFMA kernel
         ('bottleneck', 64, 64, 1, 1, 2, 32, True),          # layer1.0
         ('basic', 64, 64, 1, 1, 4, 32, False),              # ResNet-18 layer1
         ('bottleneck', 1024, 256, 1, 1, 64, 16, False)]      # a BASELINE-size layer3 block at batch 64 (split-K forward / dgrad of the 3x3)


def build(pkg, kind, inplanes, planes, stride, dil, with_ds, seed):
    tr = pkg._trunk
    block_cls = tr.Bottleneck if kind == 'bottleneck' else tr.BasicBlock
    ds = None
    if with_ds:
        ds = tr.Sequential(pkg.nn.Conv2d(inplanes, planes * block_cls.expansion, kernel_size=1, stride=stride, bias=False),
                           pkg.nn.BatchNorm2d(planes * block_cls.expansion))
    torch.manual_seed(seed)
    block = block_cls(inplanes, planes, stride, dil, ds)
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    return block.cuda().train()


def run(pkg, block, x0, dy, fused):
    before = pkg._trunk.FUSED_BLOCKS
    pkg._trunk.FUSED_BLOCKS = fused
    try:
        state = {k: v.clone() for k, v in block.state_dict().items()}
        block.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = block(x)
        y.backward(dy)
        torch.cuda.synchronize()
        res = dict(y=y.detach().clone(), dx=x.grad.clone(), grads={n: p.grad.clone() for n, p in block.named_parameters()},
                   buffers={k: v.clone() for k, v in block.state_dict().items() if 'running' in k or 'tracked' in k})
        block.load_state_dict(state)
        return res
    finally:
        pkg._trunk.FUSED_BLOCKS = before


def reference(block, x0, dy, with_ds):
    """float64 PyTorch forward / backward of the same module (torch.nn semantics are the reference's: depthnet.py builds exactly these layers).
    Also returns the smallest |pre-ReLU activation|: two correct fp32 implementations may put an activation within rounding of zero on different
    sides, which moves single gradient entries by O(1); the tight bounds below are only meaningful when no activation sits that close."""
    import copy
    ref = copy.deepcopy(block).double()
    for m in ref.modules():                                                             # plain torch forward, not the HIP one
        if isinstance(m, torch.nn.Conv2d):
            m.forward = lambda inp, _m=m: torch.nn.functional.conv2d(inp, _m.weight, None, _m.stride, _m.padding, _m.dilation)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.forward = lambda inp, _m=m: torch.nn.functional.batch_norm(inp, _m.running_mean, _m.running_var, _m.weight, _m.bias, True, 0.1, _m.eps)
    xr = x0.double().requires_grad_(True)
    out, closest = xr, float('inf')
    chain = block._chain
    bn_io = []                                                                          # (BatchNorm name, its input, its output) for the reduction scales below
    for i, (cname, bname) in enumerate(chain):
        pre = getattr(ref, cname)(out)
        out = getattr(ref, bname)(pre)
        out.retain_grad()
        bn_io.append((bname, pre, out))
        if i < len(chain) - 1:
            closest = min(closest, float(out.detach().abs().min()))
            out = out.relu()
    if with_ds:
        pre = ref.downsample[0](xr)
        res = ref.downsample[1](pre)
        res.retain_grad()
        bn_io.append(('downsample.1', pre, res))
    else:
        res = xr
    closest = min(closest, float((out + res).detach().abs().min()))
    yr = (out + res).relu()
    yr.backward(dy.double())
    # what each BatchNorm parameter gradient sums over: dbeta = sum g, dgamma = sum g * xhat; sum |terms| is the scale rounding errors of the sum live on
    ref._grad_scales = {}
    with torch.no_grad():
        for bname, pre, o in bn_io:
            g = o.grad
            mean, var = pre.mean((0, 2, 3), keepdim=True), pre.var((0, 2, 3), unbiased=False, keepdim=True)
            xhat = (pre - mean) / torch.sqrt(var + 1e-5)
            ref._grad_scales[bname + '.bias'] = g.abs().sum((0, 2, 3))
            ref._grad_scales[bname + '.weight'] = (g * xhat).abs().sum((0, 2, 3))
    return ref, xr, yr.detach(), closest


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def rel2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def make_case(pkg, kind, inplanes, planes, stride, dil, n, h, with_ds, want_clean):
    """Block + data; with want_clean the seed is advanced until no pre-ReLU activation lies within 2e-6 of zero (see `reference`)."""
    for seed in range(3, 13):
        block = build(pkg, kind, inplanes, planes, stride, dil, with_ds, seed=inplanes + planes + seed)
        gen = torch.Generator(device='cuda').manual_seed(seed)
        x0 = torch.randn(n, inplanes, h, h, device='cuda', generator=gen).relu_()           # a block input is a ReLU output
        with torch.no_grad():
            shape = block(x0).shape
        dy = torch.randn(shape, device='cuda', generator=gen)
        ref = reference(block, x0, dy, with_ds)
        if not want_clean or ref[3] > 2e-6:
            return block, x0, dy, ref
    raise AssertionError('no seed without a near-zero activation')


@pytest.mark.parametrize('case', CASES, ids=['%s_c%d_p%d_s%d_d%d_n%d_h%d%s' % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], '_ds' if c[7] else '') for c in CASES])
def test_fused_block_matches_per_layer_path_and_float64(case, pkg):
    kind, inplanes, planes, stride, dil, n, h, with_ds = case
    small = n * h * h <= 8192
    block, x0, dy, (ref, xr, yr, closest) = make_case(pkg, kind, inplanes, planes, stride, dil, n, h, with_ds, want_clean=small)
    assert pkg.ops_block.usable(block, x0)
    plain = run(pkg, block, x0, dy, fused=False)
    fused = run(pkg, block, x0, dy, fused=True)
    assert not torch.equal(plain['y'], fused['y'])                                      # the other path really ran
    # no activation near zero: element-wise bounds; the batch-64 case (50 M activations behind ReLUs, some within 1e-8 of zero): norm-wise
    err, tol = (rel, 2e-5) if small else (rel2, 2e-3)
    for name, got in (('fused', fused), ('per-layer', plain)):
        assert rel(got['y'], yr) < 2e-5, (name, 'y')
        assert err(got['dx'], xr.grad) < tol, (name, 'dx', err(got['dx'], xr.grad), closest)
        for pname, p in ref.named_parameters():
            assert err(got['grads'][pname], p.grad) < 4 * tol, (name, pname, err(got['grads'][pname], p.grad), closest)
        for k, v in ref.state_dict().items():
            if 'running' in k:
                assert rel(got['buffers'][k], v) < 1e-5, (name, k)
            if 'tracked' in k:
                assert int(got['buffers'][k]) == int(v) + 1          # (the float64 copy runs functional batch_norm, which does not count)
    assert rel(fused['y'], plain['y']) < 1e-5 and err(fused['dx'], plain['dx']) < tol


# The eight block geometries of ResNet-50 at the bench's size (256 x 256 crops, -stride 16, batch 64: depthnet.py:59-116,130-146), i.e. the kernel
# instances, split plans, partial-sum row counts and half-dead tiles that carry the bench step, each through p3d_block_fwd / p3d_block_bwd:
#            name       inplanes planes stride dil  H  downsample
R50_BLOCKS = [('layer1.0', 64, 64, 1, 1, 64, True), ('layer1.1', 256, 64, 1, 1, 64, False),
              ('layer2.0', 256, 128, 2, 1, 64, True), ('layer2.1', 512, 128, 1, 1, 32, False),
              ('layer3.0', 512, 256, 2, 1, 32, True), ('layer3.1', 1024, 256, 1, 1, 16, False),
              ('layer4.0', 1024, 512, 1, 2, 16, True), ('layer4.1', 2048, 512, 1, 1, 16, False)]


def _decide_relus(block, seed):
    """BatchNorm parameters that put every pre-ReLU value far from zero: beta = +-8 per channel (random sign) and gamma in [0.5, 1] on the inner
    layers, beta = +-14 on the closing layer (its ReLU sees bn(c) + shortcut, |shortcut| < 6), a small-offset downsample BatchNorm.  A ReLU
    then is on or off for a whole channel, whatever the arithmetic, and gradients can be compared element by element at batch 64 -- with
    ordinary parameters ~1e-6 of the 10^7..10^8 activations of these blocks lie within fp32 rounding of zero, and a ReLU that two correct
    implementations put on different sides moves single gradient entries by percents (the norm-wise half of the test covers that regime)."""
    gen = torch.Generator().manual_seed(seed)
    last = len(block._chain) - 1
    with torch.no_grad():
        for i, (_, bname) in enumerate(block._chain):
            bn = getattr(block, bname)
            sign = torch.where(torch.rand(bn.bias.numel(), generator=gen) < 0.5, -1.0, 1.0)
            bn.weight.copy_((0.5 + 0.5 * torch.rand(bn.weight.numel(), generator=gen)).to(bn.weight.device))
            bn.bias.copy_((sign * (14.0 if i == last else 8.0)).to(bn.bias.device))
        if block.downsample is not None:
            bn = block.downsample[1]
            bn.weight.copy_((0.5 + 0.5 * torch.rand(bn.weight.numel(), generator=gen)).to(bn.weight.device))
            bn.bias.copy_((0.3 * torch.randn(bn.bias.numel(), generator=gen)).to(bn.bias.device))


def _sampled_max_err(got, want, rng, count=4096):
    """max |got - want| over `count` random entries (all of them if the tensor is smaller), relative to max |want| over the whole tensor."""
    g, w = got.reshape(-1), want.reshape(-1)
    if g.numel() > count:
        idx = torch.from_numpy(rng.choice(g.numel(), count, replace=False)).to(g.device)
        g, w_s = g[idx], w[idx]
    else:
        w_s = w
    return ((g.double() - w_s.double()).abs().max() / w.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('geom', R50_BLOCKS, ids=[g[0] for g in R50_BLOCKS])
def test_resnet50_block_geometries_at_batch_64(geom, pkg):
    name, inplanes, planes, stride, dil, h, with_ds = geom
    n = 64
    rng = np.random.default_rng(inplanes + planes + h)
    # (a) decided ReLUs: element-wise, >= 4096 sampled entries of y, dx and every weight gradient, all of dgamma / dbeta, against float64
    block = build(pkg, 'bottleneck', inplanes, planes, stride, dil, with_ds, seed=11 + inplanes)
    _decide_relus(block, seed=inplanes + h)
    gen = torch.Generator(device='cuda').manual_seed(5 + planes)
    x0 = torch.randn(n, inplanes, h, h, device='cuda', generator=gen).relu_()
    with torch.no_grad():
        shape = block(x0).shape
    dy = torch.randn(shape, device='cuda', generator=gen)
    assert pkg.ops_block.usable(block, x0)
    ref, xr, yr, closest = reference(block, x0, dy, with_ds)
    assert closest > 1e-3, closest                       # no ReLU decision depends on rounding
    got = run(pkg, block, x0, dy, fused=True)
    assert _sampled_max_err(got['y'], yr, rng) < 2e-5
    assert _sampled_max_err(got['dx'], xr.grad, rng) < 2e-5, ('dx', _sampled_max_err(got['dx'], xr.grad, rng))
    for pname, p in ref.named_parameters():
        if p.dim() == 4:
            e = _sampled_max_err(got['grads'][pname], p.grad, rng)
            assert e < 5e-5, (pname, e)
        else:
            # dgamma / dbeta: every channel, against a bound that scales with the channel's own reduction -- sums of N * H * W = 16 k .. 262 k terms that
            # largely cancel (behind a BatchNorm the incoming gradient sums to ~0 per channel): 2e-6 of the sum of the terms' magnitudes
            err = (got['grads'][pname].double() - p.grad).abs()
            bound = 2e-6 * ref._grad_scales[pname] + 1e-30
            assert (err <= bound).all(), (pname, float((err / bound).max()))
    for k, v in ref.state_dict().items():
        if 'running' in k:
            assert rel(got['buffers'][k], v) < 1e-5, k
    del ref, xr, yr, got
    torch.cuda.empty_cache()
    # (b) ordinary BatchNorm parameters (ReLUs switch element by element; some pre-activations lie within rounding of zero): norm-wise
    block = build(pkg, 'bottleneck', inplanes, planes, stride, dil, with_ds, seed=13 + inplanes)
    ref, xr, yr, closest = reference(block, x0, dy, with_ds)
    got = run(pkg, block, x0, dy, fused=True)
    assert rel(got['y'], yr) < 2e-5
    assert rel2(got['dx'], xr.grad) < 2e-3, ('dx', rel2(got['dx'], xr.grad), closest)
    for pname, p in ref.named_parameters():
        assert rel2(got['grads'][pname], p.grad) < 8e-3, (pname, rel2(got['grads'][pname], p.grad), closest)


def test_blocks_outside_the_executor_stay_on_the_per_layer_path(pkg):
    """Channel counts the x3 kernels do not take (a reduction that is not a multiple of 16) keep the block on the per-layer path."""
    block = build(pkg, 'bottleneck', 160, 40, 1, 1, False, seed=1)
    x = torch.randn(2, 160, 16, 16, device='cuda')
    assert not pkg.ops_block.usable(block, x)
    assert torch.isfinite(block(x)).all()


def test_fused_block_writes_gradients_into_the_flat_buffer(pkg):
    """With FlatAdam the executor accumulates straight into the flat gradient buffer, weight gradients on the second stream."""
    block, x0, dy, (ref, xr, yr, closest) = make_case(pkg, 'bottleneck', 512, 128, 1, 1, 4, 16, False, want_clean=True)
    opt = pkg.optim.FlatAdam(list(block.named_parameters()), lr=1e-3)
    res = []
    default = pkg._trunk.FUSED_BLOCKS
    for fused in (False, True):
        pkg._trunk.FUSED_BLOCKS = fused
        try:
            for rep in range(2):                       # twice: the second pass must not see anything left over from the first
                opt.zero_grad()
                x = x0.clone().requires_grad_(True)
                block(x).backward(dy)
                pkg.ops.join_side_stream()
                torch.cuda.synchronize()
            res.append((opt.flat_g.clone(), x.grad.clone()))
        finally:
            pkg._trunk.FUSED_BLOCKS = default
    want = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in ref.parameters()])
    for flat, dx in res:
        assert rel(flat, want) < 1e-4 and rel(dx, xr.grad) < 2e-5


def test_block_buffers_two_executions_in_flight_and_deep_copies(pkg):
    """The executor's activations live in buffer sets owned by the block's plan (ops_block._Buffers), not in per-call allocations.  A block that runs twice
    before either backward (semi_train: labelled + unlabelled batch through one model) must keep both executions apart; a third concurrent execution gets a
    temporary set; a deep copy of a block that has run starts with an empty cache and gives the same results."""
    import copy
    block = build(pkg, 'bottleneck', 256, 64, 1, 1, False, seed=5)
    gen = torch.Generator(device='cuda').manual_seed(11)
    xs = [torch.randn(4, 256, 16, 16, device='cuda', generator=gen).relu_() for _ in range(3)]
    dys = [torch.randn(4, 256, 16, 16, device='cuda', generator=gen) for _ in range(3)]
    state = {k: v.clone() for k, v in block.state_dict().items()}

    def one(i):
        block.load_state_dict(state)
        block.zero_grad(set_to_none=True)
        x = xs[i].clone().requires_grad_(True)
        y = block(x)
        y.backward(dys[i])
        return y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in block.named_parameters()}

    alone = [one(i) for i in range(3)]
    # three forwards, then the three backwards in another order
    block.load_state_dict(state)
    block.zero_grad(set_to_none=True)
    inputs = [xs[i].clone().requires_grad_(True) for i in range(3)]
    outs = [block(x) for x in inputs]
    plans = block.__dict__['_blk_plans']
    assert len(plans) == 1 and len(next(iter(plans.values())).sets) == 2          # two cached sets; the third execution holds a temporary one
    assert all(b.held for b in next(iter(plans.values())).sets)
    grads = {}
    for i in (1, 2, 0):
        block.zero_grad(set_to_none=True)
        outs[i].backward(dys[i])
        grads[i] = {n: p.grad.clone() for n, p in block.named_parameters()}
    torch.cuda.synchronize()
    assert not any(b.held for b in next(iter(plans.values())).sets)
    for i in range(3):
        assert torch.equal(outs[i].detach(), alone[i][0]), i
        assert torch.equal(inputs[i].grad, alone[i][1]), i
        for n, g in grads[i].items():
            assert torch.equal(g, alone[i][2][n]), (i, n)
    # a deep copy carries no buffers over and computes the same
    twin = copy.deepcopy(block)
    assert len(twin.__dict__['_blk_plans']) == 0
    twin.load_state_dict(state)
    x = xs[0].clone().requires_grad_(True)
    y = twin(x)
    y.backward(dys[0])
    assert torch.equal(y.detach(), alone[0][0]) and torch.equal(x.grad, alone[0][1])


@pytest.mark.parametrize('geom', [('bottleneck', 256, 128, 2, True, (4, 256, 32, 32), (4, 512, 16, 16)),      # downsample branch: g is never written at all
                                  ('bottleneck', 512, 128, 1, False, (4, 512, 16, 16), (4, 512, 16, 16)),     # identity: dx = dgrad + dout * mask in the dgrad epilogue
                                  ('basic', 128, 128, 1, False, (4, 128, 16, 16), (4, 128, 16, 16))], ids=['downsample', 'identity', 'basic_identity'])
def test_out_mask_bytes_equal_reading_the_output(pkg, geom):
    """p3d_block_io.out_mask is optional: with the mask bytes forward leaves (the masked upstream gradient then never exists as a tensor), and with NULL (backward
    reads `out` and writes g), every result is bit-identical."""
    ob = pkg.ops_block
    kind, inpl, planes, stride, with_ds, xs, ys = geom
    gen = torch.Generator(device='cuda').manual_seed(3)
    x0 = torch.randn(*xs, device='cuda', generator=gen).relu_()
    dy = torch.randn(*ys, device='cuda', generator=gen)
    res = {}
    before = ob.USE_OUT_MASK
    try:
        for use in (True, False):
            ob.USE_OUT_MASK = use
            torch.manual_seed(77)                                                  # (build() seeds after it has made the downsample pair)
            block = build(pkg, kind, inpl, planes, stride, 1, with_ds, seed=9)     # (a fresh module: buffer sets are created per plan)
            res[use] = run(pkg, block, x0, dy, fused=True)
            bufs = next(iter(block.__dict__['_blk_plans'].values())).sets[0]
            assert (bufs.mask is not None) == use
    finally:
        ob.USE_OUT_MASK = before
    assert torch.equal(res[True]['y'], res[False]['y']) and torch.equal(res[True]['dx'], res[False]['dx'])
    for n, g in res[True]['grads'].items():
        assert torch.equal(g, res[False]['grads'][n]), n


def test_plans_of_old_input_shapes_are_dropped(pkg):
    """A plan owns device buffers, so a block keeps the plans of its PLAN_LIMIT most recently used input shapes only (batch sizes that come and go must not
    accumulate memory); a plan whose buffers are held by a live graph is never the one dropped."""
    ob = pkg.ops_block
    block = build(pkg, 'basic', 64, 64, 1, 1, False, seed=2)
    held = block(torch.randn(7, 64, 16, 16, device='cuda').relu_().requires_grad_(True))          # its graph stays alive: the plan for batch 7 is in use
    for n in (1, 2, 3, 4, 5):
        x = torch.randn(n, 64, 16, 16, device='cuda').relu_().requires_grad_(True)
        block(x).sum().backward()
    plans = block.__dict__['_blk_plans']
    shapes = [k[0][0] for k in plans]
    assert len(plans) <= ob.PLAN_LIMIT + 1 and 7 in shapes and shapes[-1] == 5, shapes
    held.sum().backward()                                                                          # still intact
    torch.cuda.synchronize()


def test_pair_image_pass_equals_two_passes(pkg):
    """A block with a downsample branch writes its two opening gradient images (closing BatchNorm, downsample BatchNorm) in one pass that reads the upstream gradient
    once (fx_act_image_pair); p3d_fx_tune(3, 0) switches back to two passes: every result is bit-identical."""
    L = pkg._lib.lib()
    gen = torch.Generator(device='cuda').manual_seed(5)
    x0 = torch.randn(4, 128, 32, 32, device='cuda', generator=gen).relu_()
    dy = torch.randn(4, 512, 32, 32, device='cuda', generator=gen)
    res = {}
    try:
        for pair in (1, 0):
            L.p3d_fx_tune(3, pair)
            torch.manual_seed(78)
            block = build(pkg, 'bottleneck', 128, 128, 1, 1, True, seed=4)
            res[pair] = run(pkg, block, x0, dy, fused=True)
    finally:
        L.p3d_fx_tune(3, 1)
    assert torch.equal(res[1]['y'], res[0]['y']) and torch.equal(res[1]['dx'], res[0]['dx'])
    for n, g in res[1]['grads'].items():
        assert torch.equal(g, res[0]['grads'][n]), n


@pytest.mark.parametrize('consumer', ['identity', 'downsample_s1', 'downsample_s2'])
def test_opening_sums_from_the_consumer_blocks_epilogue(pkg, consumer):
    """A chain of two blocks: the second block's backward pass reduces the channel sums the FIRST block's backward pass opens with (sum g, sum g (c - mean) of its
    closing and downsample BatchNorm over g = dout [out > 0]) in the epilogue of the data gradient that writes dout, so the first block skips its pass over dout
    (p3d_block_io.tail_* / open_sums; depthnet.py:101-116 and its autograd).  Same gradients as with the opening pass (P3D_TAIL_SUMS=0) to fp32 summation-order
    accuracy; a strided downsample consumer cannot do it (its last dx writer is not dense) and a second consumer of the tensor invalidates the sums."""
    ob = pkg.ops_block
    torch.manual_seed(5)
    first = build(pkg, 'bottleneck', 128, 64, 1, 1, True, 3)                       # output 256 channels, with a downsample branch (three sums)
    if consumer == 'identity':
        second = build(pkg, 'bottleneck', 256, 64, 1, 1, False, 4)
    elif consumer == 'downsample_s1':
        second = build(pkg, 'bottleneck', 256, 128, 1, 2, True, 4)
    else:
        second = build(pkg, 'bottleneck', 256, 128, 2, 1, True, 4)
    gen = torch.Generator(device='cuda').manual_seed(9)
    x0 = torch.randn(6, 128, 32, 32, device='cuda', generator=gen)

    def step(tail, extra_consumer=False):
        keep = ob.USE_TAIL_SUMS
        ob.USE_TAIL_SUMS = tail
        try:
            for blk in (first, second):
                blk.zero_grad(set_to_none=True)
            state = [{k: v.clone() for k, v in blk.state_dict().items()} for blk in (first, second)]
            x = x0.clone().requires_grad_(True)
            mid = first(x)
            y = second(mid)
            loss = (y * dy).sum() + ((mid * 0.5).sum() if extra_consumer else 0.0)
            loss.backward()
            pkg.ops.join_side_stream()
            torch.cuda.synchronize()
            res = {'dx': x.grad.clone()}
            for tag, blk in (('a.', first), ('b.', second)):
                res.update({tag + n: p.grad.clone() for n, p in blk.named_parameters()})
            for blk, st in zip((first, second), state):
                blk.load_state_dict(st)
            return res
        finally:
            ob.USE_TAIL_SUMS = keep

    with torch.no_grad():
        dy = torch.randn(second(first(x0)).shape, device='cuda', generator=gen)
    base = step(False)
    before = dict(ob.TAIL_STATS)
    fused = step(True)
    did = {k: ob.TAIL_STATS[k] - before[k] for k in before}
    if consumer == 'downsample_s2':
        assert did == {'reduced': 0, 'opening_passes_skipped': 0}
    else:
        assert did == {'reduced': 1, 'opening_passes_skipped': 1}, did
    for k in base:
        scale = base[k].abs().max().item()
        assert (fused[k] - base[k]).abs().max().item() <= 2e-5 * scale, (k, (fused[k] - base[k]).abs().max().item(), scale)
    if consumer != 'downsample_s2':
        # a second consumer of the first block's output: autograd adds its gradient to the second block's dx, the sums no longer describe what arrives
        base2 = step(False, extra_consumer=True)
        before = dict(ob.TAIL_STATS)
        fused2 = step(True, extra_consumer=True)
        assert ob.TAIL_STATS['opening_passes_skipped'] == before['opening_passes_skipped']
        for k in base2:
            assert (fused2[k] - base2[k]).abs().max().item() <= 2e-5 * base2[k].abs().max().item(), k


#                kind          inplanes planes stride dil  N  H   downsample
MASKED_CASES = [('bottleneck', 64, 64, 1, 1, 3, 32, True),         # partial_depthnet layer1.0: 64-channel partial convs (the 64-row tiles), dense stride-1 downsample
                ('bottleneck', 256, 64, 1, 1, 4, 32, False),       # layer1.1+
                ('bottleneck', 256, 128, 2, 1, 4, 32, True),       # layer2.0: a strided 3x3 partial conv (strided data gradient with the input-mask factor)
                ('bottleneck', 512, 128, 1, 1, 4, 16, False),      # layer2.1+
                ('basic', 64, 64, 1, 1, 4, 32, False),             # ResNet-18 layer1: conv 1 is a 3x3 partial conv on the fp32 block input
                ('basic', 64, 128, 2, 1, 4, 32, True)]             # ResNet-18 layer2.0


@pytest.mark.parametrize('case', MASKED_CASES, ids=['%s_c%d_p%d_s%d_n%d_h%d%s' % (c[0], c[1], c[2], c[3], c[5], c[6], '_ds' if c[7] else '') for c in MASKED_CASES])
def test_masked_block_on_the_executor_matches_the_per_layer_path(case, pkg):
    """A residual block of partial convolutions (partial_depthnet.py:62-75,140-157; partial_conv.py:32-57) as ONE executor call per direction -- mask_in multiplied
    into the activation images (or into the in-kernel split of the fp32 block input), mult into the conv epilogues in front of the BatchNorm statistics and into
    the gradient images -- against the per-layer path (one autograd node per PartialConv / BatchNorm, itself pinned to the reference's goldens): output, mask_out,
    input gradient, every parameter gradient, running statistics.  The mask has holes, fully masked windows included."""
    kind, inplanes, planes, stride, dil, n, h, with_ds = case
    tr = pkg._trunk
    block_cls = tr.Bottleneck if kind == 'bottleneck' else tr.BasicBlock
    ds = None
    if with_ds:
        ds = tr.Sequential(pkg.nn.Conv2d(inplanes, planes * block_cls.expansion, kernel_size=1, stride=stride, bias=False), pkg.nn.BatchNorm2d(planes * block_cls.expansion))
    torch.manual_seed(11)
    block = block_cls(inplanes, planes, stride, dil, ds, partial=True)
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    block = block.cuda().train()
    _decide_relus(block, 31)        # every ReLU is on or off for a whole channel: one flipped activation would move a whole channel's gradient through the BatchNorm sums
    gen = torch.Generator(device='cuda').manual_seed(21)
    x0 = torch.randn(n, inplanes, h, h, device='cuda', generator=gen)
    veil = (torch.rand(n, 1, h, h, device='cuda', generator=gen) > 0.3).float()
    veil[0, 0, :9, :11] = 0.0
    assert pkg.ops_block.usable(block, x0, veil)

    def go(fused):
        before = pkg._trunk.FUSED_BLOCKS
        pkg._trunk.FUSED_BLOCKS = fused
        try:
            state = {k: v.clone() for k, v in block.state_dict().items()}
            block.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            y, vo = block((x, veil))
            if not hasattr(go, 'dy'):
                go.dy = torch.randn(y.shape, device='cuda', generator=gen)
            y.backward(go.dy)
            pkg.ops.join_side_stream()
            torch.cuda.synchronize()
            res = dict(y=y.detach().clone(), vo=vo.clone(), dx=x.grad.clone(), node=type(y.grad_fn).__name__, grads={k: p.grad.clone() for k, p in block.named_parameters()},
                       buffers={k: v.clone() for k, v in block.state_dict().items() if 'running' in k})
            block.load_state_dict(state)
            return res
        finally:
            pkg._trunk.FUSED_BLOCKS = before

    plain = go(False)
    fused = go(True)
    assert fused['node'].startswith('ResidualBlockFn') and not plain['node'].startswith('ResidualBlockFn')
    assert torch.equal(fused['vo'], plain['vo'])
    assert rel(fused['y'], plain['y']) < 2e-5
    assert rel(fused['dx'], plain['dx']) < 5e-5
    for k in plain['grads']:
        assert rel(fused['grads'][k], plain['grads'][k]) < 1e-4, (k, rel(fused['grads'][k], plain['grads'][k]))
    for k in plain['buffers']:
        assert rel(fused['buffers'][k], plain['buffers'][k]) < 1e-5, k
