"""ORACLE (test infrastructure, not product code): numpy restatement of the reference's
networks and of one training step, composed from oracle/np_ops.py.

Restates (reference file:line):
  * depthnet.BasicBlock / Bottleneck / ResNet            depthnet.py:10-200
  * resnet.ResNet (legacy heads cam_regressor/mat_regressor) resnet.py:122-210
  * fusionnet.Fusion / ResNet                              fusionnet.py:130-240
  * partial_depthnet blocks / ResNet                       partial_depthnet.py:44-75,118-157,213-229
  * Trainer.vanilla_train / fusion_train step body         depth_train.py:376-462, 286-373
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

from . import np_ops as ops


def stage_geometry(stride):
    """depthnet.py:130-136 (identical in fusionnet.py:153-159, partial_depthnet.py:169-175)."""
    lg = np.log2(stride)
    s2 = int(min(max(lg, 2), 3) - 1)
    s3 = int(min(max(lg, 3), 4) - 2)
    s4 = int(min(max(lg, 4), 5) - 3)
    d2 = 3 - s2
    d3 = (3 - s2) * (3 - s3)
    d4 = (3 - s2) * (3 - s3) * (3 - s4)
    return (s2, s3, s4), (d2, d3, d4)


LAYERS = {'resnet18': ('basic', [2, 2, 2, 2]), 'resnet50': ('bottleneck', [3, 4, 6, 3])}


class NpNet:
    """family in {'depthnet', 'resnet', 'fusionnet', 'partial_depthnet'}; sd: key -> ndarray."""

    def __init__(self, sd, family='depthnet', model='resnet18', stride=16, skip_relu=False, early_dist=False,
                 joint_space=False, train=True, acc=np.float64):
        self.sd = sd
        self.family = family
        self.block, self.layers = LAYERS[model]
        self.exp = 1 if self.block == 'basic' else 4
        self.strides, self.dilates = stage_geometry(stride)
        self.skip_relu = skip_relu and family in ('depthnet', 'fusionnet')
        self.early_dist = early_dist and family in ('depthnet', 'fusionnet')
        self.joint_space = joint_space
        self.train = train
        self.acc = acc
        self.grads = {}
        self.new_buffers = {}

    # ---- primitive layers: each returns (y, bwd) with bwd(dy) -> dx -------------------
    def conv(self, name, x, stride=1, pad=0, dil=1, need_dx=True):
        w = self.sd[name + '.weight']
        b = self.sd.get(name + '.bias')
        y = ops.conv2d_fwd(x, w, b, stride, pad, dil, acc=self.acc)

        def bwd(dy):
            self.grads[name + '.weight'] = ops.conv2d_wgrad(dy, x, w.shape, stride, pad, dil, acc=self.acc)
            if b is not None:
                self.grads[name + '.bias'] = ops.conv2d_bgrad(dy, acc=self.acc)
            return ops.conv2d_dgrad(dy, w, x.shape, stride, pad, dil, acc=self.acc) if need_dx else None
        return y, bwd

    def pconv(self, name, x, veil, stride=1, pad=0, dil=1, need_dx=True):
        w = self.sd[name + '.weight']
        y, veil_out, mult = ops.partial_conv_fwd(x, veil, w, None, stride, pad, dil, acc=self.acc)

        def bwd(dy):
            dx, dw = ops.partial_conv_bwd(dy, x, veil, w, mult, stride, pad, dil, acc=self.acc)
            self.grads[name + '.weight'] = dw
            return dx if need_dx else None
        return y, veil_out, bwd

    def bn(self, name, x):
        g, b = self.sd[name + '.weight'], self.sd[name + '.bias']
        rm, rv = self.sd[name + '.running_mean'], self.sd[name + '.running_var']
        if self.train:
            y, mean, invstd, nrm, nrv = ops.bn_train_fwd(x, g, b, rm, rv, acc=self.acc)
            self.new_buffers[name + '.running_mean'] = nrm
            self.new_buffers[name + '.running_var'] = nrv

            def bwd(dy):
                dx, dg, db = ops.bn_train_bwd(dy, x, mean, invstd, g, acc=self.acc)
                self.grads[name + '.weight'], self.grads[name + '.bias'] = dg, db
                return dx
        else:
            y = ops.bn_eval_fwd(x, g, b, rm, rv, acc=self.acc)

            def bwd(dy):
                dx, dg, db = ops.bn_eval_bwd(dy, x, g, b, rm, rv, acc=self.acc)
                self.grads[name + '.weight'], self.grads[name + '.bias'] = dg, db
                return dx
        return y, bwd

    @staticmethod
    def relu(x):
        y = ops.relu_fwd(x)
        return y, (lambda dy: ops.relu_bwd(dy, y))

    @staticmethod
    def maxpool(x):
        y, idx = ops.maxpool3x3s2_fwd(x)
        return y, (lambda dy: ops.maxpool3x3s2_bwd(dy, idx, x.shape))

    # ---- residual blocks ---------------------------------------------------------------
    def res_block(self, p, x, stride, dil, has_ds, skip_relu=False, veil=None):
        """depthnet.py:40-56 / 96-116; partial variant partial_depthnet.py:62-75 / 140-157.
        Returns (out, veil_out, bwd)."""
        bw = []
        partial = veil is not None

        def cv(name, t, v, **kw):
            if partial:
                y, v2, b = self.pconv(name, t, v, **kw)
                return y, v2, b
            y, b = self.conv(name, t, **kw)
            return y, None, b

        if self.block == 'basic':
            spec = [('conv1', 'bn1', dict(stride=stride, pad=dil, dil=dil), True),
                    ('conv2', 'bn2', dict(stride=1, pad=1, dil=1), False)]
        else:
            spec = [('conv1', 'bn1', dict(stride=1, pad=0, dil=1), True),
                    ('conv2', 'bn2', dict(stride=stride, pad=dil, dil=dil), True),
                    ('conv3', 'bn3', dict(stride=1, pad=0, dil=1), False)]
        out = x
        for cname, bname, kw, act in spec:
            out, veil, b1 = cv(p + '.' + cname, out, veil, **kw)
            out, b2 = self.bn(p + '.' + bname, out)
            bw += [b1, b2]
            if act:
                out, b3 = self.relu(out)
                bw.append(b3)
        res_bw = []
        res = x
        if has_ds:
            res, b1 = self.conv(p + '.downsample.0', x, stride=stride)      # dense even in partial blocks
            res, b2 = self.bn(p + '.downsample.1', res)
            res_bw = [b1, b2]
        out = (out + res).astype(np.float32)
        fin = None
        if not skip_relu:
            out, fin = self.relu(out)

        def bwd(dy):
            if fin is not None:
                dy = fin(dy)
            d = dy
            for f in reversed(bw):
                d = f(d)
            dr = dy
            for f in reversed(res_bw):
                dr = f(dr)
            return (d + dr).astype(np.float32)
        return out, veil, bwd

    def stage(self, lname, x, planes, blocks, stride=1, dil=1, inplanes=None, skip_relu=False, veil=None):
        """_make_layer, depthnet.py:163-186: first block carries stride/dilation (+downsample), last block skip_relu."""
        bw = []
        for i in range(blocks):
            first = i == 0
            has_ds = first and (stride != 1 or inplanes != planes * self.exp)
            x, veil, b = self.res_block('%s.%d' % (lname, i), x, stride if first else 1, dil if first else 1, has_ds,
                                        skip_relu=(skip_relu and i == blocks - 1), veil=veil)
            bw.append(b)

        def bwd(dy):
            for f in reversed(bw):
                dy = f(dy)
            return dy
        return x, veil, bwd

    # ---- whole networks ----------------------------------------------------------------
    def forward(self, x, y=None):
        """Returns (z, feat) like depthnet.ResNet.forward (depthnet.py:188-200); for family 'resnet'
        feat is z_mat or None (resnet.py:196-210).  Stores the backward chain in self._bwd."""
        s, d = self.strides, self.dilates
        L = self.layers
        e = self.exp
        chain = []
        partial = self.family == 'partial_depthnet'
        if partial:
            veil = (x != 0).astype(np.float32)                                   # partial_depthnet.py:215
            x, veil, b = self.pconv('conv1', x, veil, stride=2, pad=3, need_dx=False)
        else:
            veil = None
            x, b = self.conv('conv1', x, stride=2, pad=3, need_dx=False)
        chain.append(b)
        x, b = self.bn('bn1', x); chain.append(b)
        x, b = self.relu(x); chain.append(b)
        x, b = self.maxpool(x); chain.append(b)
        if partial:
            veil, _ = ops.maxpool3x3s2_fwd(veil)                                 # partial_depthnet.py:220
        x, veil, b = self.stage('layer1', x, 64, L[0], inplanes=64, veil=veil); chain.append(b)
        x, veil, b = self.stage('layer2', x, 128, L[1], s[0], d[0], inplanes=64 * e, veil=veil); chain.append(b)

        ychain = []
        if self.family in ('fusionnet', 'partial_fusionnet'):                    # fusionnet.py:221-233
            yveil = None
            if self.family == 'partial_fusionnet':       # intended wiring of partial_fusionnet.py:250-275 (dense RGB stem, partial depth stream)
                yveil = (y != 0).astype(np.float32)
                y, yveil, b = self.pconv('conv2', y, yveil, stride=2, pad=3, need_dx=False); ychain.append(b)
                yveil, _ = ops.maxpool3x3s2_fwd(yveil)
            else:
                y, b = self.conv('conv2', y, stride=2, pad=3, need_dx=False); ychain.append(b)
            y, b = self.bn('bn2', y); ychain.append(b)
            y, b = self.relu(y); ychain.append(b)
            y, b = self.maxpool(y); ychain.append(b)
            y, yveil, b = self.stage('layer5', y, 64, L[0], inplanes=64, veil=yveil); ychain.append(b)
            y, yveil, b = self.stage('layer6', y, 128, L[1], s[0], d[0], inplanes=64 * e, veil=yveil); ychain.append(b)
            cx = x.shape[1]
            cat = np.concatenate([x, y], axis=1)
            f, b1 = self.conv('fusion.conv', cat)
            f, b2 = self.bn('fusion.bn', f)
            f, b3 = self.relu(f)
            x = f

            def fuse_bwd(dy):
                dcat = b1(b2(b3(dy)))
                return dcat[:, :cx], dcat[:, cx:]
        m, _, bl3 = self.stage('layer3', x, 256, L[2], s[1], d[1], inplanes=128 * e, skip_relu=self.skip_relu)
        m_in, r3 = (self.relu(m) if self.skip_relu else (m, None))
        n, _, bl4 = self.stage('layer4', m_in, 512, L[3], s[2], d[2], inplanes=256 * e, skip_relu=self.skip_relu)
        n_in, r4 = (self.relu(n) if self.skip_relu else (n, None))
        reg = 'cam_regressor' if self.family == 'resnet' else 'regressor'
        z, breg = self.conv(reg, n_in, pad=1)
        zmat = bmat = None
        if self.family == 'resnet' and self.joint_space:
            zmat, bmat = self.conv('mat_regressor', n_in, pad=1)

        def bwd(dz, dfeat=None):
            dn = breg(dz)
            if bmat is not None and dfeat is not None:
                dn = dn + bmat(dfeat)
                dfeat_n = None
            else:
                dfeat_n = dfeat
            if r4 is not None:
                dn = r4(dn)
            if dfeat_n is not None and not self.early_dist:
                dn = dn + dfeat_n
            dm = bl4(dn)
            if r3 is not None:
                dm = r3(dm)
            if dfeat_n is not None and self.early_dist:
                dm = dm + dfeat_n
            dx = bl3(dm)
            if self.family in ('fusionnet', 'partial_fusionnet'):
                dx, dy = fuse_bwd(dx)
                for f in reversed(ychain):
                    dy = f(dy)
            for f in reversed(chain):
                dx = f(dx)
        self._bwd = bwd
        if self.family == 'resnet':
            return z, zmat
        return z, (m if self.early_dist else n)

    def backward(self, dz, dfeat=None):
        self.grads = {}
        self._bwd(dz, dfeat)
        return self.grads


def train_step(sd, color, depth, true_cam, true_val, family='depthnet', model='resnet18', depth_only=False,
               stride=16, depth_dim=16, num_joints=17, depth_range=1000.0, loss_div=10.0, key_index=16,
               criterion='SmoothL1', lr=1e-5, weight_decay=4e-5, grad_norm=5.0, adam_state=None, step=1, acc=np.float64):
    """One iteration of depth_train.Trainer.vanilla_train / fusion_train (depth_train.py:376-462 / 286-373)
    in fp32 mode.  Returns dict(loss, spec_cam, z, grads, clip_total, clip_coef, new_sd, adam_state)."""
    net = NpNet(sd, family=family, model=model, stride=stride, train=True, acc=acc)
    if family in ('fusionnet', 'partial_fusionnet'):
        z, feat = net.forward(color, depth)
    else:
        z, feat = net.forward(depth if (depth_only or family == 'partial_depthnet') else color)
    side_out = z.shape[-1]
    relat = ops.softargmax3d_fwd(z, depth_dim, num_joints, side_out, side_out, depth_range, acc=acc)
    loss, spec, drelat = ops.pose_loss_fwd_bwd(relat, true_cam, true_val, key_index, loss_div, criterion, acc=acc)
    dz = ops.softargmax3d_bwd(drelat, z, depth_dim, num_joints, side_out, side_out, depth_range, acc=acc)
    grads = net.backward(dz)
    names = [k for k in sd if k in grads]
    total, coef = ops.clip_grad_norm([grads[k] for k in names], grad_norm, acc=acc)
    if adam_state is None:
        adam_state = {k: (np.zeros_like(sd[k]), np.zeros_like(sd[k])) for k in names}
    new_sd = dict(sd)
    new_state = {}
    for k in names:
        m, v = adam_state[k]
        p2, m2, v2 = ops.adam_step(sd[k], grads[k], m, v, step, lr, weight_decay=weight_decay, grad_scale=coef, acc=acc)
        new_sd[k] = p2
        new_state[k] = (m2, v2)
    new_sd.update(net.new_buffers)
    for k in sd:
        if k.endswith('num_batches_tracked'):
            new_sd[k] = sd[k] + 1
    return dict(loss=loss, spec_cam=spec, z=z, feat=feat, grads=grads, clip_total=total, clip_coef=coef,
                new_sd=new_sd, adam_state=new_state)
