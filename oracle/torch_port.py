"""ORACLE / CPU BASELINE (test infrastructure, not product code): the reference's training step restated with
plain PyTorch CPU operators (torch.nn.functional + autograd + a hand-written Adam), fp32.

The reference *is* eager PyTorch, so this is the closest thing to "the reference's CPU path" that can
travel to the GPU box (the reference itself cannot).  It is pinned to the reference by the same golden
vectors as the numpy oracle (tests/test_oracle_golden.py) and is what bench.py times as `cpu_baseline`
(kind "port") on the box's host cores.  Only tests/, smoke() and bench.py's cpu_baseline leg may import it.

Restates: depthnet.py:40-56,96-116,188-200 (blocks, forward), fusionnet.py:130-140,221-240,
partial_conv.py:32-57, partial_depthnet.py:213-229, utils.py:154-194, depth_train.py:376-462 (step).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .np_net import LAYERS, stage_geometry


class TorchPort:

    def __init__(self, sd, family='depthnet', model='resnet18', stride=16, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.family = family
        self.block, self.layers = LAYERS[model]
        self.exp = 1 if self.block == 'basic' else 4
        self.strides, self.dilates = stage_geometry(stride)
        self.p = {}
        self.buf = {}
        for k, v in sd.items():
            t = torch.from_numpy(np.array(v))
            if t.dtype == torch.float32 and not k.endswith(('running_mean', 'running_var')):
                self.p[k] = t.clone().requires_grad_(True)
            else:
                self.buf[k] = t.clone()
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.step = 0

    # ---- layers ----
    def conv(self, name, x, stride=1, pad=0, dil=1):
        return F.conv2d(x, self.p[name + '.weight'], self.p.get(name + '.bias'), stride, pad, dil)

    def pconv(self, name, x, veil, stride=1, pad=0, dil=1):
        w = self.p[name + '.weight']
        k = w.shape[-1]
        with torch.no_grad():
            cnt = F.conv2d(veil, torch.ones(1, 1, k, k), None, stride, pad, dil)
            mult = (k * k) / (cnt + 1e-6)
            veil_out = torch.clamp(cnt, 0, 1)
            mult = mult * veil_out
        return F.conv2d(x * veil, w, None, stride, pad, dil) * mult, veil_out

    def bn(self, name, x):
        return F.batch_norm(x, self.buf[name + '.running_mean'], self.buf[name + '.running_var'], self.p[name + '.weight'],
                            self.p[name + '.bias'], True, 0.1, 1e-5)

    def block_fwd(self, p, x, stride, dil, has_ds, veil=None):
        partial = veil is not None
        if self.block == 'basic':
            spec = [('conv1', 'bn1', (stride, dil, dil), True), ('conv2', 'bn2', (1, 1, 1), False)]
        else:
            spec = [('conv1', 'bn1', (1, 0, 1), True), ('conv2', 'bn2', (stride, dil, dil), True), ('conv3', 'bn3', (1, 0, 1), False)]
        out = x
        for cname, bname, (s, pd, dl), act in spec:
            if partial:
                out, veil = self.pconv(p + '.' + cname, out, veil, s, pd, dl)
            else:
                out = self.conv(p + '.' + cname, out, s, pd, dl)
            out = self.bn(p + '.' + bname, out)
            if act:
                out = F.relu(out)
        res = x
        if has_ds:
            res = self.bn(p + '.downsample.1', self.conv(p + '.downsample.0', x, stride))
        return F.relu(out + res), veil

    def stage(self, lname, x, planes, blocks, stride=1, dil=1, inplanes=None, veil=None):
        for i in range(blocks):
            first = i == 0
            has_ds = first and (stride != 1 or inplanes != planes * self.exp)
            x, veil = self.block_fwd('%s.%d' % (lname, i), x, stride if first else 1, dil if first else 1, has_ds, veil)
        return x, veil

    def forward(self, x, y=None):
        s, d, L, e = self.strides, self.dilates, self.layers, self.exp
        veil = None
        if self.family == 'partial_depthnet':
            veil = (x != 0).float()
            x, veil = self.pconv('conv1', x, veil, 2, 3)
        else:
            x = self.conv('conv1', x, 2, 3)
        x = F.max_pool2d(F.relu(self.bn('bn1', x)), 3, 2, 1)
        if veil is not None:
            veil = F.max_pool2d(veil, 3, 2, 1)
        x, veil = self.stage('layer1', x, 64, L[0], inplanes=64, veil=veil)
        x, veil = self.stage('layer2', x, 128, L[1], s[0], d[0], inplanes=64 * e, veil=veil)
        if self.family in ('fusionnet', 'partial_fusionnet'):
            yveil = None
            if self.family == 'partial_fusionnet':      # intended wiring of partial_fusionnet.py:250-275: dense RGB stem, partial depth stream
                yveil = (y != 0).float()
                y, yveil = self.pconv('conv2', y, yveil, 2, 3)
                yveil = F.max_pool2d(yveil, 3, 2, 1)
            else:
                y = self.conv('conv2', y, 2, 3)
            y = F.max_pool2d(F.relu(self.bn('bn2', y)), 3, 2, 1)
            y, yveil = self.stage('layer5', y, 64, L[0], inplanes=64, veil=yveil)
            y, yveil = self.stage('layer6', y, 128, L[1], s[0], d[0], inplanes=64 * e, veil=yveil)
            x = F.relu(self.bn('fusion.bn', self.conv('fusion.conv', torch.cat([x, y], 1))))
        x, _ = self.stage('layer3', x, 256, L[2], s[1], d[1], inplanes=128 * e)
        x, _ = self.stage('layer4', x, 512, L[3], s[2], d[2], inplanes=256 * e)
        return self.conv('regressor', x, 1, 1)

    # ---- head + loss + update (utils.py:154-194; depth_train.py:397-405,455-456) ----
    @staticmethod
    def head(z, depth, joints, depth_range):
        b, _, h, w = z.shape
        heat = z.view(b, depth, joints, h, w).permute(0, 2, 3, 4, 1).reshape(b, joints, -1)
        heat = torch.softmax(heat, dim=2).view(b, joints, h, w, depth)
        gy, gx, gz = (torch.linspace(0.0, 2.0, n) for n in (h, w, depth))
        cy = (heat.sum(dim=(3, 4)) * gy).sum(2)
        cx = (heat.sum(dim=(2, 4)) * gx).sum(2)
        cz = (heat.sum(dim=(2, 3)) * gz).sum(2)
        return torch.stack((cx, cy, cz), dim=2) * depth_range

    def train_step(self, color, depth, true_cam, true_val, depth_only=False, depth_dim=16, num_joints=17, depth_range=1000.0,
                   loss_div=10.0, key_index=16, lr=1e-5, weight_decay=4e-5, grad_norm=5.0):
        color, depth = torch.as_tensor(color), torch.as_tensor(depth)
        true_cam, true_val = torch.as_tensor(true_cam), torch.as_tensor(true_val)
        if self.family in ('fusionnet', 'partial_fusionnet'):
            z = self.forward(color, depth)
        else:
            z = self.forward(depth if (depth_only or self.family == 'partial_depthnet') else color)
        relat = self.head(z, depth_dim, num_joints, depth_range)
        relat = relat - relat[:, key_index:key_index + 1]
        spec = relat + true_cam[:, key_index:key_index + 1]
        sel = true_val.view(-1)
        loss = F.smooth_l1_loss(spec.view(-1, 3)[sel] / loss_div, true_cam.view(-1, 3)[sel] / loss_div)
        for t in self.p.values():
            t.grad = None
        loss.backward()
        params = [t for t in self.p.values() if t.grad is not None]
        total = torch.sqrt(sum((t.grad.double() ** 2).sum() for t in params))
        coef = min(grad_norm / (float(total) + 1e-6), 1.0)
        self.step += 1
        bc1, bc2 = 1 - 0.9 ** self.step, 1 - 0.999 ** self.step
        with torch.no_grad():
            for k, t in self.p.items():
                if t.grad is None:
                    continue
                g = t.grad * coef + weight_decay * t
                self.m[k].mul_(0.9).add_(g, alpha=0.1)
                self.v[k].mul_(0.999).addcmul_(g, g, value=0.001)
                t.addcdiv_(self.m[k], (self.v[k].sqrt() / bc2 ** 0.5).add_(1e-8), value=-lr / bc1)
        return dict(loss=float(loss), spec_cam=spec.detach().numpy(), z=z.detach().numpy(), clip_total=float(total),
                    grads={k: t.grad.numpy() for k, t in self.p.items() if t.grad is not None})

    def state(self):
        out = {k: v.detach().numpy() for k, v in self.p.items()}
        out.update({k: v.numpy() for k, v in self.buf.items()})
        return out
