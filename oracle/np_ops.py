"""ORACLE (test infrastructure, not product code): numpy restatement of every arithmetic
step on the hot path of Hunger-Prevails/3D-Pose-Estimation-with-Previleged-Information.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (the package's HIP kernels behind include/p3d_hip.h) never does.

Each function cites the reference file:line whose arithmetic it restates.  The reference
itself is eager PyTorch (nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d / F.relu / optim.Adam);
where the reference line is a torch call the restatement follows torch's documented
semantics for that call.  Pinned against the reference by tests/golden/*.npz, which
tests/golden/make_golden.py produced by importing and running the real reference modules
(tests/test_oracle_golden.py).

All functions take/return numpy arrays in NCHW.  `acc` is the accumulation dtype: float64
(default) gives the tightest yardstick for the fp32 GPU kernels; float32 mimics the
reference's own arithmetic width.
"""
import numpy as np


# --------------------------------------------------------------------------------------
# convolution (nn.Conv2d: depthnet.py:16-33,65-89,138,156; resnet.py:142,160-172;
#              fusionnet.py:135,164-165; downsample depthnet.py:167-173)
# --------------------------------------------------------------------------------------

def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def im2col(x, kh, kw, stride, pad, dil):
    """x [N,C,H,W] -> cols [N, C, kh, kw, Ho, Wo] (zero padded taps)."""
    n, c, h, w = x.shape
    ho = conv_out_size(h, kh, stride, pad, dil)
    wo = conv_out_size(w, kw, stride, pad, dil)
    xp = np.zeros((n, c, h + 2 * pad, w + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + h, pad:pad + w] = x
    cols = np.empty((n, c, kh, kw, ho, wo), dtype=x.dtype)
    for r in range(kh):
        for s in range(kw):
            cols[:, :, r, s] = xp[:, :, r * dil:r * dil + stride * (ho - 1) + 1:stride,
                                  s * dil:s * dil + stride * (wo - 1) + 1:stride]
    return cols


def col2im(cols, x_shape, stride, pad, dil):
    """Adjoint of im2col: cols [N,C,kh,kw,Ho,Wo] -> x [N,C,H,W]."""
    n, c, h, w = x_shape
    _, _, kh, kw, ho, wo = cols.shape
    xp = np.zeros((n, c, h + 2 * pad, w + 2 * pad), dtype=cols.dtype)
    for r in range(kh):
        for s in range(kw):
            xp[:, :, r * dil:r * dil + stride * (ho - 1) + 1:stride,
               s * dil:s * dil + stride * (wo - 1) + 1:stride] += cols[:, :, r, s]
    return xp[:, :, pad:pad + h, pad:pad + w]


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, dil=1, acc=np.float64):
    n, c, h, ww = x.shape
    k, c2, kh, kw = w.shape
    assert c == c2
    cols = im2col(x.astype(acc), kh, kw, stride, pad, dil)
    ho, wo = cols.shape[-2:]
    y = np.einsum('km,nmp->nkp', w.reshape(k, -1).astype(acc), cols.reshape(n, c * kh * kw, ho * wo), optimize=True)
    y = y.reshape(n, k, ho, wo)
    if bias is not None:
        y = y + bias.astype(acc).reshape(1, k, 1, 1)
    return y.astype(np.float32)


def conv2d_dgrad(dy, w, x_shape, stride=1, pad=0, dil=1, acc=np.float64):
    n, k, ho, wo = dy.shape
    _, c, kh, kw = w.shape
    cols = np.einsum('km,nkp->nmp', w.reshape(k, -1).astype(acc), dy.reshape(n, k, ho * wo).astype(acc), optimize=True)
    cols = cols.reshape(n, c, kh, kw, ho, wo)
    return col2im(cols, x_shape, stride, pad, dil).astype(np.float32)


def conv2d_wgrad(dy, x, w_shape, stride=1, pad=0, dil=1, acc=np.float64):
    k, c, kh, kw = w_shape
    n = x.shape[0]
    cols = im2col(x.astype(acc), kh, kw, stride, pad, dil)
    ho, wo = cols.shape[-2:]
    dw = np.einsum('nkp,nmp->km', dy.reshape(n, k, ho * wo).astype(acc), cols.reshape(n, c * kh * kw, ho * wo), optimize=True)
    return dw.reshape(w_shape).astype(np.float32)


def conv2d_bgrad(dy, acc=np.float64):
    return dy.astype(acc).sum(axis=(0, 2, 3)).astype(np.float32)


# --------------------------------------------------------------------------------------
# batch norm (nn.BatchNorm2d: depthnet.py:25,34,71,82,90,139,174; momentum 0.1, eps 1e-5)
# --------------------------------------------------------------------------------------

def bn_train_fwd(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, acc=np.float64):
    """Train-mode BN: biased variance normalises, unbiased variance updates running_var.
    Returns y, mean, invstd, new_running_mean, new_running_var."""
    xa = x.astype(acc)
    m = x.shape[0] * x.shape[2] * x.shape[3]
    mean = xa.mean(axis=(0, 2, 3))
    var = ((xa - mean.reshape(1, -1, 1, 1)) ** 2).mean(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + eps)
    y = (xa - mean.reshape(1, -1, 1, 1)) * (invstd * gamma.astype(acc)).reshape(1, -1, 1, 1) + beta.astype(acc).reshape(1, -1, 1, 1)
    new_rm = new_rv = None
    if running_mean is not None:
        unbiased = var * (m / max(m - 1, 1))
        new_rm = ((1 - momentum) * running_mean.astype(acc) + momentum * mean).astype(np.float32)
        new_rv = ((1 - momentum) * running_var.astype(acc) + momentum * unbiased).astype(np.float32)
    return y.astype(np.float32), mean.astype(np.float32), invstd.astype(np.float32), new_rm, new_rv


def bn_train_bwd(dy, x, mean, invstd, gamma, acc=np.float64):
    """Returns dx, dgamma, dbeta for train-mode BN."""
    dya = dy.astype(acc)
    m = x.shape[0] * x.shape[2] * x.shape[3]
    xhat = (x.astype(acc) - mean.astype(acc).reshape(1, -1, 1, 1)) * invstd.astype(acc).reshape(1, -1, 1, 1)
    dbeta = dya.sum(axis=(0, 2, 3))
    dgamma = (dya * xhat).sum(axis=(0, 2, 3))
    scale = (gamma.astype(acc) * invstd.astype(acc)).reshape(1, -1, 1, 1)
    dx = scale * (dya - dbeta.reshape(1, -1, 1, 1) / m - xhat * dgamma.reshape(1, -1, 1, 1) / m)
    return dx.astype(np.float32), dgamma.astype(np.float32), dbeta.astype(np.float32)


def bn_eval_fwd(x, gamma, beta, running_mean, running_var, eps=1e-5, acc=np.float64):
    """Eval-mode BN (model.eval(), depth_train.py:611; freeze_batchnorm depthnet.py:158-161)."""
    scale = gamma.astype(acc) / np.sqrt(running_var.astype(acc) + eps)
    shift = beta.astype(acc) - running_mean.astype(acc) * scale
    return (x.astype(acc) * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)).astype(np.float32)


def bn_eval_bwd(dy, x, gamma, beta, running_mean, running_var, eps=1e-5, acc=np.float64):
    """Frozen-statistics BN backward: dx = dy*scale; dgamma = sum(dy*xhat); dbeta = sum(dy)."""
    invstd = 1.0 / np.sqrt(running_var.astype(acc) + eps)
    dya = dy.astype(acc)
    xhat = (x.astype(acc) - running_mean.astype(acc).reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    dx = dya * (gamma.astype(acc) * invstd).reshape(1, -1, 1, 1)
    return dx.astype(np.float32), (dya * xhat).sum(axis=(0, 2, 3)).astype(np.float32), dya.sum(axis=(0, 2, 3)).astype(np.float32)


# --------------------------------------------------------------------------------------
# ReLU / residual add (F.relu, out + res: depthnet.py:44,53-56,101,105,113-116)
# --------------------------------------------------------------------------------------

def relu_fwd(x):
    return np.maximum(x, 0).astype(np.float32)


def relu_bwd(dy, y):
    """Gradient masked by the *output* being positive (identical to input > 0)."""
    return (dy * (y > 0)).astype(np.float32)


# --------------------------------------------------------------------------------------
# max pool 3x3 stride 2 pad 1 (nn.MaxPool2d: depthnet.py:140,192; on the veil partial_depthnet.py:220)
# --------------------------------------------------------------------------------------

def maxpool3x3s2_fwd(x):
    """Returns y and the argmax tap index (0..8, row-major in the window; first maximum wins,
    as torch's CPU kernel does with its strict '>' comparison)."""
    n, c, h, w = x.shape
    ho = conv_out_size(h, 3, 2, 1, 1)
    wo = conv_out_size(w, 3, 2, 1, 1)
    xp = np.full((n, c, h + 2, w + 2), -np.inf, dtype=x.dtype)
    xp[:, :, 1:1 + h, 1:1 + w] = x
    best = np.full((n, c, ho, wo), -np.inf, dtype=x.dtype)
    idx = np.zeros((n, c, ho, wo), dtype=np.uint8)
    for r in range(3):
        for s in range(3):
            tap = xp[:, :, r:r + 2 * (ho - 1) + 1:2, s:s + 2 * (wo - 1) + 1:2]
            better = tap > best
            best = np.where(better, tap, best)
            idx = np.where(better, np.uint8(r * 3 + s), idx)
    return best.astype(np.float32), idx


def maxpool3x3s2_bwd(dy, idx, x_shape):
    n, c, h, w = x_shape
    ho, wo = dy.shape[-2:]
    dxp = np.zeros((n, c, h + 2, w + 2), dtype=np.float64)
    for r in range(3):
        for s in range(3):
            dxp[:, :, r:r + 2 * (ho - 1) + 1:2, s:s + 2 * (wo - 1) + 1:2] += dy * (idx == r * 3 + s)
    return dxp[:, :, 1:1 + h, 1:1 + w].astype(np.float32)


# --------------------------------------------------------------------------------------
# partial convolution (partial_conv.py:32-57; multi_channel=False so the mask has 1 channel
# and slide_winsize = kh*kw, partial_conv.py:26-28)
# --------------------------------------------------------------------------------------

def mask_count_fwd(mask, kh, kw, stride, pad, dil):
    """partial_conv.py:35-43: cnt = conv(mask, ones); mult = k*k/(cnt+1e-6) * clamp(cnt,0,1);
    mask_out = clamp(cnt,0,1).  mask [N,1,H,W] -> mult, mask_out [N,1,Ho,Wo] (float32 arithmetic
    as in the reference: the mask path runs in fp32 under no_grad)."""
    cols = im2col(mask.astype(np.float32), kh, kw, stride, pad, dil)
    cnt = cols.sum(axis=(2, 3), dtype=np.float32)
    mult = np.float32(kh * kw) / (cnt + np.float32(1e-6))
    mask_out = np.clip(cnt, 0, 1).astype(np.float32)
    mult = (mult * mask_out).astype(np.float32)
    return mult, mask_out


def partial_conv_fwd(x, mask, w, bias=None, stride=1, pad=0, dil=1, acc=np.float64):
    """partial_conv.py:32-57.  Returns (out, mask_out, mult)."""
    kh, kw = w.shape[2:]
    mult, mask_out = mask_count_fwd(mask, kh, kw, stride, pad, dil)
    raw = conv2d_fwd(x * mask, w, bias, stride, pad, dil, acc=acc)
    if bias is not None:
        b = bias.reshape(1, -1, 1, 1)
        out = ((raw - b) * mult + b) * mask_out          # partial_conv.py:48-51
    else:
        out = raw * mult                                   # partial_conv.py:53
    return out.astype(np.float32), mask_out, mult


def partial_conv_bwd(dout, x, mask, w, mult, stride=1, pad=0, dil=1, acc=np.float64):
    """Bias-free case (all partial convs in partial_depthnet.py are bias=False).
    d raw = dout * mult; dx = dgrad(d raw) * mask; dw = wgrad(d raw, x*mask)."""
    draw = (dout * mult).astype(np.float32)
    dx = conv2d_dgrad(draw, w, x.shape, stride, pad, dil, acc=acc) * mask
    dw = conv2d_wgrad(draw, (x * mask).astype(np.float32), w.shape, stride, pad, dil, acc=acc)
    return dx.astype(np.float32), dw


# --------------------------------------------------------------------------------------
# volumetric soft-argmax head (utils.py:154-194)
# --------------------------------------------------------------------------------------

def to_heatmap(z, depth, num_joints, height, width, acc=np.float64):
    """utils.py:154-175: channel = d*J + j (depth-major) -> [B,J,H,W,D], stable softmax over H*W*D."""
    b = z.shape[0]
    heat = z.reshape(b, depth, num_joints, height, width).transpose(0, 2, 3, 4, 1).astype(acc)
    flat = heat.reshape(b, num_joints, -1)
    flat = np.exp(flat - flat.max(axis=2, keepdims=True))
    flat = flat / flat.sum(axis=2, keepdims=True)
    return flat.reshape(b, num_joints, height, width, depth)


def decode(heat, depth_range):
    """utils.py:178-194: marginals, expectation against linspace(0,2,n), stack (x,y,z) * depth_range."""
    heat_y = heat.sum(axis=(3, 4))
    heat_x = heat.sum(axis=(2, 4))
    heat_z = heat.sum(axis=(2, 3))
    gy = np.linspace(0.0, 2.0, heat_y.shape[-1]).reshape(1, 1, -1)
    gx = np.linspace(0.0, 2.0, heat_x.shape[-1]).reshape(1, 1, -1)
    gz = np.linspace(0.0, 2.0, heat_z.shape[-1]).reshape(1, 1, -1)
    cy = (gy * heat_y).sum(axis=2)
    cx = (gx * heat_x).sum(axis=2)
    cz = (gz * heat_z).sum(axis=2)
    return np.stack((cx, cy, cz), axis=2) * depth_range


def softargmax3d_fwd(z, depth, num_joints, height, width, depth_range, acc=np.float64):
    """to_heatmap followed by decode: z [B, D*J, H, W] -> coords [B, J, 3] (x, y, z)."""
    return decode(to_heatmap(z, depth, num_joints, height, width, acc), depth_range).astype(np.float32)


def softargmax3d_bwd(dcoords, z, depth, num_joints, height, width, depth_range, acc=np.float64):
    """Closed form: with p = softmax(l), E_a = sum_i p_i g_a(i):  dL/dl_i = p_i * sum_a dc_a (g_a(i) - E_a) * range."""
    b = z.shape[0]
    p = to_heatmap(z, depth, num_joints, height, width, acc)                     # [B,J,H,W,D]
    gy = np.linspace(0.0, 2.0, height).reshape(1, 1, -1, 1, 1)
    gx = np.linspace(0.0, 2.0, width).reshape(1, 1, 1, -1, 1)
    gz = np.linspace(0.0, 2.0, depth).reshape(1, 1, 1, 1, -1)
    coords = decode(p, 1.0)                                                      # expectations, unit range
    dc = dcoords.astype(acc) * depth_range
    t = (dc[:, :, 0, None, None, None] * (gx - coords[:, :, 0, None, None, None])
         + dc[:, :, 1, None, None, None] * (gy - coords[:, :, 1, None, None, None])
         + dc[:, :, 2, None, None, None] * (gz - coords[:, :, 2, None, None, None]))
    dl = p * t                                                                   # [B,J,H,W,D]
    dz = dl.transpose(0, 4, 1, 2, 3).reshape(b, depth * num_joints, height, width)
    return dz.astype(np.float32)


# --------------------------------------------------------------------------------------
# loss block (depth_train.py:397-405; train.py:166-174 is the same with loss_div = 1)
# --------------------------------------------------------------------------------------

def pose_loss_fwd_bwd(relat, true_cam, true_val, key_index, loss_div, criterion='SmoothL1', acc=np.float64):
    """relat [B,J,3] (decode output), true_cam [B,J,3], true_val [B,J] bool.
    spec = relat - relat[:,key] + true_cam[:,key]; loss = criterion(spec[valid]/div, true[valid]/div), mean
    over the 3*n_valid selected scalars.  Returns (loss, spec_cam, d loss / d relat)."""
    relat = relat.astype(acc)
    tc = true_cam.astype(acc)
    spec = relat - relat[:, key_index:key_index + 1] + tc[:, key_index:key_index + 1]
    val = true_val.astype(bool)
    nsel = int(val.sum()) * 3
    diff = (spec - tc) / loss_div
    if criterion == 'SmoothL1':          # nn.SmoothL1Loss(beta=1, reduction='mean')
        a = np.abs(diff)
        per = np.where(a < 1.0, 0.5 * diff * diff, a - 0.5)
        dper = np.where(a < 1.0, diff, np.sign(diff))
    elif criterion == 'L1':
        per = np.abs(diff)
        dper = np.sign(diff)
    elif criterion == 'MSE':
        per = diff * diff
        dper = 2.0 * diff
    else:
        raise ValueError(criterion)
    m = val[:, :, None]
    loss = (per * m).sum() / max(nsel, 1)
    dspec = dper * m / (loss_div * max(nsel, 1))
    drelat = dspec.copy()
    drelat[:, key_index] -= dspec.sum(axis=1)
    return np.float32(loss), spec.astype(np.float32), drelat.astype(np.float32)


# --------------------------------------------------------------------------------------
# gradient clipping + Adam (depth_train.py:455-456, :83; torch.optim.Adam coupled L2 weight decay)
# --------------------------------------------------------------------------------------

def clip_grad_norm(grads, max_norm, acc=np.float64):
    """nn.utils.clip_grad_norm_: total = ||g||_2 over all tensors; coef = max_norm/(total+1e-6) clamped to 1."""
    total = np.sqrt(sum(float((g.astype(acc) ** 2).sum()) for g in grads))
    coef = min(max_norm / (total + 1e-6), 1.0)
    return total, coef


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, acc=np.float64):
    """torch.optim.Adam (non-amsgrad): g' = g*grad_scale + wd*p; m,v EMA; bias-corrected update.
    `step` is the 1-based count *after* increment.  Returns new (p, m, v)."""
    pa, ga = p.astype(acc), g.astype(acc) * grad_scale
    ga = ga + weight_decay * pa
    m2 = beta1 * m.astype(acc) + (1 - beta1) * ga
    v2 = beta2 * v.astype(acc) + (1 - beta2) * ga * ga
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v2) / np.sqrt(bc2) + eps
    p2 = pa - (lr / bc1) * m2 / denom
    return p2.astype(np.float32), m2.astype(np.float32), v2.astype(np.float32)


def adapt_learn_rate(epoch, learn_rate, warmup=1, warmup_factor=0.2, learn_decay=0.2):
    """depth_train.py:621-638."""
    if epoch - 1 < warmup:
        return learn_rate * warmup_factor
    if epoch - 1 < 15:
        return learn_rate
    if epoch - 1 < 20:
        return learn_rate * learn_decay
    if epoch - 1 < 25:
        return learn_rate * learn_decay * learn_decay
    return learn_rate * learn_decay * learn_decay * learn_decay


# --------------------------------------------------------------------------------------
# colour / eraser augmentation (augment_colour.py:6-67, augment_occluder.py:84-105).
# PARITY UNPINNED for the HSV steps: the reference calls cv2.cvtColor on float32 images and cv2 is
# neither vendored nor installed, and the reference has no fixture for it.  The restatement follows
# OpenCV's documented float conventions (H in [0,360), S,V in [0,1], V = max, S = (V-min)/V,
# H = 60*(G-B)/(V-min) [+120, +240 by argmax], +360 if negative).  Brightness / contrast / the final
# uint8 truncation / the eraser are plain numpy in the reference and are restated exactly.
# --------------------------------------------------------------------------------------

def rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    v = np.max(rgb, axis=-1)
    mn = np.min(rgb, axis=-1)
    diff = v - mn
    eps = np.float32(1.1920929e-07)
    s = np.where(v > eps, diff / np.where(v > eps, v, 1), 0).astype(np.float32)
    safe = np.where(diff > eps, diff, 1)
    h = np.where(v == r, (g - b) * (60 / safe), np.where(v == g, (b - r) * (60 / safe) + 120, (r - g) * (60 / safe) + 240))
    h = np.where(diff > eps, h, 0)
    h = np.where(h < 0, h + 360, h).astype(np.float32)
    return np.stack([h, s, v], axis=-1)


def hsv_to_rgb(hsv):
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    hh = h / np.float32(60)
    sector = np.floor(hh)
    f = (hh - sector).astype(np.float32)
    sector = (sector.astype(np.int64) % 6 + 6) % 6
    p = v * (1 - s)
    q = v * (1 - s * f)
    t = v * (1 - s * (1 - f))
    table = [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)]
    out = np.zeros(hsv.shape, dtype=np.float32)
    for k, (rr, gg, bb) in enumerate(table):
        m = sector == k
        out[..., 0] = np.where(m, rr, out[..., 0])
        out[..., 1] = np.where(m, gg, out[..., 1])
        out[..., 2] = np.where(m, bb, out[..., 2])
    return out


def augment_colour(image_hwc, brightness, contrast, hue, saturation):
    """random_color with its four draws made explicit: image [H,W,3] holding 0..255 values -> same, truncated to integers."""
    x = (image_hwc / 255.0).astype(np.float32)
    x = np.clip(x + np.float32(brightness), 0, 1)                                   # augment_colour.py:6-12
    x = np.clip((x - np.float32(0.5)) * np.float32(contrast) + np.float32(0.5), 0, 1)   # :15-24
    hsv = rgb_to_hsv(x.astype(np.float32))
    hsv[..., 0] += np.float32(hue)                                                  # :27-36
    hsv[..., 0][hsv[..., 0] < 0] += 360
    hsv[..., 0][hsv[..., 0] >= 360] -= 360
    hsv[..., 1] = np.clip(hsv[..., 1] * np.float32(saturation), 0, 1)               # :39-45
    x = hsv_to_rgb(hsv)
    return np.floor(np.clip(x * np.float32(255), 0, 255)).astype(np.float32)        # (dest * 255).astype(np.uint8), :67


def erase_rect(image_shape_hw, area_frac, aspect, start_frac):
    """random_erase geometry (augment_occluder.py:87-101) with its draws made explicit -> (x0, y0, x1, y1)."""
    h, w = image_shape_hw
    erase_area = area_frac * h * w
    eh = (erase_area * aspect) ** 0.5
    ew = (erase_area / aspect) ** 0.5
    start = (np.array([h, w]) - np.array([eh, ew])) * np.asarray(start_frac)
    end = start + np.array([eh, ew])
    start = np.round(start).astype(int)
    end = np.round(end).astype(int)
    return int(start[1]), int(start[0]), int(end[1]), int(end[0])


def crop_homography(old_intrinsics, old_rotation, new_intrinsics, new_rotation):
    """cameralib.reproject_image_fast (cameralib.py:672-674): pixel of the NEW image -> pixel of the OLD image."""
    old_matrix = np.asarray(old_intrinsics, np.float64) @ np.asarray(old_rotation, np.float64)
    new_matrix = np.asarray(new_intrinsics, np.float64) @ np.asarray(new_rotation, np.float64)
    return (old_matrix @ np.linalg.inv(new_matrix)).astype(np.float32)


def warp_crop(image_hwc, homography, out_hw):
    """Restatement of the remap of cameralib.reproject_image_fast (cameralib.py:692-706): float32 coordinates as the reference
    computes them, bilinear interpolation with constant border 0.  cv2 itself is absent here and quantises the sample position to
    1/32 px with 15-bit weights; this plain bilinear form is the documented definition -> parity with cv2 is UNPINNED."""
    ho, wo = out_hw
    y, x = np.mgrid[:ho, :wo].astype(np.float32)
    coords = np.asarray(homography, np.float32) @ np.stack([x, y, np.ones_like(x)], 0).reshape(3, -1)
    sx, sy = (coords[0] / coords[2]).astype(np.float32), (coords[1] / coords[2]).astype(np.float32)
    img = np.asarray(image_hwc)
    hs, ws = img.shape[:2]
    img = img.reshape(hs, ws, -1).astype(np.float32)
    fx, fy = np.floor(sx), np.floor(sy)
    ax, ay = (sx - fx)[:, None], (sy - fy)[:, None]
    x0, y0 = fx.astype(np.int64), fy.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < ws) & (yy >= 0) & (yy < hs)
        v = img[np.clip(yy, 0, hs - 1), np.clip(xx, 0, ws - 1)]
        return np.where(ok[:, None], v, np.float32(0))
    out = (tap(y0, x0) * (1 - ax) + tap(y0, x0 + 1) * ax) * (1 - ay) + (tap(y0 + 1, x0) * (1 - ax) + tap(y0 + 1, x0 + 1) * ax) * ay
    if np.asarray(image_hwc).dtype == np.uint8:
        out = np.rint(out)
    return out.reshape(ho, wo, -1).transpose(2, 0, 1).astype(np.float32)


# ---- legacy joint-space head (train.py:78-114) -------------------------------------------------------------------------------------
def recon_cam(spec_mat, relat_cam, intrinsics):
    """utils.get_recon_cam / get_deter_cam (utils.py:298-366), written with the explicit [2J x 3] system as the reference builds it.
    Returns (recon, cache)."""
    spec_mat, relat_cam, intrinsics = (np.asarray(a, np.float64) for a in (spec_mat, relat_cam, intrinsics))
    batch, joints = spec_mat.shape[:2]
    kinv = np.linalg.inv(intrinsics)
    homog = np.concatenate([spec_mat, np.ones((batch, joints, 1))], -1)
    normalized = np.einsum('bij,bkj->bik', homog, kinv)[:, :, :2]
    A = np.concatenate([np.tile(np.eye(2), (batch, joints, 1)), -normalized.reshape(batch, -1, 1)], -1)
    rhs = (normalized * relat_cam[:, :, 2:] - relat_cam[:, :, :2]).reshape(batch, -1, 1)
    At = A.transpose(0, 2, 1)
    minv = np.linalg.inv(At @ A)
    refer = minv @ (At @ rhs)
    return relat_cam + refer.transpose(0, 2, 1), (A, rhs, minv, refer, normalized, kinv, relat_cam)


def recon_cam_bwd(drecon, cache):
    """Gradients of recon_cam w.r.t. (spec_mat, relat_cam) by matrix calculus on the explicit system: t = M^-1 A^T b, M = A^T A."""
    A, rhs, minv, refer, normalized, kinv, relat_cam = cache
    drecon = np.asarray(drecon, np.float64)
    batch, joints = drecon.shape[:2]
    gsum = drecon.sum(axis=1)[:, :, None]                     # d L / d t
    w = minv @ gsum                                           # M symmetric
    dM = -w @ refer.transpose(0, 2, 1)
    dA = A @ (dM + dM.transpose(0, 2, 1)) + rhs @ w.transpose(0, 2, 1)
    drhs = (A @ w).reshape(batch, joints, 2)
    dnorm = -dA[:, :, 2].reshape(batch, joints, 2) + drhs * relat_cam[:, :, 2:]
    drelat = drecon.copy()
    drelat[:, :, :2] -= drhs
    drelat[:, :, 2] += (drhs * normalized).sum(-1)
    dspec = np.einsum('bjk,bki->bji', dnorm, kinv[:, :2, :2])
    return dspec, drelat


def masked_loss(pred, target, valid, criterion='SmoothL1'):
    """criterion(pred[valid], target[valid]) with mean reduction (train.py:94,100,112) -> (loss, dpred)."""
    pred, target = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    mask = np.asarray(valid, bool)[..., None]
    diff = (pred - target) * mask
    count = max(mask.sum() * pred.shape[-1], 1)
    if criterion == 'SmoothL1':
        per = np.where(np.abs(diff) < 1, 0.5 * diff * diff, np.abs(diff) - 0.5)
        dper = np.where(np.abs(diff) < 1, diff, np.sign(diff))
    elif criterion == 'L1':
        per, dper = np.abs(diff), np.sign(diff)
    else:
        per, dper = diff * diff, 2 * diff
    return float((per * mask).sum() / count), dper * mask / count


def softargmax2d(z, map_range):
    """mat_utils.to_heatmap + decode (mat_utils.py:32-56): softmax over H*W, expectation against linspace(0, 1, n) * map_range."""
    z = np.asarray(z, np.float64)
    b, j, h, w = z.shape
    flat = z.reshape(b, j, -1)
    heat = np.exp(flat - flat.max(-1, keepdims=True))
    heat = (heat / heat.sum(-1, keepdims=True)).reshape(b, j, h, w)
    gx, gy = np.linspace(0, 1, w), np.linspace(0, 1, h)
    coords = np.stack([(heat.sum(2) * gx).sum(-1), (heat.sum(3) * gy).sum(-1)], -1) * map_range
    return coords, heat


def softargmax2d_bwd(dcoords, heat, coords, map_range):
    b, j, h, w = heat.shape
    gx, gy = np.linspace(0, 1, w) * map_range, np.linspace(0, 1, h) * map_range
    dc = np.asarray(dcoords, np.float64)
    field = dc[:, :, 0, None, None] * (gx[None, None, None, :] - coords[:, :, 0, None, None]) + \
        dc[:, :, 1, None, None] * (gy[None, None, :, None] - coords[:, :, 1, None, None])
    return heat * field


# --------------------------------------------------------------------------------------
# augmentation (augment_occluder.py:7-55, augment_colour.py:6-24,48-67)
# --------------------------------------------------------------------------------------

def paste_over(occluder, image, alpha, center):
    """augment_occluder.paste_over on a uint8 HWC image; float slice bounds truncate as in the numpy the reference was written for."""
    shape_occ = np.array(occluder.shape[:2])
    shape_image = np.array(image.shape[:2])
    center = np.round(center).astype(int)
    ideal_start_dst = center - shape_occ / 2
    ideal_end_dst = ideal_start_dst + shape_occ
    start_dst = np.maximum(ideal_start_dst, 0)
    end_dst = np.minimum(ideal_end_dst, shape_image)
    start_src = start_dst - ideal_start_dst
    end_src = shape_occ + (end_dst - ideal_end_dst)
    d0, d1 = [int(v) for v in start_dst], [int(v) for v in end_dst]
    s0, s1 = [int(v) for v in start_src], [int(v) for v in end_src]
    if alpha is None:
        alpha = np.ones(occluder.shape[:2], dtype=np.float32)
    if alpha.ndim < occluder.ndim:
        alpha = np.expand_dims(alpha, -1)
    alpha = alpha[s0[0]:s1[0], s0[1]:s1[1]]
    occ = occluder[s0[0]:s1[0], s0[1]:s1[1]]
    region = image[d0[0]:d1[0], d0[1]:d1[1]]
    image[d0[0]:d1[0], d0[1]:d1[1]] = alpha * occ + (1 - alpha) * region
    return image


def brightness_contrast(image_u8, brightness, contrast):
    """augment_colour.random_color with no hue / saturation jitter: (image/255) + b, clip, (x - .5) * c + .5, clip, * 255 -> uint8 (fp32 steps)."""
    x = (image_u8 / 255.0).astype(np.float32)
    x += brightness
    x = np.clip(x, 0, 1)
    x -= 0.5
    x *= contrast
    x += 0.5
    x = np.clip(x, 0, 1)
    return (x * 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# convolution evaluated only at chosen outputs (float64): parity checks at BASELINE's full sizes,
# where the dense oracle above would take minutes per layer
# --------------------------------------------------------------------------------------

def conv2d_fwd_at(x, w, bias, stride, pad, dil, idx):
    """y[n, k, ho, wo] of conv2d_fwd at the rows of idx [S, 4] (n, k, ho, wo)."""
    n, k, ho, wo = (idx[:, i] for i in range(4))
    _, c, h, wd = x.shape
    out = np.zeros(len(idx), dtype=np.float64)
    for r in range(w.shape[2]):
        for s in range(w.shape[3]):
            hi, wi = ho * stride - pad + r * dil, wo * stride - pad + s * dil
            ok = (hi >= 0) & (hi < h) & (wi >= 0) & (wi < wd)
            xs = x[n[ok], :, hi[ok], wi[ok]].astype(np.float64)                 # [S', C]
            out[ok] += (xs * w[k[ok], :, r, s].astype(np.float64)).sum(axis=1)
    if bias is not None:
        out += bias[k].astype(np.float64)
    return out


def conv2d_dgrad_at(dy, w, x_shape, stride, pad, dil, idx):
    """dx[n, c, hi, wi] of conv2d_dgrad at the rows of idx [S, 4] (n, c, hi, wi)."""
    n, c, hi, wi = (idx[:, i] for i in range(4))
    ho_n, wo_n = dy.shape[2:]
    out = np.zeros(len(idx), dtype=np.float64)
    for r in range(w.shape[2]):
        for s in range(w.shape[3]):
            th, tw = hi + pad - r * dil, wi + pad - s * dil
            ok = (th % stride == 0) & (tw % stride == 0)
            ho, wo = th // stride, tw // stride
            ok &= (ho >= 0) & (ho < ho_n) & (wo >= 0) & (wo < wo_n)
            ds = dy[n[ok], :, ho[ok], wo[ok]].astype(np.float64)                # [S', K]
            out[ok] += (ds * w[:, c[ok], r, s].T.astype(np.float64)).sum(axis=1)
    return out


def conv2d_wgrad_block(dy, x, ksel, csel, kh, kw, stride, pad, dil):
    """dw[ksel][:, csel] of conv2d_wgrad: the [len(ksel), len(csel), kh, kw] block of filters ksel and input channels csel (all taps)."""
    nb, _, h, wd = x.shape
    ho_n, wo_n = dy.shape[2:]
    dys = dy[:, ksel].astype(np.float64)                                        # [N, K', Ho, Wo]
    xsel = x[:, csel].astype(np.float64)
    out = np.zeros((len(ksel), len(csel), kh, kw), dtype=np.float64)
    for r in range(kh):
        for s in range(kw):
            # output rows ho whose tap (r, s) lands inside the image: hi = ho*stride - pad + r*dil
            ho_idx = np.arange(ho_n)
            hi = ho_idx * stride - pad + r * dil
            okh = (hi >= 0) & (hi < h)
            wo_idx = np.arange(wo_n)
            wi = wo_idx * stride - pad + s * dil
            okw = (wi >= 0) & (wi < wd)
            if not okh.any() or not okw.any():
                continue
            a = dys[:, :, ho_idx[okh]][:, :, :, wo_idx[okw]]                     # [N, K', h', w']
            b = xsel[:, :, hi[okh]][:, :, :, wi[okw]]                            # [N, C', h', w']
            out[:, :, r, s] = np.einsum('nkp,ncp->kc', a.reshape(nb, len(ksel), -1), b.reshape(nb, len(csel), -1), optimize=True)
    return out
