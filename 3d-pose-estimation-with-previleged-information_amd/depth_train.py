"""Training driver with the reference's Trainer interface (depth_train.py:40-691), on the HIP hot path.

    Trainer(args, model, data_info)
    .train(epoch, loader) -> dict(cam_train_loss=...)      vanilla_train / fusion_train  (depth_train.py:376-462, 286-373)
    .adapt_learn_rate(epoch)                                (depth_train.py:621-638)
    .vanilla_infer / .fusion_infer                          (depth_train.py:650-679)

One step = forward (conv/BN/pool kernels) -> fused soft-argmax head -> fused loss (value + gradient) ->
backward -> [RCCL all-reduce of the flat gradient, overlapped] -> global-norm clip + Adam (2 launches).
Also here: evaluation (`.test`, depth_train.py:477-618), distillation (`-do_teach`: `set_teacher`, `distill_train`, `semi_train`; depth_train.py:115-283),
`-half_acc` (fp16 NHWC kernels, fp32 masters, static loss scale and overflow skip on the device; depth_train.py:73-83,413-449).
Differences from the reference, all documented in DESIGN.md: one process per GPU instead of nn.DataParallel; metadata.json is optional
(`no_depth` / `thresholds` / `loader` / `root` are read when present).
"""
import json
import os

import numpy as np
import torch

from . import dist as p3d_dist
from . import augment, ops, ops_half, utils
from .optim import FlatAdam

root_me = os.path.join(os.sep, 'globalwork', 'liu')      # depth_train.py:12; override with -metadata / $P3D_METADATA


def _load_metadata(args):
    path = getattr(args, 'metadata', None) or os.environ.get('P3D_METADATA') or os.path.join(root_me, 'metadata.json')
    if os.path.exists(path):
        with open(path) as file:
            return json.load(file)
    return None


def get_loader(args):
    """depth_train.py:15-19: the loader module metadata.json names for this dataset ('datasets' = RGB-only tuples,
    'depth_datasets' = RGB + depth tuples); without a metadata file the RGB + depth loader."""
    import importlib
    metadata = _load_metadata(args)
    name = metadata['loader'][args.data_name] if metadata and 'loader' in metadata else 'depth_datasets'
    if name not in ('datasets', 'depth_datasets'):
        raise ValueError('metadata.json names loader module %r; this build ships `datasets` and `depth_datasets`' % name)
    return importlib.import_module('.' + name, package=__package__)


def wrap_by_name(names, params):
    """depth_train.py:22-25 (the two groups get identical hyper-parameters; kept for interface parity)."""
    group_a = [param for name, param in zip(names, params) if 'bn' in name]
    group_b = [param for name, param in zip(names, params) if 'bn' not in name]
    return [dict(params=group_a), dict(params=group_b)]


def to_test_worker(test_loader, no_depth, depth_only, do_fusion=False):
    """Normalises the loader's tuple arity (depth_train.py:28-37) to (color, depth, true_cam, true_val, back_rotate); the
    stream a non-fusion model does not read comes back as None."""
    for items in test_loader:
        if no_depth:
            color, true_cam, true_val, color_br = items
            yield color, None, true_cam, true_val, color_br
        else:
            color, depth, true_cam, true_val, color_br = items
            if do_fusion:
                yield color, depth, true_cam, true_val, color_br
            elif depth_only:
                yield None, depth, true_cam, true_val, color_br
            else:
                yield color, None, true_cam, true_val, color_br


class Trainer:

    def __init__(self, args, model, data_info, reducer_bucket_bytes=p3d_dist.DEFAULT_BUCKET_BYTES):
        if args.semi_teach and not args.do_teach:
            raise ValueError('-semi_teach adds unlabelled pairs to the distillation loss: it needs -do_teach')
        self.model = model
        self.data_info = data_info
        self.list_names = [name for name, param in model.named_parameters()]
        self.list_params = [param for name, param in model.named_parameters()]

        self.half_acc = bool(args.half_acc)
        self.grad_scaling = args.grad_scaling
        self.depth_only = args.depth_only
        self.do_fusion = args.do_fusion
        self.do_teach = args.do_teach
        self.sigmoid = args.sigmoid
        self.bin_dist = args.bin_dist
        self.do_freeze = args.do_freeze
        self.alpha_dest, self.alpha_init, self.alpha_span = args.alpha_dest, args.alpha_init, args.alpha_span
        self.teacher = None
        self.semi_teach = bool(args.semi_teach)
        self.semi_loader = self.semi_worker = None
        # BASELINE config 5: colour / eraser augmentation + normalisation of the RGB stream on the GPU (augment.GpuAugment)
        self.gpu_augment = augment.GpuAugment(args.colour, args.eraser) if (args.colour or args.eraser) else None

        metadata = _load_metadata(args)
        self.no_depth = metadata['no_depth'][args.data_name] if metadata else False
        self.data_name = args.data_name
        self.thresh = metadata['thresholds'][args.data_name] if metadata else None

        if self.semi_teach:
            # depth_train.py:66-70: a second loader over the unlabelled `pku` image pairs, batch -semi_batch.  (The reference rewrites
            # args.data_name / args.batch_size in place for this; a copy keeps the caller's namespace intact.)
            import copy
            semi_args = copy.copy(args)
            semi_args.data_name, semi_args.batch_size = 'pku', args.semi_batch
            self.semi_loader = get_loader(semi_args).data_loader(semi_args, 'train', data_info)
            self.semi_worker = iter(self.semi_loader)

        self.optimizer = FlatAdam(list(model.named_parameters()), args.learn_rate, weight_decay=args.weight_decay)
        self.reducer = p3d_dist.GradReducer(self.optimizer, reducer_bucket_bytes, model=model)
        self.world = self.reducer.world
        p3d_dist.broadcast_state(self.optimizer, model)       # all ranks start from rank 0's replica (no-op without a process group)
        if self.half_acc:
            # depth_train.py:73-83: the reference halves the model and keeps fp32 `copy_params` for Adam.  Here the parameters stay
            # fp32 (they are those copies, already flat inside FlatAdam) and every convolution gets fp16 weight images beside them.
            model._p3d_half = True
            ops_half.refresh_weights(model, self.optimizer.flat_p)

        self.depth = args.depth
        self.num_joints = args.num_joints
        self.side_in = args.side_in
        self.stride = args.stride
        self.depth_range = args.depth_range
        self.warmup = args.warmup
        self.learn_rate = args.learn_rate
        self.learn_decay = args.learn_decay
        self.num_epochs = args.n_epochs
        self.warmup_factor = args.warmup_factor
        self.grad_norm = args.grad_norm
        self.loss_div = args.loss_div
        self.criterion = args.criterion                     # name; the loss kernel implements SmoothL1 / L1 / MSE
        if self.criterion not in ops.CRITERIA:
            raise ValueError('criterion %r is not one of %s' % (self.criterion, sorted(ops.CRITERIA)))
        self.verbose = True
        self.sync_every = 1                                 # read the loss back every k iterations (reference: every one)
        self.last_spec_cam = None

    @property
    def skipped_steps(self):
        """-half_acc: optimizer steps dropped because a gradient overflowed (device counter; reading it synchronises)."""
        return self.optimizer.steps_skipped()

    # ---- process-group ordering ----------------------------------------------------------------
    def warm_memory(self, color_image, depth_image, true_cam, true_val):
        """One forward + backward WITHOUT an optimizer step and with the BatchNorm buffers restored afterwards: it only makes the
        caching allocator own every activation / workspace / gradient block the step needs and creates the weight-gradient stream, so that
        nothing is allocated or created inside the timed steps.  A launcher calls this before dist.init_from_env() and attach_reducer() after
        it (bench.py, depth_main.main); the order no longer matters for speed (the +4 % of round 1 was a hardware-queue
        collision of the weight-gradient stream, now excluded by a probe)."""
        saved = {k: v.clone() for k, v in self.model.state_dict().items() if 'running_' in k or 'num_batches_tracked' in k}
        opt = self.optimizer
        keep = (opt.clip_and_step, opt.clip_and_step_dev)
        opt.clip_and_step = opt.clip_and_step_dev = (lambda *a, **k: True)
        try:
            self.train_step(color_image, depth_image, true_cam, true_val)
        finally:
            opt.clip_and_step, opt.clip_and_step_dev = keep
        self.optimizer.zero_grad()
        with torch.no_grad():
            state = self.model.state_dict()
            for k, v in saved.items():
                state[k].copy_(v)
        torch.cuda.synchronize()

    def attach_reducer(self, bucket_bytes=p3d_dist.DEFAULT_BUCKET_BYTES):
        """(Re)create the gradient reducer: call after the process group has been initialised if the trainer was built before it."""
        self.reducer.remove()
        self.reducer = p3d_dist.GradReducer(self.optimizer, bucket_bytes, model=self.model)
        self.world = self.reducer.world
        p3d_dist.broadcast_state(self.optimizer, self.model)
        if self.half_acc:
            ops_half.refresh_weights(self.model, self.optimizer.flat_p)

    # ---- schedules ---------------------------------------------------------------------------
    def adapt_learn_rate(self, epoch):
        if epoch - 1 < self.warmup:
            learn_rate = self.learn_rate * self.warmup_factor
        elif epoch - 1 < 15:
            learn_rate = self.learn_rate
        elif epoch - 1 < 20:
            learn_rate = self.learn_rate * self.learn_decay
        elif epoch - 1 < 25:
            learn_rate = self.learn_rate * self.learn_decay ** 2
        else:
            learn_rate = self.learn_rate * self.learn_decay ** 3
        for group in self.optimizer.param_groups:
            group['lr'] = learn_rate

    # ---- forward helpers ---------------------------------------------------------------------
    def to(self, image, device):
        return image.to(device, non_blocking=True)

    def vanilla_infer(self, in_image, i_batch=0, ret_last=False):
        cam_feat, last_feat = self.model(in_image)
        return (cam_feat, last_feat) if ret_last else cam_feat

    def fusion_infer(self, color_image, depth_image, i_batch=0, ret_last=False):
        cam_feat, last_feat = self.model(color_image, depth_image)
        return (cam_feat, last_feat) if ret_last else cam_feat

    # ---- the hot loop --------------------------------------------------------------------------
    def train_step(self, color_image, depth_image, true_cam, true_val):
        """One optimisation step on device tensors; returns the loss as a 0-d device tensor (no host sync)."""
        self.reducer.begin_step()
        side_out = (self.side_in - 1) // self.stride + 1
        if self.do_fusion:
            cam_feat = self.fusion_infer(color_image, depth_image)
        else:
            cam_feat = self.vanilla_infer(depth_image if self.depth_only else color_image)
        heat_cam = utils.to_heatmap(cam_feat, self.depth, self.num_joints, side_out, side_out)
        relat_cam = utils.decode(heat_cam, self.depth_range)
        # mean over the valid joints of the GLOBAL batch: per-rank divisor = 3 * global count / world (device side)
        count = p3d_dist.global_valid_divisor(true_val) if self.world > 1 else None
        loss, spec_cam = ops.pose_loss(relat_cam, true_cam, true_val, self.data_info.key_index, self.loss_div,
                                       self.criterion, count_override=count)
        self.last_spec_cam = spec_cam
        self._backward_and_step(loss)
        return loss.detach()

    def _backward_and_step(self, loss):
        """zero_grad -> backward -> gradient exchange -> clip -> Adam (depth_train.py:413-456), in fp32 or with the -half_acc rules."""
        self.optimizer.zero_grad()
        if self.half_acc:
            # static loss scaling (depth_train.py:413-449): gradients carry grad_scaling through the fp16 backward, the optimizer
            # divides it out, and a step whose gradients overflowed is skipped
            loss.backward(torch.full_like(loss, self.grad_scaling))
            ops.check_joins()
            scale = self.reducer.finish()
            # overflow test, skip decision and step counter stay on the device (FlatAdam.clip_and_step_dev): no host read-back per step
            self.optimizer.clip_and_step_dev(self.grad_norm, grad_scale=scale / self.grad_scaling, skip_nonfinite=True)
            ops_half.refresh_weights(self.model, self.optimizer.flat_p)          # (a skipped step re-casts unchanged weights)
            return
        loss.backward()
        ops.check_joins()
        scale = self.reducer.finish()
        self.optimizer.clip_and_step(self.grad_norm, grad_scale=scale)

    def _run_epoch(self, epoch, data_loader, device):
        n_batches = len(data_loader)
        loss_avg = 0.0
        total = 0
        pending = []
        for i_batch, items in enumerate(data_loader):
            if self.no_depth and len(items) == 3:     # RGB-only loader (`datasets`): the reference's loop (depth_train.py:385) only takes 4-tuples,
                items = (items[0], None, items[1], items[2])     # i.e. cannot train on the datasets it marks no_depth; accepted here
                if self.do_fusion or self.depth_only:
                    raise ValueError('dataset %r has no depth stream (metadata.json no_depth): -do_fusion / -depth_only cannot train on it' % self.data_name)
            color_image, depth_image, true_cam, true_val = items
            color_image = self.to(color_image, device) if (self.do_fusion or not self.depth_only) else None
            depth_image = self.to(depth_image, device) if (self.do_fusion or self.depth_only) else None
            if self.gpu_augment is not None and color_image is not None:
                color_image = self.gpu_augment(color_image.contiguous(), train=True)
            true_cam = true_cam.to(device, non_blocking=True)
            true_val = true_val.to(device, non_blocking=True)
            batch = true_cam.size(0)
            loss = self.train_step(color_image, depth_image, true_cam, true_val)
            pending.append((i_batch, batch, loss))
            if len(pending) >= self.sync_every or i_batch == n_batches - 1:
                for ib, b, l in pending:
                    value = l.item()
                    if self.verbose:
                        print('| train Epoch[%d] [%d/%d]  Loss %1.4f' % (epoch, ib, n_batches, value), flush=True)
                    loss_avg += value * b
                    total += b
                pending = []
        loss_avg /= max(total, 1)
        if self.verbose:
            print('\n=> train Epoch[%d]  Cam Loss: %1.4f\n' % (epoch, loss_avg))
        return dict(cam_train_loss=loss_avg)

    def vanilla_train(self, epoch, data_loader, device):
        return self._run_epoch(epoch, data_loader, device)

    def fusion_train(self, epoch, data_loader, device):
        return self._run_epoch(epoch, data_loader, device)

    def train(self, epoch, data_loader):
        self.model.train()
        self.adapt_learn_rate(epoch)
        device = self.list_params[0].device
        if self.do_teach:
            return self.distill_train(epoch, data_loader, device)
        if self.do_fusion:
            return self.fusion_train(epoch, data_loader, device)
        return self.vanilla_train(epoch, data_loader, device)

    # ---- evaluation (depth_train.py:477-607) ------------------------------------------------------
    def _run_test(self, epoch, test_loader, device):
        """vanilla_test / fusion_test: no-grad forward with frozen BN statistics, the same head and loss as training, then the
        reference's host-side metrics on the back-rotated coordinates."""
        if self.thresh is None:
            raise RuntimeError('evaluation needs the `thresholds` entry of metadata.json (-metadata / $P3D_METADATA)')
        n_batches = len(test_loader)
        loss_avg, total, cam_stats = 0.0, 0, []
        side_out = (self.side_in - 1) // self.stride + 1
        fusion = self.do_fusion and not self.do_teach           # under -do_teach the student (single stream) is what gets evaluated
        for i_batch, items in enumerate(to_test_worker(test_loader, self.no_depth, self.depth_only, fusion)):
            color_image, depth_image, true_cam, true_val, color_br = items
            color_image = None if color_image is None else self.to(color_image, device)
            depth_image = None if depth_image is None else self.to(depth_image, device)
            if self.gpu_augment is not None and color_image is not None:
                color_image = self.gpu_augment(color_image.contiguous(), train=False)       # evaluation: ToTensor + Normalize only
            true_cam = true_cam.to(device)
            true_val = true_val.to(device)
            batch = true_cam.size(0)
            with torch.no_grad():
                if fusion:
                    cam_feat = self.fusion_infer(color_image, depth_image, i_batch)
                else:
                    cam_feat = self.vanilla_infer(depth_image if self.depth_only else color_image, i_batch)
                heat_cam = utils.to_heatmap(cam_feat, self.depth, self.num_joints, side_out, side_out)
                relat_cam = utils.decode(heat_cam, self.depth_range)
                loss, spec_cam = ops.pose_loss(relat_cam, true_cam, true_val, self.data_info.key_index, self.loss_div, self.criterion)
            value = loss.item()
            loss_avg += value * batch
            total += batch
            valid = true_val.cpu().numpy().astype(bool)
            rotate = np.asarray(color_br, dtype=np.float32)
            spec_np = np.einsum('Bij,BCj->BCi', rotate, spec_cam.cpu().numpy())         # back-rotation to the original camera
            true_np = np.einsum('Bij,BCj->BCi', rotate, true_cam.cpu().numpy())
            cam_stats.append(utils.analyze(spec_np, true_np, valid, self.data_info.mirror, self.thresh))
            if self.verbose:
                print('| test Epoch[%d] [%d/%d]  Cam Loss %1.4f' % (epoch, i_batch, n_batches, value))
        record = dict(test_loss=loss_avg / max(total, 1))
        record.update(utils.parse_epoch(cam_stats))
        if self.verbose:
            print('\n=> test Epoch[%d]  Cam Loss: %1.4f\n' % (epoch, record['test_loss']))
            print('=>[SPEC] cam_mean: %1.3f  [pck]: %1.3f  [auc]: %1.3f\n' % (record['cam_mean'], record['score_pck'], record['score_auc']))
        return record

    def vanilla_test(self, epoch, test_loader, device):
        return self._run_test(epoch, test_loader, device)

    def fusion_test(self, epoch, test_loader, device):
        return self._run_test(epoch, test_loader, device)

    def test(self, epoch, test_loader):
        self.model.eval()
        if os.environ.get('P3D_RELEASE_ON_EVAL'):          # evaluation needs none of the training plans' device memory (~13 GB for ResNet-50 at batch 64):
            from . import ops_block                          # opt-in, because the next training epoch then re-allocates it (ops_block.release_buffers)
            torch.cuda.synchronize()
            ops_block.release_buffers(self.model)
        return self._run_test(epoch, test_loader, self.list_params[0].device)      # -do_teach evaluates the student (depth_train.py:613-614)

    # ---- distillation: the "privileged information" training (depth_train.py:107-129,161-283,641-647,682-691) --------
    def set_teacher(self, teacher):
        self.teacher = teacher
        if self.half_acc:                                    # depth_train.py:107-108: the teacher runs in fp16 too
            teacher._p3d_half = True
            ops_half.refresh_weights(teacher)

    def get_dist_weight(self, epoch):
        alphas = np.linspace(self.alpha_init, self.alpha_dest, self.alpha_span)
        return float(alphas[epoch - 1]) if epoch - 1 < self.alpha_span else float(self.alpha_dest)

    def freeze_batchnorm(self):
        self.teacher.eval()
        self.model.freeze_batchnorm()

    def teach_infer(self, color_image, depth_image):
        if self.do_fusion:
            return self.teacher(color_image, depth_image)
        return self.teacher(depth_image if self.depth_only else color_image)

    def distill(self, batch, teach_last, last_feat, atten_map, weight=1.0, unit_grad=False):
        """Returns (weight * dist_loss, dist_loss); modes as in the reference: -bin_dist, -sigmoid, plain L2 norm."""
        mode = 'bce' if self.bin_dist else ('sigmoid' if self.sigmoid else 'l2')
        return ops.distill_loss(teach_last, last_feat, atten_map, mode, weight, unit_grad)

    def semi_train(self, device, epoch):
        """depth_train.py:132-153: the distillation loss of one batch of unlabelled image pairs (no pose loss).
        Returns (batch size, alpha-weighted loss for the graph, loss value)."""
        try:
            items = next(self.semi_worker)
        except StopIteration:
            self.semi_worker = iter(self.semi_loader)
            items = next(self.semi_worker)
        color_image, depth_image, true_cam, true_val, atten_map = items
        color_image, depth_image, atten_map = (self.to(t, device) for t in (color_image, depth_image, atten_map))
        with torch.no_grad():
            teach_cam, teach_last = self.teach_infer(color_image, depth_image)
        cam_feat, last_feat = self.vanilla_infer(color_image, 0, True)
        weighted, dist_loss = self.distill(true_cam.size(0), teach_last, last_feat, atten_map.float(), self.get_dist_weight(epoch),
                                           unit_grad=not self.half_acc)
        return true_cam.size(0), weighted, dist_loss

    def distill_step(self, epoch, color_image, depth_image, true_cam, true_val, atten_map):
        """One iteration of distill_train on device tensors; returns (cam_loss, dist_loss) as 0-d device tensors."""
        self.reducer.begin_step()
        side_out = (self.side_in - 1) // self.stride + 1
        with torch.no_grad():
            teach_cam, teach_last = self.teach_infer(color_image, depth_image)
        cam_feat, last_feat = self.vanilla_infer(color_image, 0, True)
        weighted, dist_loss = self.distill(true_cam.size(0), teach_last, last_feat, atten_map, self.get_dist_weight(epoch),
                                           unit_grad=not self.half_acc)      # (under loss scaling the incoming gradient is not 1)
        heat_cam = utils.to_heatmap(cam_feat, self.depth, self.num_joints, side_out, side_out)
        relat_cam = utils.decode(heat_cam, self.depth_range)
        count = p3d_dist.global_valid_divisor(true_val) if self.world > 1 else None
        cam_loss, spec_cam = ops.pose_loss(relat_cam, true_cam, true_val, self.data_info.key_index, self.loss_div, self.criterion,
                                           count_override=count)
        self.last_spec_cam = spec_cam
        loss = weighted + cam_loss                      # dist_loss * alpha + cam_loss (depth_train.py:220)
        self.last_semi = None
        if self.semi_teach:                              # depth_train.py:222-230
            semi_batch, semi_weighted, semi_loss = self.semi_train(color_image.device, epoch)
            loss = loss + semi_weighted
            self.last_semi = (semi_batch, semi_loss.detach())
        self._backward_and_step(loss)
        return cam_loss.detach(), dist_loss.detach()

    def distill_train(self, epoch, data_loader, device):
        if self.teacher is None:
            raise RuntimeError('distill_train: call set_teacher() first')
        n_batches = len(data_loader)
        cam_sum = dist_sum = 0.0
        samples = dist_samples = 0
        if self.do_freeze:
            self.freeze_batchnorm()
        if self.verbose:
            print('\n=> alpha value: {:.2f}'.format(self.get_dist_weight(epoch)))
        for i_batch, (color_image, depth_image, true_cam, true_val, atten_map) in enumerate(data_loader):
            color_image, depth_image, atten_map = (self.to(t, device) for t in (color_image, depth_image, atten_map))
            true_cam, true_val = true_cam.to(device), true_val.to(device)
            batch = true_cam.size(0)
            cam_loss, dist_loss = self.distill_step(epoch, color_image, depth_image, true_cam, true_val, atten_map.float())
            cam_value, dist_value = cam_loss.item(), dist_loss.item()
            cam_sum += cam_value * batch
            dist_sum += dist_value * batch
            samples += batch
            dist_samples += batch
            message = '[=] train Epoch[{0}] Batch[{1}|{2}]  Cam Loss {3:.4f}  Dist Loss {4:.4f} '.format(epoch, i_batch, n_batches, cam_value, dist_value)
            if self.last_semi is not None:               # the unlabelled pairs count towards the distillation average only
                semi_batch, semi_loss = self.last_semi
                dist_sum += semi_loss.item() * semi_batch
                dist_samples += semi_batch
                message += ' Semi Loss {:.4f}'.format(semi_loss.item())
            if self.verbose:
                print(message)
        return dict(dist_train_loss=dist_sum / max(dist_samples, 1), cam_train_loss=cam_sum / max(samples, 1))
