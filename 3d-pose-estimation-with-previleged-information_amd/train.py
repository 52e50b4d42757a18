"""Legacy RGB-only trainer (reference train.py:12-392) on the HIP path.

Same class contract as the reference: Trainer(args, model, data_info), .train(epoch, loader) -> dict(cam_train_loss),
.test(epoch, loader) -> dict(test_loss, cam_mean, score_pck, score_auc, ...), .adapt_learn_rate(epoch).
Differences from depth_train.Trainer that are reproduced: the loss is taken on raw millimetres (no -loss_div,
train.py:166-174), a single Adam group (train.py:30), and the 60 % / 90 % step schedule of train.py:380-392.
Loader tuples: train (image, true_cam, true_val) (train.py:152), test (image, true_cam, back_rotation, true_val) (train.py:327).

The reference file is Python 2 era code whose -joint_space / -do_track branches need loader fields (true_mat, intrinsics) and
flags (thresh_solid, ...) that its own opts.py / datasets.py no longer provide; those branches raise NotImplementedError here.
The PCK thresholds come from metadata.json (as in depth_train.py:61) unless args carries thresh_solid/close/rough.
"""
import numpy as np
import torch

from . import depth_train, ops, utils
from . import dist as p3d_dist
from .optim import FlatAdam


class Trainer:

    def __init__(self, args, model, data_info):
        self.model = model
        self.data_info = data_info
        self.list_params = list(model.parameters())
        if args.half_acc:
            raise NotImplementedError('-half_acc (fp16 copies + static loss scaling, train.py:20-28) is not implemented')
        if args.joint_space or args.do_track:
            raise NotImplementedError('-joint_space / -do_track (train.py:54-145) need the legacy joint-space loader')
        self.optimizer = FlatAdam(list(model.named_parameters()), args.learn_rate, weight_decay=args.weight_decay)
        self.reducer = p3d_dist.GradReducer(self.optimizer)
        self.world = self.reducer.world
        self.depth = args.depth
        self.num_joints = args.num_joints
        self.side_in = args.side_in
        self.stride = args.stride
        self.depth_range = args.depth_range
        self.half_acc = args.half_acc
        self.joint_space = args.joint_space
        self.do_track = args.do_track
        self.learn_rate = args.learn_rate
        self.num_epochs = args.n_epochs
        self.grad_norm = args.grad_norm
        self.grad_scaling = args.grad_scaling
        if hasattr(args, 'thresh_solid'):
            self.thresh = dict(solid=args.thresh_solid, close=args.thresh_close, rough=args.thresh_rough)       # train.py:45-49
        else:
            metadata = depth_train._load_metadata(args)
            self.thresh = metadata['thresholds'][args.data_name] if metadata else None
        self.criterion = args.criterion
        if self.criterion not in ops.CRITERIA:
            raise ValueError('criterion %r is not one of %s' % (self.criterion, sorted(ops.CRITERIA)))
        self.verbose = True

    def _head(self, image, true_cam, true_val, count=None):
        side_out = (self.side_in - 1) // self.stride + 1
        cam_feat = self.model(image)
        heat_cam = utils.to_heatmap(cam_feat, self.depth, self.num_joints, side_out, side_out)
        relat_cam = utils.decode(heat_cam, self.depth_range)
        return ops.pose_loss(relat_cam, true_cam, true_val, self.data_info.key_index, 1.0, self.criterion, count_override=count)

    def train_step(self, image, true_cam, true_val):
        count = p3d_dist.global_valid_divisor(true_val) if self.world > 1 else None
        loss, _ = self._head(image, true_cam, true_val, count)
        self.optimizer.zero_grad()
        loss.backward()
        scale = self.reducer.finish()
        self.optimizer.clip_and_step(self.grad_norm, grad_scale=scale)
        return loss.detach()

    def cam_train(self, epoch, data_loader, cuda_device):
        n_batches = len(data_loader)
        loss_avg, total = 0.0, 0
        for i, (image, true_cam, true_val) in enumerate(data_loader):
            image, true_cam, true_val = image.to(cuda_device), true_cam.to(cuda_device), true_val.to(cuda_device)
            batch = image.size(0)
            value = self.train_step(image, true_cam, true_val).item()
            loss_avg += value * batch
            total += batch
            if self.verbose:
                print('| train Epoch[%d] [%d/%d]  Loss %1.4f' % (epoch, i, n_batches, value))
        loss_avg /= max(total, 1)
        if self.verbose:
            print('\n=> train Epoch[%d]  Cam Loss: %1.4f\n' % (epoch, loss_avg))
        return dict(cam_train_loss=loss_avg)

    def train(self, epoch, data_loader):
        self.model.train()
        self.adapt_learn_rate(epoch)
        return self.cam_train(epoch, data_loader, self.list_params[0].device)

    def cam_test(self, epoch, test_loader, cuda_device):
        if self.thresh is None:
            raise RuntimeError('evaluation needs PCK thresholds (metadata.json `thresholds` or args.thresh_*)')
        n_batches = len(test_loader)
        loss_avg, total, cam_stats = 0.0, 0, []
        for i, (image, true_cam, back_rotation, true_val) in enumerate(test_loader):
            image, true_cam, true_val = image.to(cuda_device), true_cam.to(cuda_device), true_val.to(cuda_device)
            batch = image.size(0)
            with torch.no_grad():
                loss, spec_cam = self._head(image, true_cam, true_val)
            value = loss.item()
            loss_avg += value * batch
            total += batch
            valid = true_val.cpu().numpy().astype(bool)
            rotate = np.asarray(back_rotation, dtype=np.float32)
            spec_np = np.einsum('Bij,BCj->BCi', rotate, spec_cam.cpu().numpy())
            true_np = np.einsum('Bij,BCj->BCi', rotate, true_cam.cpu().numpy())
            cam_stats.append(utils.analyze(spec_np, true_np, valid, self.data_info.mirror, self.thresh))
            if self.verbose:
                print('| test Epoch[%d] [%d/%d]  Cam Loss %1.4f' % (epoch, i, n_batches, value))
        record = dict(test_loss=loss_avg / max(total, 1))
        record.update(utils.parse_epoch(cam_stats))
        if self.verbose:
            print('\n=> test Epoch[%d]  Cam Loss: %1.4f\n' % (epoch, record['test_loss']))
            print('=>[SPEC] cam_mean: %1.3f  [pck]: %1.3f  [auc]: %1.3f\n' % (record['cam_mean'], record['score_pck'], record['score_auc']))
        return record

    def test(self, epoch, test_loader):
        self.model.eval()
        return self.cam_test(epoch, test_loader, self.list_params[0].device)

    def adapt_learn_rate(self, epoch):
        """train.py:380-392"""
        if epoch - 1 < self.num_epochs * 0.6:
            learn_rate = self.learn_rate
        elif epoch - 1 < self.num_epochs * 0.9:
            learn_rate = self.learn_rate * 0.2
        else:
            learn_rate = self.learn_rate * 0.04
        if self.do_track and epoch != 1:
            learn_rate /= 2
        for group in self.optimizer.param_groups:
            group['lr'] = learn_rate
