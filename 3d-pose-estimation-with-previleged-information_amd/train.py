"""Legacy RGB-only trainer (reference train.py:12-392) on the HIP path.

Same class contract as the reference: Trainer(args, model, data_info), .train(epoch, loader) -> dict(cam_train_loss),
.test(epoch, loader) -> dict(test_loss, cam_mean, score_pck, score_auc, ...), .adapt_learn_rate(epoch).
Differences from depth_train.Trainer that are reproduced: the loss is taken on raw millimetres (no -loss_div,
train.py:166-174), a single Adam group (train.py:30), and the 60 % / 90 % step schedule of train.py:380-392.
Loader tuples: train (image, true_cam, true_val) (train.py:152), test (image, true_cam, back_rotation, true_val) (train.py:327).

-joint_space (train.py:54-145, 195-307): the model's second head regresses J image-space heat maps; the step adds the image-space loss and, with
-do_track from epoch 2 on, the reconstruction loss through utils.get_recon_cam (least-squares placement of the root-relative pose, p3d_recon_cam_*):
loss = (cam + mat) * 0.5 + recon.  Loader tuples there: train (image, true_cam, true_mat, true_val, intrinsics), test (image, true_cam, true_mat,
back_rotation, true_val, intrinsics).  The reference file is Python 2 era code whose own opts.py / datasets.py no longer provide those loader fields
or the thresh_* flags, and its get_recon_cam raises a NameError; the branches are implemented here as written, fed by any loader with those tuples.
The PCK thresholds come from metadata.json (as in depth_train.py:61) unless args carries thresh_solid/close/rough.
"""
import numpy as np
import torch

from . import depth_train, mat_utils, ops, utils
from . import dist as p3d_dist
from .optim import FlatAdam


class Trainer:

    def __init__(self, args, model, data_info):
        self.model = model
        self.data_info = data_info
        self.list_params = list(model.parameters())
        if args.half_acc:
            # train.py:20-28 halves the model, but its cam_train / joint_train (train.py:54-192) never cast the input nor scale the loss:
            # the legacy trainer cannot train in fp16 as shipped.  depth_main's -half_acc is the working fp16 path.
            raise NotImplementedError('-half_acc is not available in the legacy trainer (use depth_main -half_acc)')
        assert args.do_track <= args.joint_space                                  # main.py:72
        self.optimizer = FlatAdam(list(model.named_parameters()), args.learn_rate, weight_decay=args.weight_decay)
        self.reducer = p3d_dist.GradReducer(self.optimizer, model=model)
        self.world = self.reducer.world
        p3d_dist.broadcast_state(self.optimizer, model)
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
        self.reducer.begin_step()
        count = p3d_dist.global_valid_divisor(true_val) if self.world > 1 else None
        loss, _ = self._head(image, true_cam, true_val, count)
        self.optimizer.zero_grad()
        loss.backward()
        scale = self.reducer.finish()
        self.optimizer.clip_and_step(self.grad_norm, grad_scale=scale)
        return loss.detach()

    def _joint_head(self, image, true_cam, true_mat, true_val, scale=1.0):
        """Both heads and their losses (train.py:78-100): returns cam_loss, mat_loss, spec_cam, spec_mat, relat_cam (root-relative)."""
        side_out = (self.side_in - 1) // self.stride + 1
        cam_feat, mat_feat = self.model(image)
        spec_mat = mat_utils.decode(mat_utils.to_heatmap(mat_feat, self.num_joints, side_out, side_out), self.side_in)
        count = p3d_dist.global_valid_divisor(true_val) if self.world > 1 else None
        mat_loss = ops.masked_loss(spec_mat, true_mat, true_val, self.criterion, None if count is None else count * (2.0 / 3.0))
        relat_cam = utils.decode(utils.to_heatmap(cam_feat, self.depth, self.num_joints, side_out, side_out), self.depth_range)
        key = self.data_info.key_index
        relat_cam = relat_cam - relat_cam[:, key:key + 1]
        spec_cam = relat_cam + true_cam[:, key:key + 1]
        cam_loss = ops.masked_loss(spec_cam, true_cam, true_val, self.criterion, count)
        return cam_loss, mat_loss, spec_cam, spec_mat, relat_cam, count

    def joint_step(self, image, true_cam, true_mat, true_val, intrinsics, do_track):
        self.reducer.begin_step()
        cam_loss, mat_loss, _, spec_mat, relat_cam, count = self._joint_head(image, true_cam, true_mat, true_val)
        loss = cam_loss + mat_loss
        recon_loss = None
        if do_track:                                                               # train.py:105-114
            recon_cam = utils.get_recon_cam(spec_mat, relat_cam, intrinsics, true_val)
            recon_loss = ops.masked_loss(recon_cam, true_cam, true_val, self.criterion, count)
            loss = loss * 0.5 + recon_loss
        self.optimizer.zero_grad()
        loss.backward()
        scale = self.reducer.finish()
        self.optimizer.clip_and_step(self.grad_norm, grad_scale=scale)
        return cam_loss.detach(), mat_loss.detach(), None if recon_loss is None else recon_loss.detach()

    def joint_train(self, epoch, data_loader, cuda_device):
        n_batches = len(data_loader)
        cam_avg = mat_avg = recon_avg = 0.0
        total = 0
        do_track = self.do_track and (epoch != 1)                                  # train.py:64
        for i, (image, true_cam, true_mat, true_val, intrinsics) in enumerate(data_loader):
            image, true_cam, true_mat = image.to(cuda_device), true_cam.to(cuda_device), true_mat.to(cuda_device)
            true_val, intrinsics = true_val.to(cuda_device), intrinsics.to(cuda_device)
            batch = image.size(0)
            cam_loss, mat_loss, recon_loss = self.joint_step(image, true_cam, true_mat, true_val, intrinsics, do_track)
            cam_avg += cam_loss.item() * batch
            mat_avg += mat_loss.item() * batch
            message = '| train Epoch[%d] [%d/%d]  Cam Loss: %1.4f  Mat Loss: %1.4f' % (epoch, i, n_batches, cam_loss.item(), mat_loss.item())
            if do_track:
                recon_avg += recon_loss.item() * batch
                message += '  Recon Loss: %1.4f' % recon_loss.item()
            total += batch
            if self.verbose:
                print(message)
        total = max(total, 1)
        cam_avg, mat_avg, recon_avg = cam_avg / total, mat_avg / total, recon_avg / total
        if self.verbose:
            print('\n=> train Epoch[%d]  Cam Loss: %1.4f  Mat Loss: %1.4f' % (epoch, cam_avg, mat_avg) + ('  Recon Loss: %1.4f' % recon_avg if do_track else '') + '\n')
        return dict(cam_train_loss=cam_avg, mat_train_loss=mat_avg, recon_train_loss=recon_avg)

    def joint_test(self, epoch, test_loader, cuda_device):
        if self.thresh is None:
            raise RuntimeError('evaluation needs PCK thresholds (metadata.json `thresholds` or args.thresh_*)')
        n_batches = len(test_loader)
        cam_avg = mat_avg = 0.0
        total = 0
        mat_stats, cam_stats, det_stats = [], [], []
        for i, (image, true_cam, true_mat, back_rotation, true_val, intrinsics) in enumerate(test_loader):
            image, true_cam, true_mat, true_val = image.to(cuda_device), true_cam.to(cuda_device), true_mat.to(cuda_device), true_val.to(cuda_device)
            batch = image.size(0)
            with torch.no_grad():
                cam_loss, mat_loss, spec_cam, spec_mat, relat_cam, _ = self._joint_head(image, true_cam, true_mat, true_val)
            cam_avg += cam_loss.item() * batch
            mat_avg += mat_loss.item() * batch
            total += batch
            if self.verbose:
                print('| test Epoch[%d] [%d/%d]  Cam Loss: %1.4f  Mat Loss: %1.4f' % (epoch, i, n_batches, cam_loss.item(), mat_loss.item()))
            valid = true_val.cpu().numpy().astype(bool)
            spec_mat_np = spec_mat.cpu().numpy()
            mat_stats.append(mat_utils.analyze(spec_mat_np, true_mat.cpu().numpy(), valid, self.side_in))
            rotate = np.asarray(back_rotation, dtype=np.float32)
            true_np = np.einsum('Bij,BCj->BCi', rotate, true_cam.cpu().numpy())
            cam_stats.append(utils.analyze(np.einsum('Bij,BCj->BCi', rotate, spec_cam.cpu().numpy()), true_np, valid, self.data_info.mirror, self.thresh))
            if self.do_track:                                                      # train.py:274-281
                deter_cam = utils.get_deter_cam(spec_mat_np, relat_cam.cpu().numpy(), np.asarray(intrinsics, dtype=np.float32), valid)
                det_stats.append(utils.analyze(np.einsum('Bij,BCj->BCi', rotate, deter_cam), true_np, valid, self.data_info.mirror, self.thresh))
        total = max(total, 1)
        record = dict(cam_test_loss=cam_avg / total, mat_test_loss=mat_avg / total)
        record.update(mat_utils.parse_epoch(mat_stats))
        record.update(utils.parse_epoch(cam_stats))
        if self.verbose:
            print('\n=> test Epoch[%d]  Cam Loss: %1.4f  Mat Loss: %1.4f\n' % (epoch, record['cam_test_loss'], record['mat_test_loss']))
            print('=> mat_mean: %1.3f  [oks]: %1.3f\n' % (record['mat_mean'], record['score_oks']))
            print('=>[SPEC] cam_mean: %1.3f  [pck]: %1.3f  [auc]: %1.3f\n' % (record['cam_mean'], record['score_pck'], record['score_auc']))
        if self.do_track:
            track_rec = utils.parse_epoch(det_stats)
            if self.verbose:
                print('=>[DETER] cam_mean: %1.3f  [pck]: %1.3f  [auc]: %1.3f\n' % (track_rec['cam_mean'], track_rec['score_pck'], track_rec['score_auc']))
            for key in track_rec:
                record['recon_' + key] = track_rec[key]
        return record

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
        if self.joint_space:
            return self.joint_train(epoch, data_loader, self.list_params[0].device)
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
        if self.joint_space:
            return self.joint_test(epoch, test_loader, self.list_params[0].device)
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
