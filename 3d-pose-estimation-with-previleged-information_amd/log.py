"""Run directory of a training job, in the reference's file formats (log.py:6-81) so that checkpoints and records interchange with it:

    <save_path>/<model>-<suffix>/model_<epoch>.pth   {'state': {best_auc, best_pck, best_epoch, epoch}, 'model': state_dict}
    <save_path>/<model>-<suffix>/best.pth            {'best': epoch}
    <save_path>/<model>-<suffix>/train_record.pth    {metric name: [value per epoch]}

Tensors in 'model' are host copies under the reference's parameter names (conv1.weight ... regressor.bias).  Under torchrun every rank keeps the
bookkeeping (state, best epoch, record) but only rank 0 touches the disk: one process per GPU replaces nn.DataParallel, whose `.module` unwrap
(log.py:29-30) is kept for wrapped models.
"""
import os

import torch

FRESH_STATE = dict(best_auc=0, best_pck=0, best_epoch=0, epoch=0)                 # log.py:8


class Logger:

    def __init__(self, args, state):
        if args.save_record == bool(args.test_only or args.val_only):             # log.py:18: records belong to training runs, and only to them
            raise AssertionError('-save_record goes with training runs only')
        self.state = state or dict(FRESH_STATE)
        self.save_record = args.save_record
        self.is_writer = int(os.environ.get('RANK', '0')) == 0
        self.save_path = os.path.join(args.save_path, '%s-%s' % (args.model, args.suffix))
        os.makedirs(self.save_path, exist_ok=True)
        self.train_record = None
        if args.resume and os.path.exists(self._file('train_record.pth')):
            self.train_record = torch.load(self._file('train_record.pth'))

    def _file(self, name):
        return os.path.join(self.save_path, name)

    def _write(self, payload, name):
        if self.is_writer:
            torch.save(payload, self._file(name))

    def _checkpoint(self, epoch, model):
        weights = {key: tensor.detach().cpu() for key, tensor in getattr(model, 'module', model).state_dict().items()}
        self._write({'state': self.state, 'model': weights}, 'model_%d.pth' % epoch)

    def _keep_best(self, epoch, scores):
        """The epoch with the highest auc + pck so far is remembered in the state and in best.pth (log.py:42-52)."""
        if scores['score_auc'] + scores['score_pck'] <= self.state['best_auc'] + self.state['best_pck']:
            return
        self.state.update(best_epoch=epoch, best_auc=scores['score_auc'], best_pck=scores['score_pck'])
        self._write({'best': epoch}, 'best.pth')

    def _extend_record(self, row):
        past = self.train_record or {}
        self.train_record = {key: past.get(key, []) + [value] for key, value in row.items()}
        self._write(self.train_record, 'train_record.pth')
        if self.is_writer:
            print('- train record saved to', self._file('train_record.pth'), '\n')

    def record(self, epoch, train_recs, test_recs, model):
        self.state['epoch'] = epoch
        if train_recs:
            self._checkpoint(epoch, model)
        if test_recs:
            self._keep_best(epoch, test_recs)
        train_recs.update(test_recs)                                               # the caller's dict becomes the epoch's row, as in log.py:54
        if self.save_record:
            self._extend_record(train_recs)

    def final_print(self):
        print('[=] Best:  epoch: {best_epoch:3d}  auc: {best_auc:6.3f}  pck: {best_pck:6.3f}'.format(**self.state))

    def print_rec(self, record):
        for key, value in record.items():
            print('%9s: %.4f' % (key, value))
