"""Checkpoint / record keeper with the reference's file formats (log.py:6-81), so runs interchange with it:

    <save_path>/<model>-<suffix>/model_<epoch>.pth   = {'state': {best_auc, best_pck, best_epoch, epoch}, 'model': state_dict}
    <save_path>/<model>-<suffix>/best.pth            = {'best': epoch}
    <save_path>/<model>-<suffix>/train_record.pth    = {metric name: [value per epoch]}

Tensors in 'model' are saved from host copies under the reference's parameter names (conv1.weight ... regressor.bias).
Under torchrun only rank 0 writes (one process per GPU replaces nn.DataParallel; the `.module` unwrap of log.py:29-30 is kept
for wrapped models).
"""
import os

import torch


class Logger:

    def __init__(self, args, state):
        self.state = state if state else dict(best_auc=0, best_pck=0, best_epoch=0, epoch=0)          # log.py:8
        os.makedirs(args.save_path, exist_ok=True)
        self.save_path = os.path.join(args.save_path, args.model + '-' + args.suffix)
        os.makedirs(self.save_path, exist_ok=True)
        assert args.save_record != (args.test_only or args.val_only)                                   # log.py:18
        self.save_record = args.save_record
        record_path = os.path.join(self.save_path, 'train_record.pth')
        self.train_record = torch.load(record_path) if args.resume and os.path.exists(record_path) else None
        self.is_writer = int(os.environ.get('RANK', '0')) == 0

    def record(self, epoch, train_recs, test_recs, model):
        if hasattr(model, 'module'):
            model = model.module
        self.state['epoch'] = epoch
        if train_recs and self.is_writer:
            host_state = {key: value.detach().cpu() for key, value in model.state_dict().items()}
            torch.save(dict(state=self.state, model=host_state), os.path.join(self.save_path, 'model_%d.pth' % epoch))
        if test_recs:
            score_sum = test_recs['score_auc'] + test_recs['score_pck']
            best_sum = self.state['best_auc'] + self.state['best_pck']
            if score_sum > best_sum:
                self.state['best_epoch'] = epoch
                self.state['best_auc'] = test_recs['score_auc']
                self.state['best_pck'] = test_recs['score_pck']
                if self.is_writer:
                    torch.save({'best': epoch}, os.path.join(self.save_path, 'best.pth'))
        train_recs.update(test_recs)
        if self.save_record:
            if self.train_record:
                self.train_record = {key: self.train_record[key] + [train_recs[key]] for key in train_recs}
            else:
                self.train_record = {key: [train_recs[key]] for key in train_recs}
            if self.is_writer:
                torch.save(self.train_record, os.path.join(self.save_path, 'train_record.pth'))
                print('- train record saved to', os.path.join(self.save_path, 'train_record.pth'), '\n')

    def final_print(self):
        print('[=] Best:  epoch: {:3d}  auc: {:6.3f}  pck: {:6.3f}'.format(self.state['best_epoch'], self.state['best_auc'],
                                                                          self.state['best_pck']))

    def print_rec(self, record):
        for key, value in record.items():
            print('{:>9}'.format(key) + ':', '{:.4f}'.format(value))
