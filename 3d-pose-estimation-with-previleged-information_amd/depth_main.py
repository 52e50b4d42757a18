"""Entry point with the reference's factories (depth_main.py:14-164):
    get_info() -> JointInfo          create_model(args) -> (model, state)          main()
Run as  python -m "3d-pose-estimation-with-previleged-information_amd.depth_main" -model resnet50 ... -synthetic 10
(single GPU) or under torchrun for one process per GPU (replaces nn.DataParallel / -n_cudas, depth_main.py:72).
"""
import importlib
import os

import torch

from . import depth_train, dist as p3d_dist
from .utils import get_info   # noqa: F401  (depth_main.get_info, depth_main.py:14-33)


def create_model(args):
    """depth_main.py:36-74: pick {partial_}{depth|fusion}net by flags, build args.model, optionally resume."""
    name = ('partial_' if args.partial_conv else '') + ('fusion' if args.do_fusion else 'depth') + 'net'
    module = importlib.import_module('.' + name, package=__package__)
    assert hasattr(module, args.model)
    model = getattr(module, args.model)(args, args.pretrain)
    state = None
    if args.test_only or args.val_only:
        save_path = os.path.join(args.save_path, args.model + '-' + args.suffix)
        assert os.path.exists(save_path)
        checkpoint = os.path.join(save_path, 'model_{}.pth'.format(args.n_epochs))
        print('=> Loads checkpoint from ' + checkpoint)
        checkpoint = torch.load(checkpoint, map_location='cpu')['model']
        assert len(set(model.state_dict().keys()).difference(set(checkpoint.keys()))) == 0     # depth_main.py:57-60
        model.load_state_dict(checkpoint)
    if args.resume:
        print('=> Loads checkpoint from ' + args.model_path)
        checkpoint = torch.load(args.model_path, map_location='cpu')
        model.load_state_dict(checkpoint['model'])
        state = checkpoint['state']
    return model, state


def create_pair(args):
    """depth_main.py:77-108: teacher = {partial_}{fusion|depth}net loaded from args.teacher_path, student = depthnet.
    (The reference wraps `model` instead of `teacher` in DataParallel when n_cudas > 1, depth_main.py:106; not reproduced.)"""
    name = ('partial_' if args.partial_conv else '') + ('fusion' if args.do_fusion else 'depth') + 'net'
    teacher_module = importlib.import_module('.' + name, package=__package__)
    assert hasattr(teacher_module, args.model)
    teacher = getattr(teacher_module, args.model)(args, False)
    teacher.load_state_dict(torch.load(args.teacher_path, map_location='cpu')['model'])
    student_module = importlib.import_module('.depthnet', package=__package__)
    model = getattr(student_module, args.model)(args, args.pretrain)
    state = None
    if args.resume:
        print('=> Loads checkpoint from ' + args.model_path)
        checkpoint = torch.load(args.model_path, map_location='cpu')
        model.load_state_dict(checkpoint['model'])
        state = checkpoint['state']
    return model, teacher, state


def main(argv=None):
    """depth_main.py:110-161.  One process per GPU (torchrun) replaces nn.DataParallel / -n_cudas."""
    from . import log, opts
    args = opts.parse(argv)
    assert not (args.resume and args.pretrain)
    assert not (args.do_fusion and args.depth_only)
    assert not (args.depth_host and args.depth_only)
    # The process group is joined AFTER the model and the optimizer buffers exist: device memory first allocated once an RCCL
    # communicator is up is slower for the kernels (+4 % step time).
    rank, world, local_rank = p3d_dist.env_ranks()
    torch.cuda.set_device(local_rank)
    say = print if rank == 0 else (lambda *a, **k: None)
    teacher = None
    if args.do_teach:
        model, teacher, state = create_pair(args)
        teacher = teacher.cuda()
    else:
        model, state = create_model(args)
    model = model.cuda()
    say('=> Models are created and filled')

    data_info = get_info()
    module = depth_train.get_loader(args)
    data_loader = None
    if args.test_only:
        test_loader = module.data_loader(args, 'test', data_info)
    elif args.val_only:
        test_loader = module.data_loader(args, 'valid', data_info)
    else:
        test_loader = module.data_loader(args, 'valid', data_info)
        data_loader = module.data_loader(args, 'train', data_info)
    say('=> Dataloaders are ready')

    logger = log.Logger(args, state)
    say('=> Logger is ready')
    trainer = depth_train.Trainer(args, model, data_info)
    trainer.verbose = rank == 0
    if world > 1 or p3d_dist.FORCE_GROUP:
        p3d_dist.init_from_env()
        trainer.attach_reducer()
    say('=> Trainer is ready')
    if teacher is not None:
        trainer.set_teacher(teacher)

    if args.test_only or args.val_only:
        say('=> Evaluation starts')
        test_rec = trainer.test(0, test_loader)
        if rank == 0:
            logger.print_rec(test_rec)
        return test_rec
    start_epoch = logger.state['epoch'] + 1
    say('=> Train starts')
    for epoch in range(start_epoch, args.n_epochs + 1):
        train_rec = trainer.train(epoch, data_loader)
        test_rec = trainer.test(epoch, test_loader) if trainer.thresh is not None else {}     # metrics need metadata thresholds
        logger.record(epoch, train_rec, test_rec, model)
    if rank == 0:
        logger.final_print()
    return logger.state


if __name__ == '__main__':
    main()
