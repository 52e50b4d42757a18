"""Command-line flags of the reference (opts.py:3-78): same names, types and defaults.

The reference parses sys.argv when the module is imported (opts.py:78) and exposes the namespace as
`opts.args`.  Here `args` is resolved on first attribute access, so importing the package does not
demand the five required flags; `parse(argv)` builds a namespace explicitly (tests, bench.py).
Added flags (never renamed ones): -metadata (path of metadata.json, default $P3D_METADATA or the
reference's /globalwork/liu/metadata.json), -synthetic N (N synthetic batches per epoch instead of a dataset).
"""
import argparse

_BOOL_FLAGS = [
    ('shuffle', 'Reshuffle data at each epoch'),
    ('half_acc', 'whether to use float16 for speed-up'),
    ('save_record', 'Path to save train record'),
    ('test_only', 'only performs test'),
    ('val_only', 'only performs validation'),
    ('pretrain', 'whether to load an imagenet pre-train'),
    ('depth_host', 'whether to fill the depth branch with weights from a depth-only pre-train'),
    ('resume', 'whether to continue from a previous checkpoint'),
    ('extra_channel', 'whether to append an extra channel that masks the bbox'),
    ('joint_space', 'whether to allow joint-space train data'),
    ('do_track', 'whether to regress cam coords via least square optim'),
    ('depth_only', 'only accepts depth input'),
    ('nexponent', 'whether to feed in the negative exponent of raw depth values'),
    ('to_depth', 'whether to convert raw depth to actual depth'),
    ('partial_conv', 'whether to replace all convs in Resnet with partial convs'),
    ('do_fusion', 'whether to accept both color and depth input'),
    ('do_teach', 'whether to force a student to mimic its teacher'),
    ('semi_teach', 'whether to force a student to mimic its teacher on additional unlabelled image pairs'),
    ('early_dist', 'whether to impose distillation loss on the third stage feature map'),
    ('skip_relu', 'whether to impose distillation loss on the feature map before relu is applied'),
    ('sigmoid', 'whether to apply sigmoid function to the feature maps before norm is taken'),
    ('bin_dist', 'whether to do pixel-wise binary cross entropy loss for distillation instead'),
    ('attention', 'whether to apply attention map on distillation target'),
    ('save_last', 'whether to save the last feature map of the model'),
    ('do_freeze', 'whether to freeze the batchnorm layers of both networks during distillation'),
    ('geometry', 'whether to perform geometry augmentation'),
    ('colour', 'whether to perform colour augmentation'),
    ('eraser', 'whether to perform eraser augmentation'),
    ('occluder', 'whether to perform occluder augmentation'),
]
_STR_FLAGS = [('model', True, 'Backbone architecture'), ('model_path', False, 'Path to an imagenet pre-train or checkpoint'),
              ('teacher_path', False, 'Path to a checkpoint of the teacher model'),
              ('host_path', False, 'Path to a checkpoint of the depth-only host model'), ('suffix', True, 'Model suffix'),
              ('data_name', True, 'name of dataset'), ('occ_path', False, 'Root path to occluders'),
              ('save_path', True, 'Path to save train record'), ('criterion', True, 'criterion function for estimation loss')]
_INT_FLAGS = dict(warmup=1, n_epochs=20, batch_size=64, semi_batch=16, n_cudas=2, workers=2, num_processes=6, side_in=257,
                  stride=16, num_joints=19, depth=16, alpha_span=10)
_FLOAT_FLAGS = dict(warmup_factor=0.2, learn_rate=5e-5, learn_decay=0.2, grad_norm=5.0, grad_scaling=32.0, momentum=0.9,
                    weight_decay=4e-5, box_margin=0.6, alpha_dest=0.1, alpha_init=0.1, depth_range=1000.0, random_zoom=0.9,
                    loss_div=10.0)


def build_parser():
    parser = argparse.ArgumentParser(description='Parser for all the training options')
    for name, text in _BOOL_FLAGS:
        parser.add_argument('-' + name, action='store_true', help=text)
    for name, required, text in _STR_FLAGS:
        parser.add_argument('-' + name, required=required, help=text)
    for name, default in _INT_FLAGS.items():
        parser.add_argument('-' + name, default=default, type=int)
    for name, default in _FLOAT_FLAGS.items():
        parser.add_argument('-' + name, default=default, type=float)
    # additions of this build (the reference hard-codes the metadata path: depth_train.py:12)
    parser.add_argument('-metadata', default=None, help='path of metadata.json (default: $P3D_METADATA)')
    parser.add_argument('-synthetic', default=0, type=int, help='train on this many synthetic batches per epoch (no dataset needed)')
    return parser


def parse(argv=None):
    return build_parser().parse_args(argv)


def __getattr__(name):
    if name == 'args':
        value = parse()
        globals()['args'] = value
        return value
    raise AttributeError(name)
