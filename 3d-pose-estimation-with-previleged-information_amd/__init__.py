"""MI355X-native hot path of 3D-Pose-Estimation-with-Previleged-Information.

Host side mirrors the reference's module names (opts, depth_main, depth_train, depthnet, fusionnet,
partial_conv, partial_depthnet, resnet, utils, depth_datasets, datasets); all arithmetic on the path
runs in hand-written gfx950 HIP kernels behind the C ABI of include/p3d_hip.h (csrc/libp3d_hip.so).
The directory name is not a Python identifier: import it with importlib.import_module(), or through
the `p3d_amd` alias module at the repository root.  See DESIGN.md.
"""
from . import synth  # noqa: F401  (numpy only)


def __getattr__(name):
    # torch-dependent submodules are imported on first use so that numpy-only helpers stay light
    import importlib
    if name in ('ops', 'nn', 'optim', 'dist', 'utils', 'opts', 'depthnet', 'resnet', 'fusionnet', 'partial_conv',
                'partial_depthnet', 'partial_fusionnet', 'cameralib', 'crops', 'mat_utils', 'depth_train', 'depth_main', 'depth_datasets', 'datasets', 'joint_settings', '_lib', 'log', 'train', 'main',
                'augment', 'graphed', 'ops_half', 'ops_block', '_trunk'):
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
