#!/usr/bin/env python3
"""cProfile of the host side of one training step (where the Python / launch time of a step goes)."""
import cProfile
import importlib
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
import bench  # noqa: E402

args = pkg.opts.parse(['-model', 'resnet50'] + bench.FLAGS + (['-half_acc'] if '--half' in sys.argv else []))
model = pkg.depth_main.create_model(args)[0].cuda().train()
tr = pkg.depth_train.Trainer(args, model, pkg.utils.get_info())
tr.verbose = False
tr.adapt_learn_rate(1)
c, d, tc, tv = pkg.synth.make_batch(64, side=256, rank=0, step=0)
b = [torch.from_numpy(a).cuda() for a in (c, tc, tv)]
for _ in range(5):
    tr.train_step(b[0], None, b[1], b[2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.train_step(b[0], None, b[1], b[2])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
