import importlib, os, sys, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
import test_block_gpu as T

def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()

kind, inp, pl, st, dil, n, h, ds = ('bottleneck', 256, 64, 1, 1, 64, 16, False)
block = T.build(pkg, kind, inp, pl, st, dil, ds, seed=inp + pl)
gen = torch.Generator(device='cuda').manual_seed(3)
x0 = torch.randn(n, inp, h, h, device='cuda', generator=gen).relu_()
dy = torch.randn(n, 256, h, h, device='cuda', generator=gen)
order = sys.argv[1] if len(sys.argv) > 1 else 'pf'
res = {}
for ch in order:
    res[ch] = T.run(pkg, block, x0, dy, fused=(ch == 'f'))
    g = dy * (res[ch]['y'] > 0)
    want = g.double().sum(dim=(0, 2, 3))
    d = (res[ch]['grads']['bn3.bias'].double() - want).abs()
    print(ch, 'dbeta3 max err', float(d.max()), int(d.argmax()))
if 'p' in res and 'f' in res:
    print('dx', rel(res['f']['dx'], res['p']['dx']))
