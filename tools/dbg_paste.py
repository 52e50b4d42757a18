import importlib, json, os, sys
import numpy as np, torch
ROOT = '/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'augment.npz'))
for m in json.loads(str(g['meta'])):
    n = m['name']
    if 'occ' not in [k.split('.')[1] for k in g.files if k.startswith(n + '.')]:
        continue
    for rep in range(3):
        img = torch.from_numpy(g[n + '.image'].astype(np.float32).transpose(2, 0, 1)[None].copy()).cuda()
        alpha = g[n + '.alpha'] if m['alpha'] else None
        pkg.augment.paste_over_(img, [g[n + '.occ']], [alpha], g[n + '.center'][None])
        got = img.cpu().numpy()[0].transpose(1, 2, 0)
        want = g[n + '.out'].astype(np.float32)
        bad = np.argwhere(got != want)
        print(n, 'rep', rep, 'image', got.shape, 'occ', g[n + '.occ'].shape, 'alpha', m['alpha'], 'center', g[n + '.center'], 'mismatches', len(bad), [(tuple(b), got[tuple(b)], want[tuple(b)]) for b in bad[:6]])
