#!/usr/bin/env python3
"""Where does a CU-masked stream run?  Launches spinning blocks on streams with different masks and counts the distinct (XCC, SE, SH, CU) they land on."""
import collections, ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
L = pkg._lib.lib()
dev = torch.device('cuda', 0)
out = torch.zeros(8192, dtype=torch.int32, device=dev)
for spec in ('none', '256', '192', '64:hi', '64', '128', '0x' + 'f' * 8, '0x' + '01' * 32):
    st = torch.cuda.current_stream(dev) if spec == 'none' else pkg.ops.masked_stream(dev, spec)
    out.zero_()
    torch.cuda.synchronize()
    L.p3d_probe_hw_ids(ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(out.data_ptr()), 8192, 50)
    torch.cuda.synchronize()
    ids = out.cpu().tolist()
    per_xcc = collections.Counter()
    cus = set()
    for v in ids:
        xcc, hw = v >> 16, v & 0xffff
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        cus.add((xcc, se, sh, cu))
    for c in cus:
        per_xcc[c[0]] += 1
    print('%-10s distinct CUs %3d   per XCC %s' % (spec, len(cus), dict(sorted(per_xcc.items()))))
