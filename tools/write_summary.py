#!/usr/bin/env python3
"""Compose profiles/rNN_summary.md from the outputs of tools/profile_round.sh, the final bench / conv_bench logs and tools/pmc.sh."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else 'r01'
go = lambda *p: os.path.join(ROOT, 'gpurun_out', *p)


def lines(path, keep=('shape', 'c', 'total')):
    return '\n'.join(l for l in open(path).read().split('\n') if l.startswith(keep))


subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'summarize_profiles.py'), RND], stdout=subprocess.DEVNULL)
tables = open(go('%s_tables.md' % RND)).read()
fb = json.loads(open(go('final_bench.json')).read().strip().split('\n')[-1])
tr = json.load(open(os.path.join(ROOT, 'profiles', '%s_traffic.json' % RND)))
pm = ''
for f in ('pmc_a.txt', 'pmc_b.txt'):
    if os.path.exists(go(f)):
        pm += open(go(f)).read()
r = fb['roofline']
out = '''# Round %s profile set (MI355X, ROCm 7.2, one GPU)

Produced by `bash tools/profile_round.sh` (GPU box) + `python tools/write_summary.py %s`; raw per-kernel tables:
`%s_serial_kernel_stats.csv`, `%s_overlap_kernel_stats.csv`, `%s_half_kernel_stats.csv`; PMC traffic: `%s_traffic.json`.

## The contract line (un-profiled `python bench.py`, N = 1, 20 timed steps after 5)

```
%s
```

`roofline.achieved` (%.1f TF) comes from HIP-event brackets around the 161 conv calls per step in the serialised kernel pass of that run
(%.1f ms of conv per step); the `serial` rocprofv3 table below is the same configuration under the profiler: the igemm kernel alone plus the
slab_fold / wgrad_reduce / fwd_reduce / dgrad_interleave / weight_tapmajor passes that belong to the same calls.

%s

## HBM-side traffic of the conv kernel (`tools/traffic.sh`: separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, serialised command)

FETCH_SIZE %.1f MB + WRITE_SIZE %.1f MB per launch (raw counters; %d launches).  Algorithmic traffic (every operand and result once) is
~95 MB per launch: the gap is re-reads of activation tiles by the blocks of other M-tiles that miss L2 (before the tap-major weight image
of the large 3x3 layers FETCH was 232 MB per launch).  At ~0.24 ms per launch this is < 1 TB/s of 8 TB/s: MFMA-bound, not HBM-bound.

## Per-shape timing of the conv kernel (`tools/conv_bench.py`, batch 64, the ResNet-50 layer classes of SURVEY.md Appendix A)

```
%s
```

## PMC on two shapes (`tools/pmc.sh`, three `--pmc` passes, 3 launches each; counters are sums over the 8 XCDs / 1024 SIMDs; taken before
## the buffer-store epilogue)

1x1 1024->2048 @16x16 forward, then 3x3 256->256 @16x16 forward.  GRBM_GUI_ACTIVE / 8 / duration = 2.07 and 2.13 GHz under load
(157.3 TF assumes 2.4 GHz); SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) = 72 %% and 67 %% MFMA-busy under the profiler
(launches are 10-15 %% slower under counter collection than in `conv_bench`).

```
%s
```

## fp16 NHWC conv kernels (`tools/hconv_bench.py`, batch 64)

```
%s
```
''' % (RND, RND, RND, RND, RND, RND, json.dumps(fb), r['achieved'], sum(r['conv_ms_per_step'].values()), tables,
       tr['FETCH_SIZE']['kb_per_launch'] * 1024 / 1e6, tr['WRITE_SIZE']['kb_per_launch'] * 1024 / 1e6, tr['FETCH_SIZE']['launches'],
       lines(go('final_convbench.log')), pm.strip(), lines(go('final_hconvbench.log')))
x3_path = os.path.join(ROOT, 'profiles', '%s_bench_x3_optin.json' % RND)
if os.path.exists(x3_path):
    x3 = json.loads(open(x3_path).read().strip().split('\n')[-1])
    o = x3.get('optin_x3') or x3.get('optin_x3_wgrad')
    out += '''

## Opt-in exact-fp32-on-the-bf16-pipe kernels for the dense 1x1 layers (`P3D_X3=1`, DESIGN.md section 9; NOT the contract configuration)

`%s_bench_x3_optin.json`: an un-profiled `python bench.py` line; its `optin_x3` object is the same steps re-timed with the switch on (%.1f ms/step, %.0f crops/s against that
line's %.1f ms / %.0f).  `%s_x3_optin_kernel_stats.csv`: rocprofv3 --kernel-trace --stats of `P3D_X3=1 P3D_WGRAD_STREAM=0 python3 bench.py --steps 5 --warmup 2`
(`p3d::x3_conv1x1_kernel<false>` forward, `<true>` dgrad, `p3d::x3_wgrad1x1_kernel`).  `%s_conv_bench_x3_optin.txt`: per-layer timings with the switch on.
`%s_bf16x3_probe.json` / `%s_bf16x3_probe_pmc.txt`: the stand-alone GEMM probe (tile shapes, pre-split operands, counters).
''' % (RND, o['ms_per_step'], o['value'], x3['ms_per_step'], x3['value'], RND, RND, RND, RND)
open(os.path.join(ROOT, 'profiles', '%s_summary.md' % RND), 'w').write(out)
print('wrote profiles/%s_summary.md' % RND)
