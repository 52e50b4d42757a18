"""bench.py prints exactly one JSON line with the contract's keys (driver contract + the `roofline` / `cpu_baseline` objects)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_schema():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--cpu-steps', '1'],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    for key, kind in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int), ('ms_per_step', float),
                      ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str), ('config', dict), ('roofline', dict),
                      ('cpu_baseline', dict)):
        assert isinstance(line[key], kind), key
    assert 'vs_baseline' in line and line['vs_baseline'] is None          # BASELINE.md publishes no number for this metric
    assert line['unit'] == 'crops/s' and line['n_gpus'] == 1 and line['steps'] == 3 and line['warmup'] == 1
    x3 = os.environ.get('P3D_X3', '1') != '0'                       # (the suite also runs with the x3 kernels switched off)
    assert line['scaling'] == 'weak' and (line['dtype'].startswith('f32 (3xbf16 split') if x3 else line['dtype'] == 'f32') and line['data'] == 'synthetic' and line['higher_is_better'] is True
    assert line['fwd_bwd_crops_per_s'] >= line['value'] * 0.98          # forward + backward alone is never slower than the full step
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert line['value'] == pytest.approx(64 * 3 / (line['ms_per_step'] * 3 / 1e3), rel=1e-3)
    r = line['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == pytest.approx(2500.0 / 6 if x3 else 157.3, abs=0.1) and r['fp32_mfma_peak'] == 157.3
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], abs=1e-3) and 0.1 < r['frac'] < 1.0
    assert r['frac_of_fp32_mfma_peak'] == pytest.approx(r['achieved'] / 157.3, abs=1e-3)
    if x3:
        assert r['conv_paths']['x3_flop_fraction'] > 0.99 and r['conv_paths']['fp32_mfma_launches_per_step'] <= 1    # the stem included (space-to-depth restatement); every launch is counted
    else:
        assert r['conv_paths']['x3_launches_per_step'] == 0
    assert r['traffic'] is None or r['traffic'] > 0
    # `achieved` charges a launch with the split-K / slab sums behind its kernel (rounds 1-3); the conv kernels alone stay beside it and are never slower
    w = r['kernels_alone']
    assert w['achieved'] >= r['achieved'] * 0.95 and w['avg_launch_ms'] <= r['avg_launch_ms'] * 1.05 and w['frac'] == pytest.approx(w['achieved'] / r['peak'], abs=1e-3)
    assert set(r['conv_ms_per_step']) == {'fwd', 'dgrad', 'wgrad'} and 'rocprofv3' in r['accounting']
    side = line['fp32_mfma_only']                                   # informational side measurement, never `value`
    assert side['unit'] == 'crops/s' and side['value'] > 0 and 'NOT the contract' in side['note']
    c = line['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['unit'] == 'crops/s' and c['value'] > 0 and c['cores'] >= 1 and c['sample']
    # the CPU leg's first step doubles as the oracle of the contract workload (ResNet-50, batch 64): north_star bar 1e-3 on loss and joints
    par = line['parity_at_contract_batch']
    assert par['batch'] == 64 and par['loss_rel'] < 1e-3 and par['spec_cam_rel'] < 1e-3, par
    assert line['config']['dist_backend'] is None and line['config']['dist_world_size'] == 1          # N = 1: no process group


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` without a launcher starts its own two ranks (depth_main.py:72's -n_cudas N needs no launcher either).  On the one-GPU
    box both ranks share cuda:0 and exchange gradients through gloo (P3D_BENCH_SHARE_GPU=1): the launcher, rendezvous on 127.0.0.1, replica
    broadcast, bucketed all-reduce, barriers and MAX-reduced timing of the N > 1 flow run end to end; rank 0 prints ONE line."""
    env = dict(os.environ, P3D_BENCH_SHARE_GPU='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--lean', '--steps', '3', '--warmup', '1', '--batch', '8'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['lean'] is True and line['value'] > 0
    assert line['dist_backend'] == 'gloo' and line['dist_world_size'] == 2          # what the process group itself reports
