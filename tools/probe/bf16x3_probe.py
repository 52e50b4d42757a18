#!/usr/bin/env python3
"""Probe for DESIGN.md section 9: rate and accuracy of an exact-fp32 GEMM issued on the bf16 MFMA pipe (three-piece operand split), against
the fp32 MFMA paths on the same shapes (torch.matmul -> hipBLASLt/rocBLAS fp32, and the product's 1x1 convolution kernel).
Shapes are the big 1x1 layers of ResNet-50 at batch 64 as GEMMs: C[K_out][pixels] = W[K_out][C] * X[pixels][C]^T.
Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/bf16x3_gemm.so tools/probe/bf16x3_gemm.hip"""
import ctypes
import importlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
lib = ctypes.CDLL(os.path.join(HERE, 'bf16x3_gemm.so'))
lib.bf16x3_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
lib.bf16x3_gemm.restype = ctypes.c_int
lib.bf16x3_presplit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
lib.bf16x3_gemm_wide.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
lib.bf16x3_gemm_presplit.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
    torch.manual_seed(0)
    stream = torch._C._cuda_getCurrentRawStream(0)
    rows = []
    for name, m, n, k in [('c2048->k512 @16x16', 512, 16384, 2048), ('c1024->k2048 @16x16', 2048, 16384, 1024), ('c512->k2048 @16x16', 2048, 16384, 512),
                          ('c256->k1024 @16x16', 1024, 16384, 256), ('c512->k128 @32x32', 128, 65536, 512), ('c64->k256 @64x64', 256, 262144, 64)]:
        A = torch.randn(m, k, device='cuda') * (torch.rand(m, k, device='cuda') * 8 - 4).exp2()       # spread of exponents
        B = torch.randn(n, k, device='cuda')
        C = torch.empty(m, n, device='cuda')
        ref = (A.double() @ B.double().t())
        scale = ref.abs().max().item()
        out = dict(shape=name, M=m, N=n, K=k, gflop=round(2.0 * m * n * k / 1e9, 1))
        for nprod in (6, 62, 3):
            def run():
                rc = lib.bf16x3_gemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), m, n, k, nprod, stream)
                assert rc == 0, rc
            ms = timed(run)
            tag = {6: 'bf16x6', 62: 'bf16x6_bk32', 3: 'bf16x3'}[nprod]
            out[tag + '_ms'] = round(ms, 4)
            out[tag + '_tflops'] = round(2.0 * m * n * k / ms / 1e9, 1)
            out[tag + '_max_err'] = float((C.double() - ref).abs().max().item() / scale)
        for tag, big in (('wide', 0), ('wide128', 1)):
            if big and m % 256:
                continue
            def runw():
                assert lib.bf16x3_gemm_wide(A.data_ptr(), B.data_ptr(), C.data_ptr(), m, n, k, big, stream) == 0
            ms = timed(runw)
            out['bf16x6_%s_ms' % tag], out['bf16x6_%s_tflops' % tag] = round(ms, 4), round(2.0 * m * n * k / ms / 1e9, 1)
            out['bf16x6_%s_max_err' % tag] = float((C.double() - ref).abs().max().item() / scale)
        A3 = torch.empty(3, m, k, dtype=torch.int16, device='cuda')
        B3 = torch.empty(3, n, k, dtype=torch.int16, device='cuda')
        out['presplit_pass_ms'] = round(timed(lambda: (lib.bf16x3_presplit(A.data_ptr(), A3.data_ptr(), m * k, stream), lib.bf16x3_presplit(B.data_ptr(), B3.data_ptr(), n * k, stream))), 4)
        ms = timed(lambda: lib.bf16x3_gemm_presplit(A3.data_ptr(), B3.data_ptr(), C.data_ptr(), m, n, k, stream))
        out['bf16x6_presplit_ms'], out['bf16x6_presplit_tflops'] = round(ms, 4), round(2.0 * m * n * k / ms / 1e9, 1)
        out['bf16x6_presplit_max_err'] = float((C.double() - ref).abs().max().item() / scale)
        ms = timed(lambda: torch.matmul(A, B.t()))
        out['torch_fp32_ms'], out['torch_fp32_tflops'] = round(ms, 4), round(2.0 * m * n * k / ms / 1e9, 1)
        out['torch_fp32_max_err'] = float(((A @ B.t()).double() - ref).abs().max().item() / scale)
        # the product's fp32 MFMA conv on the same problem: x [64, C, h, h] NCHW, w [K_out, C, 1, 1]
        h = int(round((n // 64) ** 0.5))
        x = B.view(64, h, h, k).permute(0, 3, 1, 2).contiguous()
        w = A.view(m, k, 1, 1).contiguous()
        with torch.no_grad():
            y = pkg.ops.conv2d(x, w, None, 1, 0, 1)
            ms = timed(lambda: pkg.ops.conv2d(x, w, None, 1, 0, 1))
        out['p3d_fp32_conv_ms'], out['p3d_fp32_conv_tflops'] = round(ms, 4), round(2.0 * m * n * k / ms / 1e9, 1)
        yref = ref.view(m, 64, h, h).permute(1, 0, 2, 3)
        out['p3d_fp32_conv_max_err'] = float((y.double() - yref).abs().max().item() / scale)
        rows.append(out)
        print(json.dumps(out), flush=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'gpurun_out', 'bf16x3_probe.json'), 'w') as f:
        json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
