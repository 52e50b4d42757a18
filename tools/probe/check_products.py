import ctypes, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, 'bf16x3_gemm.so'))
lib.bf16x3_gemm_masked.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
def run(A, B, mask):
    C = torch.empty(A.shape[0], B.shape[0], device='cuda')
    assert lib.bf16x3_gemm_masked(A.data_ptr(), B.data_ptr(), C.data_ptr(), A.shape[0], B.shape[0], A.shape[1], mask, None) == 0
    torch.cuda.synchronize()
    return C
def pieces(x):
    xi = x.view(torch.int32)
    hi = (xi & -65536).view(torch.float32)
    r1 = x - hi
    mid = (r1.view(torch.int32) & -65536).view(torch.float32)
    r2 = r1 - mid
    lo = (r2.view(torch.int32) & -65536).view(torch.float32)
    return [hi, mid, lo]
torch.manual_seed(0)
A = torch.randn(128, 16, device='cuda'); B = torch.randn(128, 16, device='cuda')
pa, pb = pieces(A), pieces(B)
print('split exact', (pa[0] + pa[1] + pa[2] - A).abs().max().item())
for ia in range(3):
    for ib in range(3):
        ref = pa[ia].double() @ pb[ib].double().t()
        C = run(A, B, 1 << (ia * 3 + ib))
        print('product', ia, ib, 'rel err', ((C.double() - ref).abs().max() / ref.abs().max()).item())
