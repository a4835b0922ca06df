"""Timing probe for the t192 kernels with pieces switched off (m3l_set_t192_dbg bits) — diagnostic only."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from m3l_amd import _lib as L

LIBP = sys.argv[1] if len(sys.argv) > 1 else L.LIB_PATH
raw = C.CDLL(LIBP)
dev = "cuda:0"
M, D, mlp = int(os.environ.get("PM", "49152")), 192, 768
xn2 = torch.randn(M, D, device=dev).bfloat16()
x1 = torch.randn(M, D, device=dev)
w1 = (torch.randn(mlp, D, device=dev) * 0.05).bfloat16()
w2 = (torch.randn(D, mlp, device=dev) * 0.05).bfloat16()
b1 = torch.randn(mlp, device=dev) * 0.1
b2 = torch.randn(D, device=dev) * 0.1
u = torch.empty(M, mlp, device=dev, dtype=torch.bfloat16)
h = torch.empty_like(u)
xout = torch.empty(M, D, device=dev)
fn = raw.m3l_op_mlp_t192_fwd
fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 10
st = torch.cuda.current_stream().cuda_stream


def run():
    rc = fn(M, mlp, xn2.data_ptr(), x1.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), u.data_ptr(), h.data_ptr(),
            xout.data_ptr(), st)
    assert rc == 0, rc


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    run()
b.record()
torch.cuda.synchronize()
print(f"{os.path.basename(LIBP)} mlp_t192_fwd M={M} mlp={mlp}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us")
