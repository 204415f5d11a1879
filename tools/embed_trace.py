"""Phase stamps of workgroup 0 of the fused patch-embedding kernel (build csrc/embed.hip with -DMMAE_EMBED_TRACE first)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = sys.argv[:1]
import tools.embed_probe as P   # runs the probe (timings) and leaves its operands behind
from multimae_amd import _lib
P.fused(False); torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.mmae_debug_embed_trace(buf)
n = buf[0]; t = [buf[1 + i] for i in range(n)]
print('stamps', n)
names = ['start', 'grouped']
print('total cycles', t[-1] - t[0])
for i in range(1, n):
    print(i, t[i] - t[i - 1])
