"""One config-1-sized collapsed-bound model evaluated 20 times (for rocprofv3 --kernel-trace)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from bench_configs import data
from rgp_amd.model import DeepAutoreg
which = sys.argv[1] if len(sys.argv) > 1 else '1'
if which == '1':
    U, Y = data(1, 1024, 1, 1)
    m = DeepAutoreg([0, 10], Y, U=U, U_win=10, nDims=[1, 1], X_variance=0.05, num_inducing=20, init='Y', inference_method=None, seed=1)
else:
    U, Y = data(1, 1000, 1, 1)
    m = DeepAutoreg([0, 10, 10], Y, U=U, U_win=10, nDims=[1, 1, 1], X_variance=0.05, num_inducing=50, init='Y', inference_method=None, seed=1)
for _ in range(20): m.evaluate()
m.handle.sync()
