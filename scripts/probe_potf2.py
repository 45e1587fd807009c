import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np, gp_oracle as G
from elfi_amd.gp import GPHandle
X, y, b = G.synthetic_gp_problem(512, 10)
h = G.default_hyper(b, y)
gp = GPHandle(10, 512); gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise']); gp.set_data(X, y); gp.factorize()
lib = gp.lib
lib.elfihip_debug_potf2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
for skip, name in [(5, 'global load/store only'), (4, '+phase A'), (1, '+phase B'), (0, 'all'), (8, 'all+probe')]:
    ms = C.c_float()
    lib.elfihip_debug_potf2(gp.h, skip, 50, C.byref(ms))
    print('%-16s %.1f us' % (name, ms.value * 1e3))
