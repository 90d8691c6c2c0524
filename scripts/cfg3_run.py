import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d
from hp_vpinns_amd.init import xavier_init
L2 = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=8, N_el_y=8, with_test_grid=False)
m = poisson2d.build_model(s, L2, init_params=xavier_init(L2, 1234))
m._step(1000, False); m.h.sync()
