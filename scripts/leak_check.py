import sys, gc
sys.path.insert(0, ".")
import torch
from hp_vpinns_amd.drivers import poisson2d
from hp_vpinns_amd.init import xavier_init
L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
free0 = None
for i in range(6):
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1))
    m._step(20, True); m._step_record(10); m.predict(s["X_u_train"]); m.loss_and_grad()
    m.h.close(); del m; gc.collect()
    free, total = torch.cuda.mem_get_info()
    if i == 1: free0 = free
    print(i, "free MB", free // 2**20)
print("leak per model (MB):", (free0 - free) / 4 / 2**20)
