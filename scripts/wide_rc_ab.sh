for H in 24 32; do for rc in 1 0; do echo "H=$H RC=$rc"; HPV_WIDE_RC=$rc python scripts/wide_step.py $H 400 2>&1 | grep step; done; done
echo "1d L4 H32"; for rc in 1 0; do HPV_WIDE_RC=$rc python - <<'PY'
import os,sys,time
sys.path.insert(0,os.getcwd())
from hp_vpinns_amd.drivers import poisson1d
from hp_vpinns_amd.init import xavier_init
from hp_vpinns_amd.vpinn import VPINN1D
s=poisson1d.setup(N_Element=16); L=[1,32,32,32,32,1]
m=VPINN1D(s["X_u_train"],s["u_train"],s["X_quad_train"],s["W_quad_train"],s["F_ext_total"],s["grid"],s["X_test"],s["u_test"],L,s["X_f_train"],s["f_train"],init_params=xavier_init(L,1234))
m.h.step(16,False); t0=time.perf_counter(); m.h.step(400,False); print("  %.2f us/iter %s"%((time.perf_counter()-t0)/400*1e6, m.h.kernel_variant()[-60:]))
PY
done
python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -3
