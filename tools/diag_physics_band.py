import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import harness
from tests.test_hip_env import _pair, _seed_state
from oracle import oracle_lib
oracle_lib.build()
for name, z0 in [("anymal_c_flat", 0.50), ("anymal_c_rough", 0.55), ("cassie", 0.85), ("anymal_c_allrewards", 0.30), ("a1", 0.28), ("anymal_b", 0.50)]:
    for seed in (3, 4, 5):
        hip, ora, z, meta = _pair(name, oracle_lib, n=256)
        rng = np.random.default_rng(seed)
        n, A = 256, meta["num_dofs"]
        origins = z["const_env_origins_init"][rng.integers(0, len(z["const_env_origins_init"]), n)] if meta["custom_origins"] else None
        _seed_state([hip, ora], rng, n, A, z0, origins)
        tau = rng.uniform(-20, 20, (n, A)).astype(np.float32)
        for step in range(6):
            before = ora.get("dof_state").astype(np.float64)
            for e in (hip, ora):
                e.set("torques", tau); e.call("simulate")
            dh, do = hip.get("dof_state").astype(np.float64), ora.get("dof_state").astype(np.float64)
            d = np.abs(dh - do)
            band = d > 2e-4 + 2e-4 * np.abs(do)
            for (i, j, k) in np.argwhere(band):
                cf = np.abs(ora.get("contact_forces")[i]).sum()
                print(f"{name} seed {seed} step {step} env {i} joint {j} {'q' if k == 0 else 'qd'}: err {d[i,j,k]:.2e} value {do[i,j,k]:.3f} dqd {abs(do[i,j,1]-before[i,j,1]):.3f} contact_sum {cf:.1f}")
            hip.set("root_states", ora.get("root_states")); hip.set("dof_state", ora.get("dof_state"))
        hip.close(); ora.close()
print("done")
