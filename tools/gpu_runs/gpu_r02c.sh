cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
cat /sys/fs/cgroup/cpu.max > gpurun_out/r02c/cgroup.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)))" >> gpurun_out/r02c/cgroup.txt
cat gpurun_out/r02c/cgroup.txt
SCP_MI355X_LIB=$GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_prof.so python tools/ipm_phase_profile.py rocket_landing 4096 8 > gpurun_out/r02c/phase_4096.txt 2>&1
cat gpurun_out/r02c/phase_4096.txt
SCP_MI355X_LIB=$GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_prof.so python tools/ipm_phase_profile.py rocket_landing 1024 8 > gpurun_out/r02c/phase_1024.txt 2>&1
cat gpurun_out/r02c/phase_1024.txt
python - <<'PY' > gpurun_out/r02c/dump.txt 2>&1
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g, bench
pkg = g.load_package()
traj = pkg.TrajectoryProblem("rocket_landing")
B = 4096
pp = bench.mc_pp(traj.mdl, B, 0)
pars = pkg.PTR.Parameters(N=100, Nsub=15, iter_max=15, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
sol, h = pkg.PTR.solve(pbm, pp, device_guess=True)
np.savez_compressed("gpurun_out/r02c/dev_batch.npz", status=np.array([s != "SCP_SOLVED" for s in sol.status]), iters=sol.iterations, feas=sol.feas,
                    ipm_status=h.solver_status, ipm_iters=h.solver_iters, gap=h.gap, J_aug=h.J_aug, active=h.active, p=sol.p)
print("failed", [i for i, s in enumerate(sol.status) if s != "SCP_SOLVED"])
PY
cat gpurun_out/r02c/dump.txt
