import sys, time, json
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg = g.load_package()
import bench
t0=time.time(); r = bench.starship_scvx_record(pkg, B=256, budget_s=30.0); print("starship", time.time()-t0, json.dumps({k: r[k] for k in ("solve_seconds","loop_iterations","seconds_per_loop_iteration","scp_iterations_per_s","frac_failed","frac_dyn_feasible","conic_program","cost_nominal")}), flush=True)
t0=time.time(); r = bench.freeflyer_gusto_record(pkg); print("freeflyer", time.time()-t0, json.dumps(r), flush=True)
