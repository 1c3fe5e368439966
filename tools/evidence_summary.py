"""Print the numbers DESIGN.md section 6 quotes from one evidence directory written by tools/gpu_final.sh (CPU only).
    python tools/evidence_summary.py gpurun_out/r06_final"""
import json
import os
import sys

d = sys.argv[1]


def rd(name):
    try:
        with open(os.path.join(d, name)) as f:
            return f.read()
    except OSError:
        return ""


def last_json(name):
    t = rd(name).strip().split("\n")
    for line in reversed(t):
        if line.startswith("{"):
            return json.loads(line)
    return None


def get(x, path, default=None):
    for k in path.split("/"):
        if not isinstance(x, dict) or k not in x:
            return default
        x = x[k]
    return x


def counters(name, needle):
    out = {}
    for line in rd(name).split("\n"):
        f = line.strip().split(",")
        if len(f) == 5 and needle in f[0]:
            out[f[1]] = (int(f[2]), float(f[3]), float(f[4]))
    return out


def stats(name, needle):
    for line in rd(name).split("\n"):
        f = line.strip().split(",")
        if len(f) >= 13 and needle in f[0]:
            return dict(calls=int(f[1]), total_ms=float(f[2]) * 1e-6, avg_ms=float(f[3]) * 1e-6, min_ms=float(f[4]) * 1e-6, max_ms=float(f[5]) * 1e-6, pct=float(f[6]),
                        vgpr=f[7], sgpr=f[9], lds=f[10], scratch=f[11], grid=f[13] if len(f) > 13 else None)
    return None


print("== pytest:", [l for l in rd("pytest.log").split("\n") if " passed" in l or " failed" in l])
line = last_json("bench.json")
rec = json.load(open(os.path.join(d, "bench_records.json"))) if os.path.exists(os.path.join(d, "bench_records.json")) else {}
if line:
    print("== bench line (%d bytes): value %.1f ms_per_step %.1f steps %s warmup %s" % (len(json.dumps(line)), line["value"], line["ms_per_step"], line["steps"], line["warmup"]))
    print("   roofline", json.dumps(line["roofline"]))
    print("   roofline_discretize", json.dumps(line["roofline_discretize"]))
    print("   cpu_baseline", json.dumps(line.get("cpu_baseline"))[:400])
    print("   value_to_convergence", line.get("value_to_convergence"), "executed", line.get("scp_iterations_executed_per_step"), "failed", line.get("failed_instances"))
    print("   residual", line.get("residual")); print("   parity", json.dumps(line.get("parity"))); print("   strong", line.get("strong_scaling_proxy"))
if rec:
    print("== records: pcie", get(rec, "pcie_inclusive/value"), "kernel alone ms", get(rec, "kernel_launch_ms_alone"))
    print("   to_convergence", {k: get(rec, "to_convergence/" + k) for k in ("seconds", "converged", "iterations_to_convergence", "scp_iterations_executed")})
    g = rec.get("generic_path", {})
    for k, v in g.items():
        if isinstance(v, dict):
            keep = {kk: vv for kk, vv in v.items() if not isinstance(vv, (dict, list)) or kk in ("roofline",)}
            print("   generic/%s: %s" % (k, json.dumps(keep)[:900]))
for pre, needle, label in (("", "ipm2_solve_kernel", "K3"), ("", "discretize_foh_var_kernel", "K1v"), ("conic_16384_", "conic_ipm_kernel", "K5 full chip"),
                           ("k5_starship_", "conic_ipm_kernel", "K5 Starship 256"), ("k5_freeflyer_", "conic_ipm_kernel", "K5 free-flyer 512"), ("k1_", "discretize_foh_kernel", "K1")):
    st = stats(pre + "kernel_stats.csv", needle)
    hb = counters(pre + "pmc_hbm.csv", needle)
    sq = counters(pre + "sq_counters.csv", needle)
    print("== %s: stats %s" % (label, st))
    if hb:
        print("   HBM kB per dispatch: FETCH %.4g WRITE %.4g (dispatches %d)" % (hb.get("FETCH_SIZE", (0, 0, 0))[2], hb.get("WRITE_SIZE", (0, 0, 0))[2], hb.get("FETCH_SIZE", (0, 0, 0))[0]))
    if sq and "SQ_WAVE_CYCLES" in sq:
        wc = sq["SQ_WAVE_CYCLES"][1]
        print("   SQ: " + ", ".join("%s %.1f %%" % (k, 100 * sq[k][1] / wc) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in sq),
              "| per dispatch: " + ", ".join("%s %.3g" % (k, sq[k][2]) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if k in sq))
print("== K3 phase profile:\n" + rd("k3_phase_profile.txt"))
s = rd("starship_n100_scvx_256_100iters.json")
if s:
    j = json.loads(s)
    print("== config 3 to iter_max:", {k: j.get(k) for k in ("loop_iterations", "seconds_per_loop_iteration", "scp_iterations_per_s", "frac_failed", "frac_converged", "iterations_of_converged", "frac_dyn_feasible", "guess_seconds", "stopped_by_budget")})
    print("   oracle_monte_carlo", j.get("oracle_monte_carlo"))
