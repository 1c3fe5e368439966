"""Write the K3 / K5 entries of profiles/pmc_traffic.json from a rocpd_summary CSV of the PMC passes, stamped with the commit and
the hash of the kernel's sources (bench.py reports `roofline.traffic` only while that hash matches the tree it runs from).

    python tools/pmc_update.py k3 profiles/r04_final_pmc_hbm.csv [streams = 2] [sub_launch_problems = 2048]
    python tools/pmc_update.py k5 profiles/r04_final_conic_pmc_hbm.csv <batch> <elimination_levels>
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counters(path, needle):
    out = {}
    for line in open(path):
        f = line.strip().split(",")
        if len(f) == 5 and needle in f[0] and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            out.setdefault(f[1], []).append((f[0], int(f[2]), float(f[4])))
    return out


def main():
    what, path = sys.argv[1], sys.argv[2]
    jpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    db = json.load(open(jpath))
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    if what == "k3":
        streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
        sub = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
        c = counters(path, "ipm2_solve_kernel")
        # the kernel variant with the most dispatches is the bench's
        pick = lambda rows: max(rows, key=lambda r: r[1])
        fe, wr = pick(c["FETCH_SIZE"]), pick(c["WRITE_SIZE"])
        old = db.get("rocket_landing", {})
        hist = {k: v for k, v in old.items() if k.startswith("round") and isinstance(v, dict)}
        if "FETCH_SIZE_kB_per_sub_launch" in old:
            hist["round%s_final" % old.get("round", "?")] = {k: old[k] for k in ("FETCH_SIZE_kB_per_sub_launch", "WRITE_SIZE_kB_per_sub_launch", "source") if k in old}
        db["rocket_landing"] = dict(round=6, N=100, kernel=fe[0], sub_launch_problems=sub, streams=streams,
                                    FETCH_SIZE_kB_per_sub_launch=fe[2], WRITE_SIZE_kB_per_sub_launch=wr[2], dispatches=fe[1],
                                    commit=commit, sources_sha16=bench.sources_sha16(bench.K3_SOURCES),
                                    source="%s (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py "
                                           "--steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo`)" % os.path.relpath(path, ROOT),
                                    note="8 B/lane loads: the gfx950 x2 FETCH_SIZE correction for 16 B/lane streaming reads is not applied "
                                         "(uncalibrated width); counters are in kB", **hist)
    elif what == "k1v":        # python tools/pmc_update.py k1v <sq_counters.csv> <dispatches per kernel>: executed VALU instructions of the headline's discretize! launch
        nd = int(sys.argv[3])
        tot = 0.0
        for line in open(path):
            f = line.strip().split(",")
            if len(f) == 5 and "discretize_foh_var_kernel" in f[0] and f[1] == "SQ_INSTS_VALU":
                tot += float(f[3])
        db["k1v_rocket_landing"] = dict(round=6, SQ_INSTS_VALU_per_launch=tot / nd, dispatches=nd, commit=commit,
                                        sources_sha16=bench.sources_sha16(bench.K1_SOURCES), source=os.path.relpath(path, ROOT),
                                        note="wavefront VALU instructions of the light + heavy column kernels per discretize! launch of 4096 x 99 intervals; "
                                             "executed fp64 flops <= 64 lanes x 2 x this (every instruction an FMA on all lanes)")
        json.dump(db, open(jpath, "w"), indent=1)
        print(json.dumps(db["k1v_rocket_landing"], indent=1))
        return
    elif what == "k5geo":      # python tools/pmc_update.py k5geo <csv> <key>: a K5 geometry of a bench record (mean per launch over the dispatches)
        key = sys.argv[3]
        c = counters(path, "conic_ipm_kernel")
        tot = lambda rows: (sum(r[1] * r[2] for r in rows) / max(1, sum(r[1] for r in rows)), sum(r[1] for r in rows))
        (fe, nd), (wr, _) = tot(c["FETCH_SIZE"]), tot(c["WRITE_SIZE"])
        db[key] = dict(round=6, FETCH_SIZE_kB_per_launch=fe, WRITE_SIZE_kB_per_launch=wr, dispatches=nd, commit=commit,
                       sources_sha16=bench.sources_sha16(bench.K5_SOURCES), source=os.path.relpath(path, ROOT),
                       note="mean over all conic_ipm_kernel dispatches of the profiled command; counters in kB")
        json.dump(db, open(jpath, "w"), indent=1)
        print(json.dumps(db[key], indent=1))
        return
    else:
        batch, levels = int(sys.argv[3]), int(sys.argv[4])
        c = counters(path, "conic_ipm_kernel")
        pick = lambda rows: max(rows, key=lambda r: r[2])
        fe, wr = pick(c["FETCH_SIZE"]), pick(c["WRITE_SIZE"])
        db["conic_ipm_kernel"] = dict(round=6, program="conic_rocket_landing_N100", batch=batch, kernel=fe[0], elimination_levels=levels,
                                      FETCH_SIZE_kB_per_launch=fe[2], WRITE_SIZE_kB_per_launch=wr[2], commit=commit,
                                      sources_sha16=bench.sources_sha16(bench.K5_SOURCES), source=os.path.relpath(path, ROOT),
                                      note="counters in kB; 8 B/lane loads (no x2 correction applied)")
    json.dump(db, open(jpath, "w"), indent=1)
    print(json.dumps(db["rocket_landing" if what == "k3" else "conic_ipm_kernel"], indent=1)[:800])


if __name__ == "__main__":
    main()
