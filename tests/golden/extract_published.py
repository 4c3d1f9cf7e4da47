"""Extract known-answer values from the reference's published result traces
(/root/reference/gprf_results.tgz: one results.txt per run, line = "step secs objective lscale_ratio
mean_loc_err x_prior ...", written by gprfopt.py:486).  Run in the build container only (the GPU box has
no /root/reference); the output tests/golden/published_traces.json is committed.  Data, not source."""
import io
import json
import os
import tarfile

TGZ = "/root/reference/gprf_results.tgz"
RUNS = [
    "2000_2500_4_0.134164_0.044721_1.0000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "2000_2500_4_0.134164_0.044721_0.1000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "2000_2500_9_0.134164_0.044721_1.0000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "2000_2500_9_0.134164_0.044721_0.1000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "2000_2500_4_0.134164_0.044721_0.1000_50_l-bfgs-b_xcov_-1_0.0100_s0_gprf0",
    "2000_2500_4_0.134164_0.044721_1.0000_50_l-bfgs-b_xcov_-1_0.0100_s0_gprf0",
    "10000_10500_100_0.060000_0.020000_1.0000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "10000_10500_100_0.060000_0.020000_0.1000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    # units of more than 1024 points (round 4): 9 blocks (unaries of ~1100 points, pairs of ~2200), 25 blocks (pairs of ~800), and
    # the single block — the full GP on all 10000 points
    "10000_10500_9_0.060000_0.020000_1.0000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "10000_10500_9_0.060000_0.020000_0.1000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "10000_10500_25_0.060000_0.020000_0.1000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
    "10000_10500_1_0.060000_0.020000_1.0000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0",
]


# runs kept in FULL (round 5: every line, for the run-to-convergence comparison of tests/test_gpu_trace_full.py); the others
# keep their first six lines
FULL = {r for r in RUNS if r.startswith("2000_2500_") or r.startswith("10000_10500_100_")}


def main():
    out = {}
    with tarfile.open(TGZ) as tf:
        for run in RUNS:
            member = [m for m in tf.getmembers() if m.name.endswith(run + "/results.txt")]
            assert len(member) == 1, run
            lines = io.TextIOWrapper(tf.extractfile(member[0])).read().strip().split("\n")
            rec = {"steps": []}
            for ln in (lines if run in FULL else lines[:6]):
                f = ln.split()
                if f[0] == "trueX":
                    continue
                rec["steps"].append({"step": int(f[0]), "objective": f[2], "lscale_ratio": f[3],
                                     "mean_loc_err": f[4], "x_prior": f[5]})
            tx = [ln.split() for ln in lines if ln.startswith("trueX")]
            if tx:
                rec["trueX_objective"] = tx[0][2]
            rec["n_lines"] = len([ln for ln in lines if not ln.startswith("trueX")])
            rec["total_secs"] = max(float(ln.split()[1]) for ln in lines if not ln.startswith("trueX"))
            secs = [float(ln.split()[1]) for ln in lines if not ln.startswith("trueX")]
            d = sorted(b - a for a, b in zip(secs[:-1], secs[1:]))
            rec["median_secs_per_eval"] = d[len(d) // 2] if d else None
            out[run] = rec
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "published_traces.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
