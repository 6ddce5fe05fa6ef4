"""gpurun_out/<tag>_controls/{timing.json,fetch,write,l2} -> profiles/<tag>_controls.txt (+ .json): the dominant kernel on
the benchmark graph and on the control graphs -- ms (un-profiled HIP events), algorithmic GB/s, fraction of the 8 TB/s
roofline, HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 corrections) and L2 hit rate."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))


def main():
    tag, d, widths = sys.argv[1], sys.argv[2], [int(w) for w in sys.argv[3].split(",")]
    import kernel_controls as kc  # parser only (torch is imported but no GPU is touched)
    timing = json.loads(open(os.path.join(d, "timing.json")).read().strip().splitlines()[-1])
    sets = [kc.parse_pmc_dir(os.path.join(d, sub)) for sub in ("fetch", "write", "l2")]
    traffic = kc.pmc_to_traffic(sets, 3)
    kinds = [k for k in kc.CONTROLS if k in timing]
    rows, i = [], 0
    for kind in kinds:
        for w in widths:
            t = timing[kind]["D=%d" % w]
            tr = traffic[i] if i < len(traffic) else {}
            i += 1
            hbm = tr.get("hbm_read_bytes", 0) + tr.get("hbm_write_bytes", 0)
            rows.append({"graph": kind, "D": w, "schedule": timing[kind]["schedule"], "ms": t["ms"],
                         "achieved_GBps": t["achieved_GBps"], "frac": t["frac"], "algorithmic_bytes": t["algorithmic_bytes"],
                         "hbm_read_bytes": tr.get("hbm_read_bytes"), "hbm_write_bytes": tr.get("hbm_write_bytes"),
                         "traffic_over_algorithmic": round(hbm / t["algorithmic_bytes"], 2) if hbm else None,
                         "fabric_TBps": round(hbm / t["ms"] / 1e9, 2) if hbm else None,
                         "l2_hit": tr.get("l2_hit"), "kernel": tr.get("kernel")})
    lines = ["# copy_u/sum g-SpMM, N = 2,449,029, E = 123,718,280: benchmark graph and control graphs (kernel_controls.py)",
             "# ms: HIP events, un-profiled run; traffic: rocprofv3 --pmc passes (FETCH_SIZE*2 + WRITE_SIZE per launch); peak 8 TB/s",
             "%-9s %4s %-8s %8s %9s %7s %10s %10s %8s %8s %7s" % ("graph", "D", "schedule", "ms", "algo GB/s", "frac", "read MB",
                                                                 "write MB", "traf/alg", "fab TB/s", "L2 hit")]
    for r in rows:
        lines.append("%-9s %4d %-8s %8.3f %9.1f %7.4f %10.1f %10.1f %8s %8s %7s" % (
            r["graph"], r["D"], r["schedule"], r["ms"], r["achieved_GBps"], r["frac"], (r["hbm_read_bytes"] or 0) / 1e6,
            (r["hbm_write_bytes"] or 0) / 1e6, r["traffic_over_algorithmic"], r["fabric_TBps"], r["l2_hit"]))
    if rows and rows[0].get("kernel"):
        lines.append("# kernels: " + "; ".join(sorted({r["kernel"] for r in rows if r.get("kernel")})))
    open(os.path.join(HERE, tag + "_controls.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(rows, open(os.path.join(HERE, tag + "_controls.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
