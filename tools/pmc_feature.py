"""Post-process the two rocprofv3 --pmc passes of tools/pmc.sh (FETCH_SIZE, WRITE_SIZE) over `bench.py --no-graph`:
per-launch HBM-side traffic of the dominant kernel (the teacher's feature projection), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 128-B read requests as 64 B: x2;
WRITE_SIZE calibrated here against the exactly known output size).  Writes profiles/<name>.md and .json.

usage: python tools/pmc_feature.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db profiles/r01_p_pmc_bench_c2
"""
import json
import sqlite3
import sys

N, D, E = 10000, 1024, 512


def per_launch(path, counter):
    cur = sqlite3.connect(path).cursor()
    q = """select p.dispatch_id, sum(p.counter_value), k.duration, k.grid_x from pmc_events p join kernels k
           on k.dispatch_id = p.dispatch_id where k.name like '%feat_gemm%' and p.counter_name = ?
           group by p.dispatch_id order by p.dispatch_id"""
    rows = [r for r in cur.execute(q, (counter,))]          # the two N x D -> E projections of a step: feat_gemm_kernel
    return rows[0::2], rows[1::2]                           # (teacher launches, student launches): teacher runs first


def avg(rows, i):
    return sum(r[i] for r in rows) / len(rows)


def main():
    fdb, wdb, out = sys.argv[1:4]
    ft, fs = per_launch(fdb, "FETCH_SIZE")
    wt, ws = per_launch(wdb, "WRITE_SIZE")
    kib = 1024.0
    fetch_t, fetch_s = 2 * avg(ft, 1) * kib, 2 * avg(fs, 1) * kib       # gfx950 correction: x2
    write_t, write_s = avg(wt, 1) * kib, avg(ws, 1) * kib
    algo_read = N * D * 4 + E * D * 4
    algo_write_t = N * E * 4
    res = {"kernel": "feat_gemm_kernel (teacher feature projection on paired bf16 planes, M=10000 N=512 K=1024)",
           "launches": len(ft), "fetch_bytes": fetch_t, "write_bytes": write_t, "traffic_bytes": fetch_t + write_t,
           "fetch_size_raw_KiB": avg(ft, 1), "write_size_raw_KiB": avg(wt, 1),
           "algorithmic_read_bytes": algo_read, "algorithmic_write_bytes": algo_write_t,
           "write_calibration": avg(wt, 1) * kib / algo_write_t,
           "avg_kernel_us_under_pmc": avg(ft, 2) / 1e3,
           "student_launch": {"fetch_bytes": fetch_s, "write_bytes": write_s},
           "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md 'HBM'); "
                         "WRITE_SIZE x1 (calibrated: reported / known output bytes = write_calibration); unit KiB",
           "source": "tools/pmc.sh fetch FETCH_SIZE bench.py --no-graph ... ; tools/pmc.sh write WRITE_SIZE bench.py --no-graph ... "
                     "(two separate --pmc passes, kernel trace only)"}
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        f.write("# rocprofv3 --pmc: HBM-side traffic of the feature projection (bench.py c2, eager launches)\n\n")
        f.write("Two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with `--kernel-trace` only) of\n"
                "`python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events`; per-dispatch sums over all\n"
                "counter instances, averaged over the launches of `feat_gemm_kernel` (teacher = first of each step, student = second).\n\n")
        f.write("| launch | FETCH_SIZE raw KiB | fetch bytes (x2) | WRITE_SIZE raw KiB | write bytes | algorithmic read | algorithmic write |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|\n")
        f.write(f"| teacher X[10000,1024] -> H | {avg(ft, 1):.1f} | {fetch_t / 1e6:.2f} MB | {avg(wt, 1):.1f} | {write_t / 1e6:.2f} MB | "
                f"{algo_read / 1e6:.2f} MB | {algo_write_t / 1e6:.2f} MB |\n")
        f.write(f"| student X[rows 9700] -> H, dact | {avg(fs, 1):.1f} | {fetch_s / 1e6:.2f} MB | {avg(ws, 1):.1f} | {write_s / 1e6:.2f} MB | "
                f"{(9700 * D * 4 + E * D * 4) / 1e6:.2f} MB | {2 * 9700 * E * 4 / 1e6:.2f} MB |\n\n")
        f.write(f"WRITE_SIZE calibration (teacher): reported / known = {res['write_calibration']:.4f}.\n\n"
                f"Read over-fetch of the teacher launch: {fetch_t / algo_read:.2f}x the algorithmic bytes.  The floor for this tiling is\n"
                f"X once (41.0 MB) + W once per XCD L2 (8 x 2.1 MB = 16.8 MB) = 57.8 MB: the counters sit on the L2's fabric side, so\n"
                f"Infinity-Cache hits (W re-fetched by each XCD) are included.\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
