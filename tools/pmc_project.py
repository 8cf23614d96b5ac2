"""Post-process the two rocprofv3 --pmc passes of tools/pmc.sh (FETCH_SIZE, WRITE_SIZE) over `bench.py --no-graph`:
per-launch HBM-side traffic of the dominant kernel (bag_project_ws_kernel<0>: the teacher's AND the student's feature projection in
one pass over the raw fp32 bag), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts
128-B read requests as 64 B: x2; WRITE_SIZE calibrated here against the exactly known output size).  Writes <out>.md and .json.

usage: python tools/pmc_project.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db profiles/r02_pmc_bag_project
"""
import json
import sqlite3
import sys

N, D, E = 10000, 1024, 512


def per_launch(path, counter):
    cur = sqlite3.connect(path).cursor()
    q = """select p.dispatch_id, sum(p.counter_value), k.duration, k.grid_x from pmc_events p join kernels k
           on k.dispatch_id = p.dispatch_id where k.name like '%bag_project%' and p.counter_name = ?
           group by p.dispatch_id order by p.dispatch_id"""
    return [r for r in cur.execute(q, (counter,))]


def avg(rows, i):
    return sum(r[i] for r in rows) / len(rows)


def main():
    fdb, wdb, out = sys.argv[1:4]
    f, w = per_launch(fdb, "FETCH_SIZE"), per_launch(wdb, "WRITE_SIZE")
    kib = 1024.0
    fetch, write = 2 * avg(f, 1) * kib, avg(w, 1) * kib                 # gfx950 correction: x2 on reads
    algo_read = N * D * 4 + 2 * E * D * 4                               # X once + the two weight matrices once
    algo_write = 2 * N * E * 4 + N * E * 2                              # H_teacher, H_student fp32 + d out/d pre of the student in fp16
    res = {"kernel": "bag_project_ws_kernel<0> (teacher + student feature projection of one bag, M=10000 N=2x512 K=1024, raw fp32 X)",
           "launches": len(f), "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
           "fetch_size_raw_KiB": avg(f, 1), "write_size_raw_KiB": avg(w, 1),
           "algorithmic_read_bytes": algo_read, "algorithmic_write_bytes": algo_write,
           "write_calibration": write / algo_write, "traffic_over_algorithmic": (fetch + write) / (algo_read + algo_write),
           "avg_kernel_us_under_pmc": avg(f, 2) / 1e3,
           "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md 'HBM'); "
                         "WRITE_SIZE x1 (calibrated: reported / known output bytes = write_calibration); unit KiB",
           "source": "tools/pmc.sh fetch FETCH_SIZE bench.py --no-graph ... ; tools/pmc.sh write WRITE_SIZE bench.py --no-graph ... "
                     "(two separate --pmc passes, kernel trace only)"}
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as fo:
        fo.write("# rocprofv3 --pmc: HBM-side traffic of the single-pass feature projection (bench.py c2, eager launches)\n\n")
        fo.write("Two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with `--kernel-trace` only) of\n"
                 "`python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events`; per-dispatch sums over all\n"
                 "counter instances, averaged over the launches of `bag_project_ws_kernel<0>` (one per step).\n\n")
        fo.write("| launch | FETCH_SIZE raw KiB | fetch bytes (x2) | WRITE_SIZE raw KiB | write bytes | algorithmic read | algorithmic write |\n")
        fo.write("|---|---:|---:|---:|---:|---:|---:|\n")
        fo.write(f"| X[10000,1024] -> H_teacher, H_student, dact16 | {avg(f, 1):.1f} | {fetch / 1e6:.2f} MB | {avg(w, 1):.1f} | {write / 1e6:.2f} MB | "
                 f"{algo_read / 1e6:.2f} MB | {algo_write / 1e6:.2f} MB |\n\n")
        fo.write(f"WRITE_SIZE calibration: reported / known = {res['write_calibration']:.4f}.\n\n"
                 f"Traffic / algorithmic bytes = {res['traffic_over_algorithmic']:.2f} (reads {fetch / algo_read:.2f}x).  The read floor of this tiling is X once\n"
                 f"(41.0 MB) + both weight matrices once per XCD L2 (8 x 4.2 MB = 33.6 MB, served by the Infinity Cache after the first XCD:\n"
                 f"the counters sit on the L2's fabric side, so those hits are counted).\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
