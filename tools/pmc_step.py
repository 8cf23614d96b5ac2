"""Step-level HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc.sh) of an eager bench.py run:
per-kernel bytes per launch (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE x2, WRITE_SIZE x1, unit KiB),
launches per step, and the sum over one step.  Writes <out>.md and <out>.json (bench.py reports the total as roofline.traffic).

usage: python tools/pmc_step.py <fetch_results.db> <write_results.db> <steps in the run> <algorithmic bytes per step> <out prefix> "<title>"
"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    q = """select k.name, k.grid_x, p.dispatch_id, sum(p.counter_value), k.duration from pmc_events p join kernels k
           on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by p.dispatch_id"""
    agg = {}
    for name, gx, did, val, dur in cur.execute(q, (counter,)):
        a = agg.setdefault((name.split("(")[0][:70], gx), [0.0, 0, 0.0])
        a[0] += val; a[1] += 1; a[2] += dur
    return agg


def main():
    fdb, wdb, steps, algo, out, title = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), sys.argv[5], sys.argv[6]
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    rows, tot_r, tot_w = [], 0.0, 0.0
    for key in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 1, 0])[0] * 2 + w.get(k, [0, 1, 0])[0])):
        fr, wr = f.get(key, [0.0, 1, 0.0]), w.get(key, [0.0, 1, 0.0])
        n = max(fr[1], wr[1])
        per_step = n / steps
        if per_step < 0.5:                               # set-up launches (bag generation, fills): not part of a step
            continue
        rd, wt = 2 * fr[0] / fr[1] * 1024.0, wr[0] / wr[1] * 1024.0
        us = (fr[2] / fr[1]) / 1e3
        rows.append({"kernel": key[0], "grid_x": key[1], "launches_per_step": round(per_step, 2), "read_bytes_per_launch": rd,
                     "write_bytes_per_launch": wt, "avg_us_under_pmc": us})
        tot_r += rd * per_step
        tot_w += wt * per_step
    res = {"title": title, "steps_in_run": steps, "step_read_bytes": tot_r, "step_write_bytes": tot_w, "step_traffic_bytes": tot_r + tot_w,
           "algorithmic_bytes_per_step": algo, "traffic_over_algorithmic": (tot_r + tot_w) / algo if algo else None, "kernels": rows,
           "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE x1, unit KiB; two separate --pmc passes "
                         "(kernel trace only) of the same eager command; the counters sit on the L2s' fabric side: Infinity-Cache hits are counted"}
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as fo:
        fo.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: HBM-side traffic per launch and per step - {title}\n\n")
        fo.write(res["correction"] + ".\n\n")
        fo.write("| kernel | grid (threads) | launches / step | read MB / launch | write MB / launch | avg us (under counters) |\n|---|---:|---:|---:|---:|---:|\n")
        for r in rows:
            fo.write(f"| `{r['kernel']}` | {r['grid_x']} | {r['launches_per_step']} | {r['read_bytes_per_launch'] / 1e6:.2f} | "
                     f"{r['write_bytes_per_launch'] / 1e6:.2f} | {r['avg_us_under_pmc']:.1f} |\n")
        fo.write(f"\n**One step: read {tot_r / 1e6:.1f} MB + write {tot_w / 1e6:.1f} MB = {(tot_r + tot_w) / 1e6:.1f} MB**")
        if algo:
            fo.write(f" = {(tot_r + tot_w) / algo:.2f} x the algorithmic {algo / 1e6:.1f} MB (SURVEY 8(d))")
        fo.write(".\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
