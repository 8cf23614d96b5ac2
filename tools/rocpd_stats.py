"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into a per-kernel table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/xxx.md

Kernels are grouped by (name, grid) so the different uses of one GEMM kernel stay apart.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, grid_x, grid_y, grid_z, duration, start, vgpr_count, accum_vgpr_count, lds_size "
                      "from kernels order by start").fetchall()
    if not rows:
        print("no kernel dispatches")
        return
    t0, t1 = rows[0][5], max(r[5] + r[4] for r in rows)
    agg = {}
    for name, gx, gy, gz, dur, start, vg, ag, lds in rows:
        short = name.split("(")[0]
        key = (short, gx, gy, gz)
        a = agg.setdefault(key, [0, 0, 1 << 62, 0, vg, ag, lds])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {sys.argv[1].split('/')[-1]}")
    print(f"\n{len(rows)} dispatches, GPU busy {total/1e6:.3f} ms over a {(t1-t0)/1e6:.3f} ms window "
          f"({100.0*total/(t1-t0):.1f} % busy)\n")
    print("| kernel | grid (threads) | calls | total ms | avg us | min us | max us | % of GPU time | vgpr | agpr | lds B |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short, gx, gy, gz = key
        print(f"| `{short[:70]}` | {gx}x{gy}x{gz} | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | "
              f"{a[3]/1e3:.2f} | {100.0*a[1]/total:.1f} | {a[4]} | {a[5]} | {a[6]} |")


if __name__ == "__main__":
    main()
