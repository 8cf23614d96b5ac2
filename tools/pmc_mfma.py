"""MFMA-busy fraction per kernel from ONE rocprofv3 --pmc pass ("MFMA utilisation (rocprofv3)").

    python tools/pmc_mfma.py <results.db> [min_avg_us] > profiles/rNN_pmc_mfma_<workload>.md

The pass must have collected SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_MFMA (MOPS form: SQ_INSTS_VALU_MFMA_MOPS_*
where present) and GRBM_GUI_ACTIVE with --kernel-trace only.  Per dispatch the counter instances (one per XCD / SE) are summed.

    MFMA busy %  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs),  kernel cycles = GRBM_GUI_ACTIVE / its instance count
                   (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32)
    effective clock = kernel cycles / kernel duration (the chip clocks to its power budget: DVFS note of the guide)
"""
import sqlite3
import sys

N_SIMD = 256 * 4


def main():
    db = sqlite3.connect(sys.argv[1])
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    cur = db.cursor()
    rows = cur.execute("""select k.name, k.grid_x, p.counter_name, p.dispatch_id, sum(p.counter_value), count(*), k.duration
                          from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id
                          group by p.dispatch_id, p.counter_name""").fetchall()
    agg = {}
    for name, gx, ctr, did, val, ninst, dur in rows:
        key = (name.split("(")[0][:64], gx)
        a = agg.setdefault(key, {})
        c = a.setdefault(ctr, [0.0, 0, 0])
        c[0] += val
        c[1] += 1
        c[2] = ninst
        d = a.setdefault("_dur", [0.0, 0, 0])
        if ctr == "GRBM_GUI_ACTIVE":
            d[0] += dur
            d[1] += 1
    print(f"# rocprofv3 --pmc: matrix-core busy fraction per kernel ({sys.argv[1].split('/')[-1]})\n")
    print("One pass, counters `SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE` (+ SQ wait/active), `--kernel-trace` only;")
    print("per dispatch the counter instances are summed; averages over the dispatches of a (kernel, grid).  MFMA busy % =")
    print("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per instance x 1024 SIMDs); `MFMA insts` = SQ_INSTS_MFMA per dispatch.\n")
    print("| kernel | grid | calls | avg us (under pmc) | eff. clock GHz | MFMA busy % | MFMA insts | SQ_BUSY_CYCLES | wave cycles (quad) | wait_inst % | wait_any % | active_inst % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    out = []
    for (name, gx), a in agg.items():
        if "GRBM_GUI_ACTIVE" not in a:
            continue
        g = a["GRBM_GUI_ACTIVE"]
        calls = g[1]
        cyc = g[0] / g[1] / max(1, g[2])
        dur_us = a["_dur"][0] / max(1, a["_dur"][1]) / 1e3
        if dur_us < min_us:
            continue

        def avg(c):
            return a[c][0] / a[c][1] if c in a else float("nan")

        mf = avg("SQ_VALU_MFMA_BUSY_CYCLES")
        wc = avg("SQ_WAVE_CYCLES")
        # (GRBM_GUI_ACTIVE brackets more than the kernel for short launches - it reads 3-5 "GHz" below ~20 us: there the fraction is
        # stated against the kernel's own duration at the 2.4 GHz peak clock instead)
        busy = 100.0 * mf / (cyc * N_SIMD) if dur_us >= 30 else 100.0 * mf / (dur_us * 2400.0 * N_SIMD)
        out.append((dur_us * calls, f"| `{name}` | {gx} | {calls} | {dur_us:.1f} | {cyc / (dur_us * 1e3):.2f} | {busy:.1f} | "
                    f"{avg('SQ_INSTS_MFMA'):.0f} | {avg('SQ_BUSY_CYCLES'):.0f} | {wc:.0f} | {100 * avg('SQ_WAIT_INST_ANY') / wc:.1f} | "
                    f"{100 * avg('SQ_WAIT_ANY') / wc:.1f} | {100 * avg('SQ_ACTIVE_INST_ANY') / wc:.1f} |"))
    for _, line in sorted(out, reverse=True):
        print(line)


if __name__ == "__main__":
    main()
