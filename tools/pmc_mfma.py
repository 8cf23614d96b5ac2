"""Matrix-pipe time per kernel from ONE rocprofv3 --pmc pass ("MFMA utilisation (rocprofv3)").

    python tools/pmc_mfma.py <results.db> [min_avg_us] [clocks.json] > profiles/rNN_pmc_mfma_<workload>.md

The pass must have collected SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_MFMA with --kernel-trace only.  Per dispatch
the counter instances (one per XCD / SE) are summed.

    pipe cycles  = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs      (the counter adds every SIMD's busy cycles: 16 per v_mfma_f32_16x16x32_bf16,
                   32 per 32x32x16 - MI355X_MICROARCH.md; cross-check column: SQ_INSTS_MFMA x 16 / 1024)
    MFMA busy %  = pipe cycles / (kernel duration x shader clock)

Round 6: the shader clock is NOT taken from GRBM_GUI_ACTIVE any more.  That counter brackets more than the kernel: the
"effective clock" column of the round-3..5 tables read 2.1 GHz for the 65 us projection, 3.1-3.4 for the 20 us scorers and 5.6-6.0 GHz for
5 us launches - above the part's 2.4 GHz - so every busy % derived from it was biased DOWN.  The clock a launch really holds is measured
inside the kernel (s_memtime against the constant 100 MHz s_memrealtime: tools/exp_proj_prof.py, tools/exp_wgrad.py WG_PROF=1;
profiles/r06_clock.md); clocks.json maps a kernel-name substring to that figure.  Kernels without a stamp are listed at the 2.4 GHz peak
clock AND at 1.6 GHz, which brackets what dense matrix-core launches were stamped at.
"""
import sqlite3
import sys

N_SIMD = 256 * 4


def main():
    import json
    db = sqlite3.connect(sys.argv[1])
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    clocks = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
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
        if ctr == "SQ_WAVE_CYCLES":
            d[0] += dur
            d[1] += 1
    print(f"# rocprofv3 --pmc: matrix-pipe time per kernel ({sys.argv[1].split('/')[-1]})\n")
    print("One pass, counters `SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA` (+ SQ wait/active), `--kernel-trace` only; per dispatch the")
    print("counter instances are summed; averages over the dispatches of a (kernel, grid).  pipe k-cycles = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs")
    print("(`x16` = SQ_INSTS_MFMA x 16 / 1024, the same figure if every MFMA is a 16x16x32).  MFMA busy % = pipe cycles / (duration x clock): at the")
    print("clock STAMPED inside the kernel where one exists (`profiles/r06_clock.md`), else at 2.4 GHz (peak) and at 1.6 GHz.  No GRBM_GUI_ACTIVE:")
    print("that counter brackets more than the kernel (it read up to 6 \"GHz\" for 5 us launches in the round-3..5 tables).\n")
    print("| kernel | grid | calls | avg us (under pmc) | pipe k-cycles | x16 | stamped GHz | MFMA busy % (stamped) | busy % @2.4 | busy % @1.6 | wait_inst % | wait_any % | active_inst % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    out = []
    for (name, gx), a in agg.items():
        if "SQ_WAVE_CYCLES" not in a:
            continue
        calls = a["SQ_WAVE_CYCLES"][1]
        dur_us = a["_dur"][0] / max(1, a["_dur"][1]) / 1e3
        if dur_us < min_us:
            continue

        def avg(c):
            return a[c][0] / a[c][1] if c in a else float("nan")

        pipe = avg("SQ_VALU_MFMA_BUSY_CYCLES") / N_SIMD
        x16 = avg("SQ_INSTS_MFMA") * 16 / N_SIMD
        wc = avg("SQ_WAVE_CYCLES")
        ghz = next((v for k, v in clocks.items() if k in name), None)

        def busy(clk):
            return 100.0 * pipe / (dur_us * 1e3 * clk)

        out.append((dur_us * calls, f"| `{name}` | {gx} | {calls} | {dur_us:.1f} | {pipe / 1e3:.1f} | {x16 / 1e3:.1f} | "
                    f"{'%.2f' % ghz if ghz else '-'} | {'%.1f' % busy(ghz) if ghz else '-'} | {busy(2.4):.1f} | {busy(1.6):.1f} | "
                    f"{100 * avg('SQ_WAIT_INST_ANY') / wc:.1f} | {100 * avg('SQ_WAIT_ANY') / wc:.1f} | {100 * avg('SQ_ACTIVE_INST_ANY') / wc:.1f} |"))
    for _, line in sorted(out, reverse=True):
        print(line)


if __name__ == "__main__":
    main()
