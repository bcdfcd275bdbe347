"""Derived figures from the three SQ / TA counter passes of tools/pmc_kernel.sh:
    python tools/pmc_counters_summary.py gpurun_out/pmc_<tag> <kernel-name-substring>[,<substring>...]
kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the eight XCDs); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); LDS busy =
SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); TA busy = TA_BUSY_avr / kernel cycles; medians per launch over the run's launches of every kernel whose name holds the substring."""
import csv, glob, statistics, sys, collections
O, pats = sys.argv[1], sys.argv[2].split(",")
rows = []
for f in glob.glob(f"{O}/p*/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for pat in pats:
    agg = collections.defaultdict(list)
    for r in rows:
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        print("==", pat, ": no launches"); continue
    m = {k: statistics.median(v) for k, v in agg.items()}
    n = len(agg["GRBM_GUI_ACTIVE"])
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    wc = m["SQ_WAVE_CYCLES"]
    print(f"== {pat}  (n = {n} launches, {cyc:.0f} kernel cycles)")
    print(f"   MFMA pipe busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.1f} % ({m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(m['SQ_INSTS_MFMA'], 1):.1f} cycles per MFMA); "
          f"LDS busy {100 * m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.1f} %, bank-conflict cycles {m['SQ_LDS_BANK_CONFLICT']:.0f}; TA busy {100 * m['TA_BUSY_avr'] / cyc:.1f} %")
    print(f"   wave cycles: s_waitcnt / barrier waits {100 * m['SQ_WAIT_ANY'] / wc:.1f} %, issue stalls {100 * m['SQ_WAIT_INST_ANY'] / wc:.1f} %, waiting on LDS instructions "
          f"{100 * m['SQ_WAIT_INST_LDS'] / wc:.1f} %; instructions VALU {m['SQ_INSTS_VALU'] / 1e6:.1f} M, MFMA {m['SQ_INSTS_MFMA'] / 1e6:.2f} M, LDS {m['SQ_INSTS_LDS'] / 1e6:.2f} M, "
          f"VMEM reads {m['SQ_INSTS_VMEM_RD'] / 1e6:.2f} M; L2 read requests {m['TCP_TCC_READ_REQ_sum'] / 1e6:.2f} M")
