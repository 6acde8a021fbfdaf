#!/usr/bin/env python3
"""Unit-level picture of the particle kernels from tools/pmc_particles.sh's passes (gpurun_out/pmcp/s*/...counter_collection.csv):
per kernel, averages of every collected counter and a few ratios (TA busy share, VALU share, LDS conflict share, wait share)."""
import collections, csv, glob, os, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmcp"
KERNELS = tuple(os.environ.get("PMCP_KERNELS", "k_force_gaussian,k_locate_deposit,k_bin_gather,k_pack_cells").split(","))
data = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/s*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<\d>)?)", r["Kernel_Name"])
        if m and re.sub(r"<\d>", "", m.group(1)) in KERNELS:
            data[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in data.items():
    av = {n: sum(v) / len(v) for n, v in d.items()}
    # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (2.7e7 "cycles" for a 1.3 ms kernel): per-unit shares are taken against one XCD's count
    g = av.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0
    print(k)
    def pct(a, b): return f"{100 * av[a] / av[b]:.1f} %" if a in av and b in av and av[b] else "n/a"
    print("   TA busy / GUI active (per TA, 256 of them):", f"{100 * av.get('TA_TA_BUSY_sum', float('nan')) / 256 / g:.1f} %" if g == g else "n/a",
          "| TD busy:", f"{100 * av.get('TD_TD_BUSY_sum', float('nan')) / 256 / g:.1f} %" if g == g else "n/a",
          "| L1 line accesses per clock per CU:", f"{av.get('TCP_TOTAL_CACHE_ACCESSES_sum', float('nan')) / 256 / g:.2f}" if g == g else "n/a",
          "| LDS index pipe busy:", f"{100 * av.get('SQ_LDS_IDX_ACTIVE', float('nan')) / 256 / g:.1f} %" if g == g else "n/a")
    print("   VALU active / busy cycles:", pct("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"), "| VMEM:", pct("SQ_ACTIVE_INST_VMEM", "SQ_BUSY_CYCLES"), "| LDS:", pct("SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"))
    print("   wave cycles waiting:", pct("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), "| LDS bank conflict share of LDS cycles:", pct("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))
    print("   VALU / VMEM-read / VMEM-write / SALU instructions:", *(f"{av.get(n, float('nan')):.3g}" for n in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU")))
    print("   duration-ish: GUI active cycles", f"{g:.3g}", "| TCP->TCC read latency per req:", f"{av.get('TCP_TCC_READ_REQ_LATENCY_sum', float('nan')) / max(av.get('TCP_TCC_READ_REQ_sum', 1), 1):.0f}", "| TCP pending stall", f"{av.get('TCP_PENDING_STALL_CYCLES_sum', float('nan')):.3g}", "| TA data stall", f"{av.get('TCP_TCP_TA_DATA_STALL_CYCLES_sum', float('nan')):.3g}", "| TCC busy", f"{av.get('TCC_BUSY_sum', float('nan')):.3g}", "| TCC EA rd", f"{av.get('TCC_EA0_RDREQ_sum', float('nan')):.3g}")
    print("   L2 requests / atomics:", f"{av.get('TCC_REQ_sum', float('nan')):.3g}", f"{av.get('TCC_ATOMIC_sum', float('nan')):.3g}", "| TCP accesses / TCC read req:", f"{av.get('TCP_TOTAL_CACHE_ACCESSES_sum', float('nan')):.3g}", f"{av.get('TCP_TCC_READ_REQ_sum', float('nan')):.3g}")
