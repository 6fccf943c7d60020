"""Per-kernel sums of the counters collected by tools/pmc_run.sh.  usage: python tools/pmc_kernel.py <outdir> <kernel substring>"""
import collections
import csv
import glob
import sys

def main():
    out, key = sys.argv[1], sys.argv[2]
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for f in sorted(glob.glob(out + '/*/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            if key in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value'])
                n[r['Counter_Name']] += 1
    for k in sorted(tot):
        print(f"{k:34s} {tot[k]:18.0f}  ({n[k]} dispatches, {tot[k] / n[k]:.0f} each)")
    g = tot.get
    if g('SQ_WAVE_CYCLES'):
        w = g('SQ_WAVE_CYCLES')
        print(f"wait_any {g('SQ_WAIT_ANY', 0) / w:.3f}  wait_inst_any {g('SQ_WAIT_INST_ANY', 0) / w:.3f}  active_any {g('SQ_ACTIVE_INST_ANY', 0) / w:.3f}  "
              f"active_valu {g('SQ_ACTIVE_INST_VALU', 0) / w:.3f}")
    if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('SQ_INSTS_VALU'):
        print(f"VALU instructions per MFMA-busy 32 cycles: {g('SQ_INSTS_VALU') / (g('SQ_VALU_MFMA_BUSY_CYCLES') / 32):.1f}")
    if g('TCC_HIT_sum'):
        print(f"L2 hit rate {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
    if g('SQ_LDS_IDX_ACTIVE'):
        print(f"LDS bank conflict / active {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.4f}")

if __name__ == "__main__":
    main()
