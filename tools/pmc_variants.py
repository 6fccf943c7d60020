"""Counter ratios per kernel INSTANTIATION (template arguments kept apart) from a tools/pmc_run.sh directory.
usage: python tools/pmc_variants.py <outdir> <kernel substring>"""
import collections
import csv
import glob
import sys

def main():
    out, key = sys.argv[1], sys.argv[2]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for f in sorted(glob.glob(out + '/*/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if key in k:
                name = k[:k.index('(')] if '(' in k else k
                tot[name][r['Counter_Name']] += float(r['Counter_Value'])
                n[name][r['Counter_Name']] += 1
    for name, v in tot.items():
        g = v.get
        w = g('SQ_WAVE_CYCLES', 1)
        c = n[name]
        print(name, '--', c['SQ_WAVE_CYCLES'], 'dispatches')
        print('  wait_any %.3f  wait_inst %.3f  active %.3f  valu %.3f | VALU per 32 MFMA cycles %.1f | L2 hit %.3f | LDS conflict %.3f' % (
            g('SQ_WAIT_ANY', 0) / w, g('SQ_WAIT_INST_ANY', 0) / w, g('SQ_ACTIVE_INST_ANY', 0) / w, g('SQ_ACTIVE_INST_VALU', 0) / w,
            g('SQ_INSTS_VALU', 0) / max(1, g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 32), g('TCC_HIT_sum', 0) / max(1, g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0)),
            g('SQ_LDS_BANK_CONFLICT', 0) / max(1, g('SQ_LDS_IDX_ACTIVE', 0))))
        print('  per dispatch: FETCH %.1f MB  WRITE %.1f MB  waves %d  MFMA-busy cycles per SIMD %.0f  wave-cycles per wave %.0f' % (
            g('FETCH_SIZE', 0) / max(1, c['FETCH_SIZE']) * 2048 / 1e6, g('WRITE_SIZE', 0) / max(1, c['WRITE_SIZE']) * 1024 / 1e6,
            g('SQ_WAVES', 0) / max(1, c['SQ_WAVES']), g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, c['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024,
            4 * w / max(1, c['SQ_WAVE_CYCLES']) / max(1, g('SQ_WAVES', 0) / max(1, c['SQ_WAVES']))))

if __name__ == "__main__":
    main()
