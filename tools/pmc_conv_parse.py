import collections, csv, glob, re, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + '/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv_igemm' not in n and 'conv_wgrad' not in n and 'conv_halo' not in n and 'conv_stem' not in n:
            continue
        s = re.sub(r'\(anonymous namespace\)::', '', n); s = re.sub(r'^void ', '', s).split('(')[0]
        key = '%s grid=%s' % (s, r['Grid_Size'])
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[key][r['Counter_Name']] += 1
for k in agg:
    a = {c: agg[k][c] / cnt[k][c] for c in agg[k]}
    print(k)
    wc = a.get('SQ_WAVE_CYCLES', 1)
    for c in sorted(a):
        print('   %-28s %14.0f  %6.2f%% of wave cycles' % (c, a[c], 100 * a[c] / wc))
    if 'SQ_INSTS_MFMA' in a:
        print('   VALU:MFMA %.2f  SALU:MFMA %.2f  LDS:MFMA %.2f  VMEM:MFMA %.2f' % (a.get('SQ_INSTS_VALU', 0) / a['SQ_INSTS_MFMA'] - 1, a.get('SQ_INSTS_SALU', 0) / a['SQ_INSTS_MFMA'], a.get('SQ_INSTS_LDS', 0) / a['SQ_INSTS_MFMA'], a.get('SQ_INSTS_VMEM_RD', 0) / a['SQ_INSTS_MFMA']))
    if 'SQ_BUSY_CYCLES' in a and 'SQ_VALU_MFMA_BUSY_CYCLES' in a:
        print('   MFMA busy / SQ busy: %.3f' % (a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_BUSY_CYCLES']))
