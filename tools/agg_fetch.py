import csv, glob, sys
from collections import defaultdict
for d in sys.argv[1:]:
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_pcg_fused_q' not in n: continue
        g = int(r['Grid_Size'])
        agg[(n.split('octane::')[1].split('(')[0], g, r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(d.split('/')[-1], k, len(v), 'mean MB (x2 KiB): %.0f' % (sum(v) / len(v) * 2 * 1024 / 1e6))
