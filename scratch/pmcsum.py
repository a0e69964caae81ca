import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        tot = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'biwfa' in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in tot.items(): print(f"{k:24s} {v:.4e}")
