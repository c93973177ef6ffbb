import csv, sys, glob, collections
# usage: pmcsum.py dir  -> per kernel name: mean counter value per dispatch
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0]
        if name.startswith('void '):
            name = name[5:]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if not k.startswith('pd::'): continue
        print(k, {c: (len(v), sum(v)/len(v)) for c, v in d.items()})
