#!/usr/bin/env python3
"""Per-launch means of the trace kernel's counters under gpurun_out/TAG/*/ (rocprofv3 --pmc csv)."""
import collections, csv, glob, sys
for tag in sys.argv[1:]:
    print("==", tag)
    for f in sorted(glob.glob('gpurun_out/%s/*/*/*counter_collection.csv' % tag)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if 'trace_kernel' in r['Kernel_Name'] or 'trace3_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
        for k, d in acc.items():
            v = list(d.values())
            print('   %-36s %.5g  (launches %d)' % (k, sum(v) / len(v), len(v)))
