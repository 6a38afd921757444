"""Sums rocprofv3 --pmc counter_collection csv files per (kernel, counter): total over the run, dispatches, and mean per dispatch."""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "xm_wave_kernel" in k:
                k = "xm_wave_kernel:" + next((c for c in ("LightSE", "LightPE", "MidSE", "MidPE", "Heavy") if c in k), "?")
            elif "xm_wave_search_kernel" in k:
                k = "xm_wave_search_kernel"
            elif "xm_align_kernel" in k:
                k = "xm_align_kernel"
            else:
                continue
            acc[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(k, row["Counter_Name"])][1] += 1
    print("==", d)
    for (k, c), (v, n) in sorted(acc.items()):
        print("%-30s %-20s total %18.0f  dispatches %4d  mean/dispatch %16.0f" % (k, c, v, n, v / n))
