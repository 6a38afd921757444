"""Sums rocprofv3 --pmc counter_collection csv files per (kernel, counter): mean per dispatch."""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "xm_wave_kernel" in k:
                k = "xm_wave_kernel:" + ("LightSE" if "LightSE" in k else ("LightPE" if "LightPE" in k else "Heavy"))
            elif "xm_align_kernel" in k:
                k = "xm_align_kernel"
            else:
                continue
            acc[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(k, row["Counter_Name"])][1] += 1
    print("==", d)
    for (k, c), (v, n) in sorted(acc.items()):
        if "wave_kernel" in k or "align_kernel" in k:
            print("%-44s %-22s mean/dispatch %16.0f  dispatches %d" % (k, c, v / n, n))
