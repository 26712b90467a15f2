"""Print a rocprofv3 kernel_stats.csv compactly: calls, average us, share, short kernel name."""
import csv, sys, re
for row in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"szg::\(anonymous namespace\)::", "", row["Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print("%6d  %10.1f us  %5.1f %%  %s" % (int(row["Calls"]), float(row["AverageNs"]) / 1e3, float(row["Percentage"]), name[:90]))
