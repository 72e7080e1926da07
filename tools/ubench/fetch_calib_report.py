#!/usr/bin/env python3
"""Reads the two rocprofv3 --pmc outputs of tools/ubench/fetch_calib (see its header) and prints, per kernel, counter KiB / true KiB.

usage: fetch_calib_report.py <dir holding f/ and w/> [out.json]
"""
import csv, glob, json, sys
from collections import defaultdict

GIB, MIB4 = 1 << 30, 4 << 20
# launch order inside one repetition -> (label, bytes read, bytes written)
ORDER = [("read 1 GiB, 4 B/lane", GIB, 0), ("read 1 GiB, 8 B/lane", GIB, 0), ("read 1 GiB, 16 B/lane", GIB, 0),
         ("write 1 GiB, 4 B/lane", 0, GIB), ("write 1 GiB, 8 B/lane", 0, GIB), ("write 1 GiB, 16 B/lane", 0, GIB),
         ("(evict)", GIB, 0), ("read 4 MiB first touch, 8 B/lane", MIB4, 0), ("(evict)", GIB, 0),
         ("read 4 MiB as 128 rows x 4096 doubles", MIB4, 0), ("(evict)", GIB, 0), ("write 4 MiB, 8 B/lane", 0, MIB4)]


def counter_rows(d, name):
	rows = []
	for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
		for r in csv.DictReader(open(f)):
			if r["Counter_Name"] == name and ("read_" in r["Kernel_Name"] or "write_" in r["Kernel_Name"]):
				rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
	agg = defaultdict(float)
	for d_, v in rows:
		agg[d_] += v
	return [agg[k] for k in sorted(agg)]


def main():
	base = sys.argv[1]
	out = {}
	for sub, name, col in (("f", "FETCH_SIZE", 1), ("w", "WRITE_SIZE", 2)):
		vals = counter_rows(f"{base}/{sub}", name)
		per = defaultdict(list)
		for i, v in enumerate(vals):
			label, rd, wr = ORDER[i % len(ORDER)]
			true = (rd, wr)[col - 1]
			if label != "(evict)" and true:
				per[label].append(v * 1024 / true)  # the counter is in KiB
		for label, fr in per.items():
			out.setdefault(label, {})[name + "_over_true"] = sorted(fr)[len(fr) // 2]
	for k, v in out.items():
		print(f"{k:44s} " + "  ".join(f"{n} = {x:.4f}" for n, x in v.items()))
	if len(sys.argv) > 2:
		json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
	main()
