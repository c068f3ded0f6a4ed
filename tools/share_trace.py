import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sonic::", "").replace("void ", "")[:40], r["Queue_Id"], r.get("Grid_Size","?"), r.get("Workgroup_Size","?")) for r in rows)
# the emulated shares: find k_sum_slices occurrences; print the kernels between the last-but-one and last k_part_hist before the last k_sum_slices
idx = [i for i, k in enumerate(ks) if "k_sum_slices" in k[2]]
if not idx: sys.exit("no k_sum_slices")
end = idx[-2]
# walk back to the fr_check / first kernel of that share: previous k_sum_slices + its tree
start = idx[-3]
sel = ks[start:end + 12]
s0 = sel[0][0]; prev = None
for s, e, n, q, g, w in sel:
    gap = (s - prev) / 1e3 if prev else 0
    print("%8.3f dur %7.1f us gap %6.1f %-40s q%s grid %s wg %s" % ((s - s0) / 1e6, (e - s) / 1e3, gap, n, q, g, w))
    prev = e
