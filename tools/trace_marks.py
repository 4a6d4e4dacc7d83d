import csv, sys
rows=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
m=[s for s,e,n in rows if "splice_input" in n]
print(len(m), "marks; intervals (ms):", [round((b-a)/1e6,2) for a,b in zip(m[:-1], m[1:])])
print("first kernel -> last kernel end: %.1f ms" % ((max(e for s,e,n in rows)-rows[0][0])/1e6))
