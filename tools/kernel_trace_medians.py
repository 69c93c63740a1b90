import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows: d[(r["Kernel_Name"][:60],r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()): print(k,len(v),"median %.1f us"%(sorted(v)[len(v)//2]/1e3))
