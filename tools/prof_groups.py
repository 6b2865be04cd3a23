import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
steps=13
tot=sum(sum(v) for v in d.values())/steps/1e3
print("total ms/step", round(tot,3))
out=sorted(d.items(), key=lambda kv: -sum(kv[1]))
for k,v in out[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
    print(round(sum(v)/steps/1e3,3), len(v)//steps, round(sum(v)/len(v),1), k)
