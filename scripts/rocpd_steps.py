import sqlite3, re, sys
c=sqlite3.connect(sys.argv[1]); steps=float(sys.argv[2]) if len(sys.argv)>2 else 56
rows=c.execute("select name,(end-start) from kernels").fetchall()
agg={}
for n,d in rows:
    s=re.sub(r"\(.*","",n).replace("void ","")
    a=agg.setdefault(s,[0,0]); a[0]+=1; a[1]+=d
tot=0
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    if not k.startswith(("fx_","_Z")): continue
    print(f"{k[:58]:58s} calls/step {a[0]/steps:5.1f}  us/step {a[1]/steps/1e3:8.1f}  avg {a[1]/a[0]/1e3:7.1f}")
    tot+=a[1]/steps/1e3
print("total us/step", round(tot,1))
