import sqlite3, re, sys
c=sqlite3.connect(sys.argv[1]); which=int(sys.argv[2]) if len(sys.argv)>2 else 20
rows=c.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'fx_step_begin' in r[0]]
s=idx[which]; e=idx[which+1]
t0=rows[s][1]
for n,st,en in rows[s:e]:
    nm=re.sub(r"\(.*","",n).replace("void ","")[:46]
    print(f"{(st-t0)/1e3:9.1f} {(en-t0)/1e3:9.1f} {(en-st)/1e3:7.1f}  {nm}")
print("step span us", (rows[e][1]-t0)/1e3)
