import sys,re,subprocess
# kernel resource usage from the code object's metadata notes
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf","--notes",sys.argv[1]],capture_output=True,text=True).stdout
cur={}
rows=[]
for line in out.splitlines():
    m=re.match(r"\s*-?\s*\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|agpr_count):\s*(.*)",line)
    if m:
        k,v=m.groups()
        if k=="name" and not v.endswith(".kd") and len(v)>20:
            cur["name"]=v
        elif k!="name": cur[k]=v
    if line.strip().startswith(".wavefront_size"):
        if "name" in cur: rows.append(cur)
        cur={}
for r in rows:
    n=r["name"]
    d=subprocess.run(["c++filt",n],capture_output=True,text=True).stdout.strip()
    d=re.sub(r"\(anonymous namespace\)::","",d)
    if len(sys.argv)>2 and sys.argv[2] not in d: continue
    print(f"v{r.get('vgpr_count'):>4} a{r.get('agpr_count','0'):>3} s{r.get('sgpr_count'):>4} scr{r.get('private_segment_fixed_size'):>5} lds{r.get('group_segment_fixed_size'):>6}  {d[:150]}")
