export TMPDIR=/tmp
for m in 0 auto; do
  if [ $m = auto ]; then unset ORE_XMAP; else export ORE_XMAP=$m; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/xmap_$m -- python3 tools/pmc_pass.py > gpurun_out/xmap_$m.log 2>&1 || exit 1
done
python - <<'PY'
import csv, glob
from collections import defaultdict
for m in ("0","auto"):
    f = glob.glob("gpurun_out/xmap_%s/**/*counter_collection.csv" % m, recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE"]
    per=defaultdict(float); names={}; grid={}
    for r in rows:
        k=int(r["Dispatch_Id"]); per[k]+=float(r["Counter_Value"]); names[k]=r["Kernel_Name"]; grid[k]=(r.get("Grid_Size"),r.get("Workgroup_Size"))
    ids=sorted(per); start=max(i for i in ids if "k_stem1" in names[i])
    tot=defaultdict(float)
    line=[]
    for i in ids:
        if i<start: continue
        n=names[i]
        fam = "kw" if "k_conv_kw" in n else ("ws" if "k_conv3x3_ws" in n else ("patch" if "k_conv3x3_patch" in n else ("igemm" if "k_conv_igemm" in n else ("gs" if "k_conv_gs" in n else "rest"))))
        tot[fam]+=per[i]*2*1024/1e6
        if fam=="kw": line.append("%.1f"%(per[i]*2*1024/1e6))
    print(m, {k:round(v,1) for k,v in tot.items()}, "kw launches MB:", " ".join(line))
PY
