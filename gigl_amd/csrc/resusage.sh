#!/bin/bash
# Prints per-kernel register / LDS / scratch / occupancy figures for gigl_rt.hip (gfx950).
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 --cuda-device-only -c \
  -Rpass-analysis=kernel-resource-usage -o /tmp/gigl_rt_res.o ${1:-gigl_rt.hip} $2 $3 $4 2>&1 | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r"remark: ([A-Za-z ]+): (.*) \[-Rpass", line)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name":
        if cur: print(cur)
        cur={"fn":v}
    else: cur[k]=v
if cur: print(cur)
' | python3 -c '
import sys,ast,subprocess
for line in sys.stdin:
    d=ast.literal_eval(line)
    name=subprocess.run(["c++filt",d["fn"]],capture_output=True,text=True).stdout.strip()
    name=name.replace("(anonymous namespace)::","")[:70]
    print("%-72s vgpr=%s agpr=%s sgpr=%s scratch=%s lds=%s occ=%s"%(name,d.get("VGPRs"),d.get("AGPRs"),d.get("TotalSGPRs"),d.get("ScratchSize [bytes/lane]"),d.get("LDS Size [bytes/block]"),d.get("Occupancy [waves/SIMD]")))
'
