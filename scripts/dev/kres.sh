#!/bin/bash
# Kernel resource usage (VGPRs, occupancy, LDS, scratch) of one csrc/*.hip file.  usage: scripts/dev/kres.sh ipx_conv.hip [extra flags]
R=$(cd "$(dirname "$0")/../.." && pwd)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/image-processing_amd/csrc --cuda-device-only -c $R/image-processing_amd/csrc/$f -o /dev/null "$@" -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re,sys,subprocess
name=None; rec={}
for l in sys.stdin:
    m=re.search(r"remark: (?:\S+ )?\s*(Function Name|VGPRs|AGPRs|SGPRs|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)", l)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=="Function Name":
        if name: print(name,rec)
        name=subprocess.run(["c++filt",v],capture_output=True,text=True).stdout.strip(); name=re.sub(r"\(anonymous namespace\)::","",name); name=re.sub(r"\(.*","",name).replace("void ",""); rec={}
    else: rec[k.split(" ")[0]]=v
if name: print(name,rec)
'
