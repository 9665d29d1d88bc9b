#!/bin/bash
# Per-L2-channel request counts of fast vs slow dispatches of the 5-node chain kernel (placement study).
OUT=/root/repo/gpurun_out/chpmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export DSPFX_RING_TUNE=0
for set in "TCC_EA0_WRREQ TCC_EA0_RDREQ" "TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL" "TCC_REQ TCC_TAG_STALL"; do
  n=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format json -d $OUT/$n -o m -- python3 /root/repo/tools/mode_pmc.py > $OUT/$n.log 2>&1
  ls -la $OUT/$n/*/ 2>/dev/null | head; ls -la $OUT/$n | head
done
python3 - <<'PY'
import json,glob
for f in glob.glob('/root/repo/gpurun_out/chpmc/*/*.json')+glob.glob('/root/repo/gpurun_out/chpmc/*/*/*.json'):
    d=json.load(open(f))
    def walk(o,depth=0,path=''):
        if depth>5: return
        if isinstance(o,dict):
            for k,v in list(o.items())[:30]:
                t=type(v).__name__
                print('  '*depth+f'{k}: {t}'+(f' len={len(v)}' if hasattr(v,'__len__') else ''))
                walk(v,depth+1)
        elif isinstance(o,list) and o:
            print('  '*depth+f'[0]:'); walk(o[0],depth+1)
    print(f); walk(d)
    break
PY
