for cfg in "lib_dev:" "lib_dev:0" "lib_dev:999999999999" "lib_devNT:" "lib_devNT:0"; do
  lib=${cfg%%:*}; nt=${cfg##*:}
  echo "== $lib IPX_NT_BYTES=$nt"
  if [ -n "$nt" ]; then export IPX_NT_BYTES=$nt; else unset IPX_NT_BYTES; fi
  IPX_LIB=image-processing_amd/$lib/libipx.so timeout -k 10 120 python bench.py --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step',j['ms_per_step'],'value',j['value'],{k:v['us_per_launch'] for k,v in j['kernels'].items()})" || exit 1
done
