cp koifish_amd/libkf_hip.so /tmp/cur.so
for v in cur hg6 hg8 cur hg6; do
  if [ $v = cur ]; then cp /tmp/cur.so koifish_amd/libkf_hip.so; else cp scratch/ab/libkf_hip_$v.so koifish_amd/libkf_hip.so; fi
  echo "== $v"; python bench.py --steps 40 --warmup 10 --lean 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('us_per_launch'))"
done
cp /tmp/cur.so koifish_amd/libkf_hip.so
