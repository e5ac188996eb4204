set -e
cd $GRAFT_REPO_ROOT
cp hevcasm_amd/lib/libhevcasm_hip.so /tmp/keep.so
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for v in A S; do
  cp tools/_ab/lib_$v.so hevcasm_amd/lib/libhevcasm_hip.so
  echo "== $v rep $rep"
  python bench.py --steps 6 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps(dict(kernel='value',size='all',ms=d['ms_per_step'])))
for k in d['detail']: print(json.dumps(dict(kernel=k['kernel'],size=k['size'],ms=k['ms_per_pass'])))
"
done; done
cp /tmp/keep.so hevcasm_amd/lib/libhevcasm_hip.so
