cd $GRAFT_REPO_ROOT
for cfg in "CGPT_WF_SORT=0 libcpugpupt.so" "CGPT_WF_SORT=1 libcpugpupt.so" "CGPT_WF_SORT=1 libcpugpupt_band.so"; do
  L=${cfg##* }; E=${cfg% *}
  echo "== $cfg"
  export CGPT_LIB_PATH=$GRAFT_REPO_ROOT/cpugpupathtracing_amd/lib/$L
  env $E timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 2 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step | trace excl', r.get('kernel_ms_per_step'), 'round0', r.get('trace_ms_round0'), 'later', r.get('trace_ms_later'), '| excl pass', r.get('exclusive_pass_ms_per_step'))
"
done
