cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --kernel wavefront --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run CGPT_WF_POOLS=4 CGPT_WF_GROUP=4
run CGPT_WF_POOLS=8 CGPT_WF_GROUP=4
run CGPT_WF_POOLS=8 CGPT_WF_GROUP=16
run CGPT_WF_POOLS=8 CGPT_WF_GROUP=64
run CGPT_WF_POOLS=8 CGPT_WF_GROUP=1
run CGPT_WF_POOLS=8 CGPT_WF_GROUP=16 CGPT_WF_BATCH=32
run CGPT_WF_POOLS=4 CGPT_WF_GROUP=16
