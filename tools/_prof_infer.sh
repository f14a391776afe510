cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for k in 1 5; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/infer_k$k -o k$k -- python3 tools/profile_beam.py 128 $k > gpurun_out/infer_k$k.log 2>&1
python3 - $k <<'PY'
import csv,glob,sys
k=sys.argv[1]
f=glob.glob('gpurun_out/infer_k%s/**/*kernel_stats.csv'%k,recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('beam',k,'total kernel ms per call', tot/1e6/3, 'launches per call', sum(int(r['Calls']) for r in rows)/3)
for r in rows[:16]:
    print(r['Name'][:100], int(r['Calls'])/3, round(float(r['TotalDurationNs'])/1e6/3,3), round(float(r['AverageNs'])/1e3,1))
PY
done
