set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/kt; mkdir -p $O; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- $BENCH > $O/kt.log 2>&1
python $ROOT/tools/rocprof_summary.py $O/kt > $O/kernel_stats.txt 2>&1
python - $O/kt <<'PY'
import glob,sqlite3,sys,os
db=sorted(glob.glob(os.path.join(sys.argv[1],'**','*_results.db'),recursive=True))[-1]
con=sqlite3.connect(db)
tabs=[r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
disp=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; sym=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=con.execute('select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start'%(disp,sym)).fetchall()
# timeline of ~2 steps in the middle
mid=len(rows)//2
t0=rows[mid][1]
for n,s,e in rows[mid:mid+16]:
    print('%-40s start %9.1f us  dur %7.1f us'%(n[:40],(s-t0)/1e3,(e-s)/1e3))
PY
rm -rf $O/kt
cat $O/kernel_stats.txt
