#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 60 --warmup 8 --reps 1 --no-cpu-baseline --no-side-arithmetics --streams 4"
rocprofv3 --kernel-trace --stats -d $O/prof_s4 -o r -- $B > $O/prof_s4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_s4_fp16x4 -o r -- $B --config fp16x4 > $O/prof_s4_fp16x4.log 2>&1
for d in prof_s4 prof_s4_fp16x4; do python $R/tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/kernel_stats_$d.txt 2>&1; done
python - $O <<'PY'
import sqlite3,sys,glob,collections
# concurrency seen by the trace during the frames-in-flight passes: the window in which the least-used stream (a slot that only ever
# runs in-flight passes) has dispatches; union of dispatch intervals vs sum of durations inside it
for d in ("prof_s4","prof_s4_fp16x4"):
    db=glob.glob(f"{sys.argv[1]}/{d}/**/*.db",recursive=True)[0]
    c=sqlite3.connect(db)
    t=[r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    cols=[r[1] for r in c.execute(f"pragma table_info({t})")]
    qc="queue_id" if "queue_id" in cols else "stream_id"
    rows=c.execute(f"select start,end,{qc} from {t} order by start").fetchall()
    cnt=collections.Counter(r[2] for r in rows)
    q=min((k for k,v in cnt.items() if v>300), key=lambda k: cnt[k])
    w0=min(r[0] for r in rows if r[2]==q); w1=max(r[1] for r in rows if r[2]==q)
    rows=[r for r in rows if r[0]>=w0 and r[1]<=w1]
    tot=sum(e-s for s,e,_ in rows); cur_s,cur_e=rows[0][:2]; union=0
    for s,e,_ in rows[1:]:
        if s>cur_e: union+=cur_e-cur_s; cur_s,cur_e=s,e
        else: cur_e=max(cur_e,e)
    union+=cur_e-cur_s
    span=w1-w0
    print(f"{d}: queues {dict(cnt)}; in-flight window {span/1e6:.2f} ms, {len(rows)} dispatches ({len(rows)/48:.0f} frames -> {len(rows)/48/(span/1e9):.0f} frames/s under the tracer), "
          f"GPU busy (union of dispatch intervals) {union/span:.3f} of the window, sum of durations / union = {tot/union:.2f} kernels running on average while busy")
PY
find $O -name "*.db" -delete
head -14 $O/kernel_stats_prof_s4.txt | cut -c1-44,75-140
