cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/b1 -o t -- python /root/repo/tools/lat_b1.py 256 > /tmp/b1.log 2>&1
mkdir -p /root/repo/gpurun_out/profiles_r02; python /root/repo/tools/rocprof_summary.py /tmp/b1/t_results.db > /root/repo/gpurun_out/profiles_r02/r02_b1_256_kernel_trace.md; head -30 /root/repo/gpurun_out/profiles_r02/r02_b1_256_kernel_trace.md
python - >> /root/repo/gpurun_out/profiles_r02/r02_b1_256_kernel_trace.md <<'PY'
import sqlite3
c=sqlite3.connect('/tmp/b1/t_results.db')
rows=c.execute("select start, end, name from kernels order by start").fetchall()
# take a window of the eager f16x3 timing loop: find median generate by splitting on conv_img kernel
import statistics
ends=[i for i,r in enumerate(rows) if 'conv_img' in r[2]]
spans=[]
for a,b in zip(ends[5:25], ends[6:26]):
    seg=rows[a+1:b+1]
    busy=sum(r[1]-r[0] for r in seg)
    wall=seg[-1][1]-seg[0][0]
    spans.append((len(seg), busy/1e3, wall/1e3))
print('launches per render, sum of kernel us, wall us (first->last kernel):', spans[:6])
PY
