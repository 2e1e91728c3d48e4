import sqlite3, glob, sys
db=glob.glob(sys.argv[1]+"/**/*.db",recursive=True)[0]
c=sqlite3.connect(db)
cols=[r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows=c.execute("select name,start,end,duration from kernels order by start").fetchall()
# keep second hinv run: find last copy_damped_lower
idx=[i for i,r in enumerate(rows) if 'copy_damped_lower' in r[0]]
rows=rows[idx[-1]:]
phase='chol'; acc={}
first_put=None
for name,s,e,d in rows:
    if 'put_diag_inverses' in name and phase=='chol': phase='trtri'
    if 'symmetrize' in name: phase='final'
    key=(phase, name.split('(')[0][:40])
    acc[key]=acc.get(key,0)+d
# the product GEMM happens after last put_diag: detect
last_put=max(i for i,r in enumerate(rows) if 'put_diag_inverses' in r[0])
prod=sum(r[3] for r in rows[last_put+1:] if 'gemm_kernel' in r[0] or 'splitk' in r[0])
for k,v in sorted(acc.items()): print(k, round(v/1000 if v>1e6 else v,1))
print('product gemm after trtri', prod)
print('wall', (rows[-1][2]-rows[0][1]))
