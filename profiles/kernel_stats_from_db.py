import sqlite3, sys, glob
for f in glob.glob(sys.argv[1]+'/**/*.db', recursive=True):
    c=sqlite3.connect(f).cursor()
    for r in c.execute("select name,total_calls,average,percentage from top_kernels limit 12"): print("%-70s %4d %10.1f us %5.1f%%"%(r[0][:70],r[1],r[2],r[3]))
