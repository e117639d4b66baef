import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
import arrow_ballista_amd as g, tpch_util as T
tc=g.TaskContext(device=0)
n=T.LINEITEM_ROWS[10]
li=T.gen_lineitem_device(tc,n)
full=T.q1_plan(g.MemoryExec([li]),two_phase=True)
for i in range(3): g.plan.materialize(tc, full.execute(0,tc))
for loop in range(6):
    torch.cuda.synchronize(); t0=time.perf_counter()
    ts=[]
    for i in range(10):
        t1=time.perf_counter()
        out=g.plan.materialize(tc, full.execute(0,tc))
        torch.cuda.synchronize()
        ts.append((time.perf_counter()-t1)*1e3)
    print("loop",loop," ".join("%.2f"%t for t in ts))
